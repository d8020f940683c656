// Decode step, single sequence: one launch per Linear, five per decoder layer (round 4).
//
// Rounds 2-3 ran every Linear of a decode token as a weight-streaming GEMV with split-K partials followed by a glue kernel (decode_glue.hip): ten
// launches per layer, and at ~4 us per dependent launch inside the replayed HIP graph the 28 layers spend 1.1 ms of a 3.6 ms token between kernels
// (the weights stream in 2.5 ms).  Here a wave owns TWO complete output rows (all K), so no partial sums exist and everything up to the next Linear's
// input happens where the dot products end:
//     qkv      prologue: RMSNorm of the residual stream (every wave re-derives the row statistic from the 7 KiB row it needs anyway)
//              epilogue: + bias, bf16, rotate the (d, d + D/2) pair the wave owns (modeling_qwen2.py:112-135), q to its buffer, k / v into the KV cache
//     o_proj   epilogue: bf16, + residual (Qwen2DecoderLayer :284)                                           -> x2
//     gate|up  prologue: RMSNorm(x2) (:294, Qwen2RMSNorm :247-252); the wave owns gate row c AND up row I + c:  silu(g) * u (Qwen2MLP :46-48) -> a
//     down     epilogue: bf16, + residual (:297)                                                             -> x  (the next layer's qkv launch normalises it)
// with the bf16 rounding points of the stand-alone kernels.  Attention is ONE launch as well (attention_decode.hip with nsplit = 1 writes the output
// itself).  Parallelism comes from the row count instead of split-K: 2 304 / 1 792 / 18 944 / 1 792 waves per Linear of AF3-7B, each with 8 x 16-byte
// weight loads in flight (the K loop is unrolled by four), non-temporal (streamed once).
#include "common.h"
#include "../../include/afk.h"

namespace {

enum { PRO_PLAIN = 0, PRO_RMS = 1 };
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

struct ChainArgs {
    const bf16* x;        // input row [K]
    const bf16* normw;    // PRO_RMS: RMSNorm weight [K]
    float eps;
    const bf16* W;        // [N, K] row-major (nn.Linear layout)
    int64_t ldw;
    int N, K;
    const bf16* bias;     // EPI_QKV
    const bf16* cos_t;    // [pos][D]
    const bf16* sin_t;
    const int* pos;       // position of the new token (device)
    const int* start;     // cache slot of the new token (device)
    bf16* q_out;          // [Hq * D]
    bf16* Kc;             // [Smax][Hkv * D]
    bf16* Vt;             // [Hkv * D][spad]
    int spad, Hq, Hkv, D;
    const bf16* residual; // EPI_RESID [N]
    bf16* out;            // EPI_RESID [N], EPI_SWIGLU [N / 2]
};

__device__ __forceinline__ float dot8(const bf16x8 a, const bf16x8 b, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16x2 u = {a[2 * e], a[2 * e + 1]}, v = {b[2 * e], b[2 * e + 1]};
        acc = __builtin_amdgcn_fdot2_f32_bf16(u, v, acc, false);
    }
    return acc;
}

constexpr int MAXCH = 8;   // PRO_RMS keeps the normalised row in registers: K <= 8 x 512

template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_chain_kernel(ChainArgs p) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npairs = p.N >> 1;
    if (pair >= npairs) return;
    // ---- the two rows of this wave
    int r0, r1;
    const int half = p.D >> 1, nq = p.Hq * p.D, nk = p.Hkv * p.D;
    if (EPI == EPI_QKV) {
        const int rot_pairs = (p.Hq + p.Hkv) * half;
        if (pair < rot_pairs) {
            r0 = (pair / half) * p.D + pair % half;
            r1 = r0 + half;
        } else {
            r0 = nq + nk + 2 * (pair - rot_pairs);
            r1 = r0 + 1;
        }
    } else if (EPI == EPI_SWIGLU) {
        r0 = pair;
        r1 = npairs + pair;
    } else {
        r0 = 2 * pair;
        r1 = r0 + 1;
    }
    const bf16* w0 = p.W + (int64_t)r0 * p.ldw;
    const bf16* w1 = p.W + (int64_t)r1 * p.ldw;
    const int nch = (p.K + 511) >> 9;
    float a0 = 0.f, a1 = 0.f;
    if (PRO == PRO_RMS) {
        // h = w_norm * bf16(x * rsqrt(mean(x^2) + eps))  (cast BEFORE the weight multiply, Qwen2RMSNorm :247-252), kept as packed bf16
        // the wave's whole weight rows first (2 x nch 16-byte loads in flight per lane: they stream while the norm below is computed)
        bf16x8 wu[MAXCH], wv[MAXCH], hv[MAXCH];
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int k = (c << 9) + lane * 8;
            const bool ok = c < nch && k < p.K;
            const int kk = ok ? k : 0;                // masked lanes re-read the row's first vector (valid memory) and drop it
            wu[c] = __builtin_nontemporal_load((const bf16x8*)(w0 + kk));
            wv[c] = __builtin_nontemporal_load((const bf16x8*)(w1 + kk));
            hv[c] = *(const bf16x8*)(p.x + kk);
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[c][e] = (bf16)0.f;
            }
        }
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)hv[c][e] * (float)hv[c][e];
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss * (1.f / (float)p.K) + p.eps);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int k = (c << 9) + lane * 8;
            const bool ok = c < nch && k < p.K;
            const bf16x8 nw = *(const bf16x8*)(p.normw + (ok ? k : 0));
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[c][e] = ok ? (bf16)((float)nw[e] * rbf((float)hv[c][e] * rstd)) : (bf16)0.f;   // masked: h = 0 -> no contribution
            a0 = dot8(wu[c], hv[c], a0);
            a1 = dot8(wv[c], hv[c], a1);
        }
    } else {
        int c = 0;
        for (; c + 4 <= nch && ((c + 4) << 9) <= p.K; c += 4) {   // four whole chunks: 8 weight loads in flight
            bf16x8 u[4], v[4], xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = ((c + i) << 9) + lane * 8;
                u[i] = __builtin_nontemporal_load((const bf16x8*)(w0 + k));
                v[i] = __builtin_nontemporal_load((const bf16x8*)(w1 + k));
                xv[i] = *(const bf16x8*)(p.x + k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0 = dot8(u[i], xv[i], a0);
                a1 = dot8(v[i], xv[i], a1);
            }
        }
        for (; c < nch; ++c) {
            const int k = (c << 9) + lane * 8;
            if (k < p.K) {
                const bf16x8 u = __builtin_nontemporal_load((const bf16x8*)(w0 + k));
                const bf16x8 v = __builtin_nontemporal_load((const bf16x8*)(w1 + k));
                const bf16x8 xv = *(const bf16x8*)(p.x + k);
                a0 = dot8(u, xv, a0);
                a1 = dot8(v, xv, a1);
            }
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane != 0) return;
    if (EPI == EPI_QKV) {
        const float a = rbf(a0 + (float)p.bias[r0]);
        const float b = rbf(a1 + (float)p.bias[r1]);
        const int rot_pairs = (p.Hq + p.Hkv) * half;
        const int start = *p.start;
        if (pair < rot_pairs) {
            const int d = pair % half;
            const int64_t ps = (int64_t)(*p.pos) * p.D;
            const float o1 = rbf(rbf(a * (float)p.cos_t[ps + d]) + rbf(-b * (float)p.sin_t[ps + d]));
            const float o2 = rbf(rbf(b * (float)p.cos_t[ps + half + d]) + rbf(a * (float)p.sin_t[ps + half + d]));
            if (r0 < nq) {
                p.q_out[r0] = (bf16)o1;
                p.q_out[r1] = (bf16)o2;
            } else {
                bf16* kr = p.Kc + (int64_t)start * nk + (r0 - nq);
                kr[0] = (bf16)o1;
                kr[half] = (bf16)o2;
            }
        } else {
            p.Vt[(int64_t)(r0 - nq - nk) * p.spad + start] = (bf16)a;
            p.Vt[(int64_t)(r1 - nq - nk) * p.spad + start] = (bf16)b;
        }
    } else if (EPI == EPI_RESID) {
        p.out[r0] = (bf16)(rbf(a0) + (float)p.residual[r0]);
        p.out[r1] = (bf16)(rbf(a1) + (float)p.residual[r1]);
    } else {
        const float g = rbf(a0), u = rbf(a1);
        p.out[pair] = (bf16)(rbf(g * sigmoid_f(g)) * u);
    }
}

template <int PRO, int EPI>
int launch_chain(const ChainArgs& p, hipStream_t st) {
    const int npairs = p.N >> 1;
    hipLaunchKernelGGL((gemv_chain_kernel<PRO, EPI>), dim3((unsigned)afk_cdiv(npairs, 4)), dim3(256), 0, st, p);
    return AFK_OK;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int afk_decode_chain_qkv(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int K, const void* bias, const void* cos_t,
                                    const void* sin_t, const int* pos, void* q_out, void* kcache, void* vtcache, int spad, const int* start_dev, int Hq,
                                    int Hkv, int D, void* stream) {
    AFK_REQUIRE(x && norm_w && W && bias && cos_t && sin_t && pos && q_out && kcache && vtcache && start_dev, "afk_decode_chain_qkv: null pointer");
    AFK_REQUIRE(K > 0 && K % 8 == 0 && K <= MAXCH * 512 && ldw % 8 == 0 && Hq > 0 && Hkv > 0 && D % 2 == 0 && spad > 0, "afk_decode_chain_qkv: unsupported shape (K <= 4096, K %% 8 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = (Hq + 2 * Hkv) * D; p.K = K;
    p.bias = (const bf16*)bias; p.cos_t = (const bf16*)cos_t; p.sin_t = (const bf16*)sin_t; p.pos = pos; p.start = start_dev; p.q_out = (bf16*)q_out;
    p.Kc = (bf16*)kcache; p.Vt = (bf16*)vtcache; p.spad = spad; p.Hq = Hq; p.Hkv = Hkv; p.D = D;
    launch_chain<PRO_RMS, EPI_QKV>(p, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_qkv");
    return AFK_OK;
}

extern "C" int afk_decode_chain_linear_residual(const void* x, const void* W, int64_t ldw, int N, int K, const void* residual, void* out, void* stream) {
    AFK_REQUIRE(x && W && residual && out && N > 0 && N % 2 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "afk_decode_chain_linear_residual: bad arguments");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.residual = (const bf16*)residual; p.out = (bf16*)out; p.D = 2;
    launch_chain<PRO_PLAIN, EPI_RESID>(p, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual");
    return AFK_OK;
}

extern "C" int afk_decode_chain_gate_up(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int I, int K, void* act_out, void* stream) {
    AFK_REQUIRE(x && norm_w && W && act_out && I > 0 && K > 0 && K % 8 == 0 && K <= MAXCH * 512 && ldw % 8 == 0, "afk_decode_chain_gate_up: unsupported shape (K <= 4096, K %% 8 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = 2 * I; p.K = K; p.out = (bf16*)act_out; p.D = 2;
    launch_chain<PRO_RMS, EPI_SWIGLU>(p, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_gate_up");
    return AFK_OK;
}
