// Decode-time attention (one query row per sample, keys/values in the KV cache): split-KV ("flash decoding").
//
// Oracle: Qwen2Attention.forward with a cache, modeling_qwen2.py:195-234 (softmax(q k^T * scale) v over the cached positions).
// A Q = 1 problem has no query dimension to parallelise over, so one block per (batch, head) walks the whole cache serially
// (measured 54 us per layer at 800 cached tokens on the interval MFMA kernel: 1.5 ms of a 5.1 ms token).  Here the key range of each
// sample is cut into `nsplit` chunks: grid (nsplit, Hq, B); every block scores its chunk with coalesced 16-byte key-row loads
// (D/8 lanes per key, xor-shuffle reduce), takes its own softmax (max m, sum l) and its partial output  sum_k p_k V[k]  from the
// TRANSPOSED value cache Vt[b][hk][d][key] (thread = one d, 8 keys per 16-byte load), and leaves (m, l, o[D]) in a workspace; a second
// tiny kernel merges the chunks:  o = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s.   HBM-bound: the cache is read exactly once.
// The visible key interval [lo, hi) of every sample comes from DEVICE memory (krange), so a captured HIP graph of the decode step can be
// replayed while the sequence grows.
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int MAXCHUNK = 4096;  // keys per split held in LDS

// FINAL (round 4, afk_attn_decode_fused): no combine launch - the LAST chunk block of a (batch, head) pair to finish merges the chunks itself.
// Every block publishes (m, l, o[D]), fences at agent scope (the chunks of one head run on different XCDs whose L2s are not coherent) and bumps the
// pair's counter; the block that reads nsplit - 1 acquires, folds the partials in chunk order (same arithmetic as attn_decode_combine_kernel: the
// result does not depend on which block came last), writes the bf16 output and resets the counter for the next call.  Nobody waits for anybody.
template <int D, bool FINAL>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(const bf16* __restrict__ Q, int64_t q_bs, int64_t q_hs, const bf16* __restrict__ Kc,
                                                                int64_t k_bs, int64_t k_rs, int64_t k_hs, const bf16* __restrict__ Vt,
                                                                int64_t vt_bs, int spad, const int* __restrict__ krange, int Hq, int Hkv,
                                                                float scale, float* __restrict__ ws, bf16* __restrict__ O, int64_t o_bs, int64_t o_hs) {
    constexpr int LPK = D / 8;        // lanes per key row
    constexpr int KPP = 256 / LPK;    // keys scored per pass of the block
    constexpr int PARTS = 256 / D;    // threads per output feature
    __shared__ float sc[MAXCHUNK];
    __shared__ float red[8];
    __shared__ float part[256];
    const int split = blockIdx.x, nsplit = gridDim.x, h = blockIdx.y, b = blockIdx.z, hk = h / (Hq / Hkv);
    const int t = threadIdx.x;
    const int lo = krange[2 * b], hi = krange[2 * b + 1];
    // chunk boundaries on multiples of 8 keys (16-byte alignment of the transposed value rows)
    const int a0 = lo & ~7, total = max(hi - a0, 0);
    const int chunk = min((((total + nsplit - 1) / nsplit) + 7) & ~7, MAXCHUNK);
    const int c0 = a0 + split * chunk, c1 = min(c0 + chunk, hi);  // keys [max(c0, lo), c1)
    const int n = max(c1 - c0, 0);
    float* out = ws + ((int64_t)(b * Hq + h) * nsplit + split) * (D + 2);
    __shared__ int last_flag;
    // FINAL: the merge by the last-arriving block of this (batch, head) pair
    auto finish = [&]() {
        if (!FINAL) return;
        int* counters = (int*)(ws + (int64_t)gridDim.z * Hq * nsplit * (D + 2));
        __threadfence();                                   // release (every writer): this block's (m, l, o) reach memory before the counter moves
        __syncthreads();
        if (t == 0) {
            const int prev = atomicAdd(&counters[b * Hq + h], 1);
            last_flag = (prev == nsplit - 1);
            if (last_flag) counters[b * Hq + h] = 0;       // self-resetting: the next call (a HIP-graph replay) starts from zero again
        }
        __syncthreads();
        if (!last_flag) return;
        __threadfence();                                   // acquire (every reader): drop whatever this CU / XCD cached of the other blocks' slots
        const float* base = ws + (int64_t)(b * Hq + h) * nsplit * (D + 2);
        if (t < D) {
            float M = NEG_INF;
            for (int s = 0; s < nsplit; ++s) M = fmaxf(M, __builtin_nontemporal_load(base + s * (D + 2)));
            float L = 0.f, o = 0.f;
            if (M != NEG_INF) {
                for (int s = 0; s < nsplit; ++s) {
                    const float w = __builtin_amdgcn_exp2f(__builtin_nontemporal_load(base + s * (D + 2)) - M);
                    L += w * __builtin_nontemporal_load(base + s * (D + 2) + 1);
                    o += w * __builtin_nontemporal_load(base + s * (D + 2) + 2 + t);
                }
            }
            O[b * o_bs + h * o_hs + t] = (bf16)(L > 0.f ? o / L : 0.f);
        }
    };
    if (n <= 0 || c1 <= lo) {
        if (t < D) out[2 + t] = 0.f;
        if (t == 0) { out[0] = NEG_INF; out[1] = 0.f; }
        finish();
        return;
    }
    // ---- scores
    const int sub = t % LPK;
    const bf16x8 qv = *(const bf16x8*)(Q + b * q_bs + h * q_hs + sub * 8);
    const bf16* kbase = Kc + b * k_bs + hk * k_hs + sub * 8;
    const float c2 = scale * LOG2E;
    float mx = NEG_INF;
    for (int i0 = 0; i0 < n; i0 += KPP) {
        const int i = i0 + t / LPK;
        const int key = c0 + i;
        float s = 0.f;
        if (i < n) {
            const bf16x8 kv = *(const bf16x8*)(kbase + (int64_t)key * k_rs);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16x2 a = {qv[2 * e], qv[2 * e + 1]}, c = {kv[2 * e], kv[2 * e + 1]};
                s = __builtin_amdgcn_fdot2_f32_bf16(a, c, s, false);
            }
        }
#pragma unroll
        for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (i < n && sub == 0) {
            const float v = (key >= lo) ? s * c2 : NEG_INF;  // log2 domain
            sc[i] = v;
        }
        if (i < n && key >= lo) mx = fmaxf(mx, s * c2);
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float msafe = (m == NEG_INF) ? 0.f : m;
    float ls = 0.f;
    for (int i = t; i < n; i += 256) {
        const float p = __builtin_amdgcn_exp2f(sc[i] - msafe);
        sc[i] = p;
        ls += p;
    }
    ls = wave_sum(ls);
    if ((t & 63) == 0) red[4 + (t >> 6)] = ls;
    __syncthreads();
    const float l = red[4] + red[5] + red[6] + red[7];
    // ---- partial output: thread -> feature d, part -> every PARTS-th group of 8 keys
    const int d = t % D, pt = t / D;
    const bf16* vrow = Vt + b * vt_bs + ((int64_t)hk * D + d) * spad + c0;  // c0 % 8 == 0
    float acc = 0.f;
    for (int g = pt * 8; g < n; g += PARTS * 8) {
        const bf16x8 vv = *(const bf16x8*)(vrow + g);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (g + e < n) acc += sc[g + e] * (float)vv[e];
    }
    part[t] = acc;
    __syncthreads();
    if (t < D) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) o += part[t + q * D];
        out[2 + t] = o;
    }
    if (t == 0) { out[0] = m; out[1] = l; }
    finish();
}

template <int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float* __restrict__ ws, int nsplit, bf16* __restrict__ O, int64_t o_bs,
                                                               int64_t o_hs, int Hq) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* base = ws + (int64_t)(b * Hq + h) * nsplit * (D + 2);
    float M = NEG_INF;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[s * (D + 2)]);
    float L = 0.f, o = 0.f;
    if (M != NEG_INF) {
        for (int s = 0; s < nsplit; ++s) {
            const float w = __builtin_amdgcn_exp2f(base[s * (D + 2)] - M);  // m = -inf -> 0
            L += w * base[s * (D + 2) + 1];
            o += w * base[s * (D + 2) + 2 + d];
        }
    }
    O[b * o_bs + h * o_hs + d] = (bf16)(L > 0.f ? o / L : 0.f);
}

}  // namespace

extern "C" int afk_attn_decode_workspace_floats(int B, int Hq, int D, int nsplit) { return B * Hq * nsplit * (D + 2) + B * Hq; }   // + arrival counters (afk_attn_decode_fused)

static int attn_decode_impl(bool fused, const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                            const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                            int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    AFK_REQUIRE(Q && Kc && Vt && O && krange && workspace, "afk_attn_decode: null pointer");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && (D == 64 || D == 128) && nsplit >= 1 && nsplit <= 64, "afk_attn_decode: bad shape");
    AFK_REQUIRE(q_bs % 8 == 0 && q_hs % 8 == 0 && k_bs % 8 == 0 && k_rs % 8 == 0 && k_hs % 8 == 0 && vt_bs % 8 == 0 && spad % 8 == 0,
                "afk_attn_decode: strides must keep 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nsplit, (unsigned)Hq, (unsigned)B);
#define AFK_AD(DD)                                                                                                                                      \
    do {                                                                                                                                                \
        if (fused) {                                                                                                                                    \
            hipLaunchKernelGGL((attn_decode_split_kernel<DD, true>), grid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs,   \
                               k_hs, (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace, (bf16*)O, o_bs, o_hs);                            \
        } else {                                                                                                                                        \
            hipLaunchKernelGGL((attn_decode_split_kernel<DD, false>), grid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs,  \
                               k_hs, (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace, (bf16*)O, o_bs, o_hs);                            \
            hipLaunchKernelGGL(attn_decode_combine_kernel<DD>, dim3((unsigned)Hq, (unsigned)B), dim3(DD), 0, st, workspace, nsplit, (bf16*)O, o_bs,     \
                               o_hs, Hq);                                                                                                               \
        }                                                                                                                                               \
    } while (0)
    if (D == 128) AFK_AD(128); else AFK_AD(64);
#undef AFK_AD
    AFK_LAUNCH_CHECK("afk_attn_decode");
    return AFK_OK;
}

extern "C" int afk_attn_decode(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                               const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                               int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    return attn_decode_impl(false, Q, q_bs, q_hs, Kc, k_bs, k_rs, k_hs, Vt, vt_bs, spad, O, o_bs, o_hs, krange, B, Hq, Hkv, D, scale, nsplit, workspace, stream);
}

// one launch: the last chunk block of every (batch, head) pair merges the chunks.  workspace: afk_attn_decode_workspace_floats(...) floats whose LAST B * Hq words are the
// per-pair arrival counters - they must read ZERO before the first call (the kernel leaves them at zero)
extern "C" int afk_attn_decode_fused(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                                     const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                                     int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    return attn_decode_impl(true, Q, q_bs, q_hs, Kc, k_bs, k_rs, k_hs, Vt, vt_bs, spad, O, o_bs, o_hs, krange, B, Hq, Hkv, D, scale, nsplit, workspace, stream);
}
