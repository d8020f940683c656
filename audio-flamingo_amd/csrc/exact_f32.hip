// afk_x32_*: the AF3 inference forward in EXACT fp32 ("an fp32 mode of our kernels - f32 MFMA v_mfma_f32_32x32x2_f32, exact fp32 - should
// give bit-exact tokens unconditionally on the tiny config").  A VERIFICATION mode, not a fast path: activations fp32 in HBM, weights read
// as the bf16 values the checkpoint holds and widened in registers (the reference's fp32 run uses exactly those values), every product and sum in fp32 -
// the matrix pipe's fp32 MFMA for the Linears, fp32 VALU everywhere else.  No bf16 rounding point exists anywhere between the log-mel features and the
// logits, so the argmax equals the reference's fp32 argmax wherever two logits are not tied to ~1e-6 (summation order is the only difference left).
// Host side: audio_flamingo_amd/exact.py (AFK_EXACT_FP32=1); test: tests/test_exact_gpu.py (argmax equal at EVERY valid position, generate() ids identical,
// no "confident rows" filter).  Oracle lines as in the bf16 kernels: modeling_audioflamingo3.py:117-245,380-439, modeling_qwen2.py:46-48,105-135,195-298.
#include <math.h>
#include "common.h"
#include "../../include/afk.h"

namespace {

// ------------------------------------------------------------------ Linear: C[M,N] = epi((A[M,K] . W[N,K]^T + bias) * alpha), A / C fp32, W / bias bf16
// One wave = one 32 x 32 output tile on v_mfma_f32_32x32x2_f32 (A: lane -> (row l % 32, k l / 32); B: lane -> (k l / 32, col l % 32); D: register r of lane l
// = (row 8 (r / 4) + 4 (l / 32) + r % 4, col l % 32)).  Per step of 8 along K a lane loads ONE float4 of its A row (k = k0 + 4 (l / 32) + 0..3) and the four
// bf16 of its W row at the same k: MFMA j of the step multiplies the k pairs {k0 + j, k0 + 4 + j} - any pairing of the reduction index is a valid order of
// the fp32 sum as long as A and B agree on it.  Operands come straight from global memory (L2-resident at the sizes this mode is for).
__global__ __launch_bounds__(256) void x32_linear_kernel(const float* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw, float* __restrict__ C,
                                                         int64_t ldc, int M, int N, int K, const bf16* __restrict__ bias, const float* __restrict__ res,
                                                         int64_t ldr, int res_mod, float alpha, int gelu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = blockIdx.x * 4 + wave, mt = blockIdx.y;
    if (nt * 32 >= N) return;
    const int l31 = lane & 31, hi = lane >> 5;
    const int am = min(mt * 32 + l31, M - 1), bn = min(nt * 32 + l31, N - 1);   // clamped rows: edge tiles compute garbage rows / columns that are never stored
    const float* ap = A + (int64_t)am * lda + 4 * hi;
    const bf16* wp = W + (int64_t)bn * ldw + 4 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        const f32x4 a = *(const f32x4*)(ap + k0);
        const bf16x4 w = *(const bf16x4*)(wp + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], (float)w[j], acc, 0, 0, 0);
    }
    const int n = nt * 32 + l31;
    if (n >= N) return;
    const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
        if (m >= M) continue;
        float v = (acc[r] + bv) * alpha;
        if (gelu) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));   // exact-erf GELU (transformers/activations.py:70-89)
        if (res) v += res[(int64_t)(res_mod > 0 ? m % res_mod : m) * ldr + n];
        C[(int64_t)m * ldc + n] = v;
    }
}

// ------------------------------------------------------------------ LayerNorm (eps 1e-5, bias) / RMSNorm (eps 1e-6): one wave per row, two passes, fp32
__global__ __launch_bounds__(256) void x32_norm_kernel(const float* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ b, float* __restrict__ y,
                                                       int64_t rows, int D, float eps, int rms) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float s = 0.f;
    if (!rms) {
        for (int c = lane; c < D; c += 64) s += xr[c];
        s = wave_sum(s) / (float)D;
    }
    float v = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float d = xr[c] - s;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    for (int c = lane; c < D; c += 64) {
        const float t = (xr[c] - s) * rstd;
        y[row * D + c] = rms ? (float)w[c] * t : t * (float)w[c] + (float)b[c];
    }
}

// ------------------------------------------------------------------ attention: one wave per (query row, head); scores of the row in LDS; fp32 softmax
// visible keys of query s of sample b: [lo_b, min(hi_b, causal ? s + 1 : S)); a row that sees no key gives a zero output row (torch SDPA's result, which
// the reference's masked rows produce - oracle/af3_oracle.py _sdpa)
__global__ __launch_bounds__(256) void x32_attention_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ Kp, int64_t ldk,
                                                            const float* __restrict__ Vp, int64_t ldv, float* __restrict__ O, int64_t ldo, int S, int Hq, int Hkv,
                                                            int D, float scale, int causal, const int* __restrict__ kv_lo, const int* __restrict__ kv_len) {
    extern __shared__ __attribute__((aligned(16))) float x32_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    if (s >= S) return;   // whole wave: no block-level barrier below
    // the score row is padded to a multiple of 4 floats: the query row behind it is read with 16-byte LDS loads, and greedy decoding grows S by one
    // per token (3 of 4 steps would otherwise read it misaligned: ADVICE r05).  D % 4 == 0 (checked by the launcher) keeps every wave's base aligned too.
    const int Sp = (S + 3) & ~3;
    float* sc = x32_smem + (int64_t)wave * (Sp + D);
    float* qs = sc + Sp;
    const int hk = h / (Hq / Hkv);
    const int lo = kv_lo ? max(kv_lo[b], 0) : 0;
    int hiK = kv_len ? min(kv_len[b], S) : S;
    if (causal) hiK = min(hiK, s + 1);
    const float* qrow = Q + ((int64_t)b * S + s) * ldq + (int64_t)h * D;
    for (int d = lane; d < D; d += 64) qs[d] = qrow[d];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lo + lane; j < hiK; j += 64) {
        const float* kr = Kp + ((int64_t)b * S + j) * ldk + (int64_t)hk * D;
        float dot = 0.f;
        for (int d = 0; d < D; d += 4) {
            const f32x4 kv = *(const f32x4*)(kr + d);
            const f32x4 qv = *(const f32x4*)(qs + d);
            dot = fmaf(qv[0], kv[0], dot);
            dot = fmaf(qv[1], kv[1], dot);
            dot = fmaf(qv[2], kv[2], dot);
            dot = fmaf(qv[3], kv[3], dot);
        }
        dot *= scale;
        sc[j] = dot;
        mx = fmaxf(mx, dot);
    }
    mx = wave_max(mx);
    float* orow = O + ((int64_t)b * S + s) * ldo + (int64_t)h * D;
    if (mx == -INFINITY) {
        for (int d = lane; d < D; d += 64) orow[d] = 0.f;
        return;
    }
    float l = 0.f;
    for (int j = lo + lane; j < hiK; j += 64) {
        const float p = expf(sc[j] - mx);
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / l;
    for (int d = lane; d < D; d += 64) {
        float o = 0.f;
        for (int j = lo; j < hiK; ++j) o = fmaf(sc[j] * inv, Vp[((int64_t)b * S + j) * ldv + (int64_t)hk * D + d], o);   // softmax first, then P . V (the reference's order)
        orow[d] = o;
    }
}

// ------------------------------------------------------------------ RoPE (rotate-half, modeling_qwen2.py:105-135), in place on the q | k heads of a fused row
__global__ __launch_bounds__(256) void x32_rope_kernel(float* __restrict__ x, int64_t ld, const float* __restrict__ cs, const float* __restrict__ sn, int64_t rows, int S,
                                                       int nheads, int D) {
    const int half = D / 2;
    const int64_t n = rows * nheads * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % half);
        const int hd = (int)((i / half) % nheads);
        const int64_t row = i / ((int64_t)half * nheads);
        const int pos = (int)(row % S);
        float* p = x + row * ld + (int64_t)hd * D;
        const float a = p[d], c = p[d + half];
        // x * cos + rotate_half(x) * sin: first half pairs with -x[d + half], second half with +x[d]; cos / sin tables are [S, D] (both halves equal)
        p[d] = a * cs[(int64_t)pos * D + d] + (-c) * sn[(int64_t)pos * D + d];
        p[d + half] = c * cs[(int64_t)pos * D + d + half] + a * sn[(int64_t)pos * D + d + half];
    }
}

__global__ __launch_bounds__(256) void x32_silu_mul_kernel(const float* __restrict__ gu, float* __restrict__ out, int64_t rows, int I) {
    const int64_t n = rows * I;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / I;
        const int c = (int)(i % I);
        const float g = gu[r * 2 * I + c], u = gu[r * 2 * I + I + c];
        out[i] = (g / (1.f + expf(-g))) * u;   // silu(gate) * up (modeling_qwen2.py:46-48)
    }
}

// ------------------------------------------------------------------ Conv1d(k = 3, pad = 1, stride 1 | 2) + GELU (+ position table), direct form
// x_cmajor != 0: x is [W][C][T] (the log-mel features); else [W][T][C] (the output of the first convolution).  w = [E][C][3] bf16 as nn.Conv1d stores it.
__global__ __launch_bounds__(256) void x32_conv_kernel(const float* __restrict__ x, int x_cmajor, const bf16* __restrict__ w, const bf16* __restrict__ bias,
                                                       const bf16* __restrict__ pos, float* __restrict__ y, int Wn, int C, int T, int E, int stride) {
    const int To = (T - 1) / stride + 1;
    const int64_t n = (int64_t)Wn * To * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i % E);
        const int to = (int)((i / E) % To);
        const int wi = (int)(i / ((int64_t)E * To));
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int t = to * stride + k - 1;
                if (t < 0 || t >= T) continue;
                const float xv = x_cmajor ? x[((int64_t)wi * C + c) * T + t] : x[((int64_t)wi * T + t) * C + c];
                acc = fmaf(xv, (float)w[((int64_t)e * C + c) * 3 + k], acc);
            }
        }
        float v = acc + (float)bias[e];
        v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (pos) v += (float)pos[(int64_t)to * E + e];   // + embed_positions after the permute (modeling_audioflamingo3.py:382-385)
        y[i] = v;
    }
}

__global__ __launch_bounds__(256) void x32_avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t Wn, int T, int E) {
    const int To = T / 2;
    const int64_t n = Wn * To * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i % E);
        const int64_t r = i / E, wi = r / To, t = r % To;
        y[i] = 0.5f * (x[(wi * T + 2 * t) * E + e] + x[(wi * T + 2 * t + 1) * E + e]);
    }
}

// out[row] = audio[src[row]] where src[row] >= 0 (a <sound> placeholder), else embed[ids[row]] widened
__global__ __launch_bounds__(256) void x32_embed_scatter_kernel(const int64_t* __restrict__ ids, const int* __restrict__ src, const float* __restrict__ audio,
                                                                const bf16* __restrict__ embed, float* __restrict__ out, int64_t rows, int H) {
    const int64_t n = rows * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H;
        const int c = (int)(i % H);
        const int sidx = src ? src[r] : -1;
        out[i] = sidx >= 0 ? audio[(int64_t)sidx * H + c] : (float)embed[ids[r] * H + c];
    }
}

inline unsigned grid1d(int64_t n) { return (unsigned)std::min<int64_t>(afk_cdiv(n, 256), 65535); }

}  // namespace

extern "C" int afk_x32_linear(const float* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K, const void* bias,
                              const float* residual, int64_t ldr, int res_mod, float alpha, int gelu, void* stream) {
    AFK_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "afk_x32_linear: bad arguments");
    AFK_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ldw % 4 == 0, "afk_x32_linear: K %% 8 == 0 and 16-byte (A) / 8-byte (W) aligned rows required (K %d, lda %lld, ldw %lld)", K,
                (long long)lda, (long long)ldw);
    hipLaunchKernelGGL(x32_linear_kernel, dim3((unsigned)afk_cdiv(afk_cdiv(N, 32), 4), (unsigned)afk_cdiv(M, 32)), dim3(256), 0, (hipStream_t)stream, A, lda,
                       (const bf16*)W, ldw, C, ldc, M, N, K, (const bf16*)bias, residual, ldr, res_mod, alpha, gelu);
    AFK_LAUNCH_CHECK("afk_x32_linear");
    return AFK_OK;
}

extern "C" int afk_x32_norm(const float* x, const void* w, const void* b, float* y, int64_t rows, int D, float eps, int rms, void* stream) {
    AFK_REQUIRE(x && w && y && rows > 0 && D > 0 && (rms || b), "afk_x32_norm: bad arguments");
    hipLaunchKernelGGL(x32_norm_kernel, dim3((unsigned)afk_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, (const bf16*)w, (const bf16*)b, y, rows, D, eps, rms);
    AFK_LAUNCH_CHECK("afk_x32_norm");
    return AFK_OK;
}

extern "C" int afk_x32_attention(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo, int B, int S, int Hq,
                                 int Hkv, int D, float scale, int causal, const int* kv_lo, const int* kv_len, void* stream) {
    AFK_REQUIRE(Q && K && V && O && B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && D > 0 && D % 4 == 0, "afk_x32_attention: bad arguments");
    AFK_REQUIRE(ldk % 4 == 0 && ((uintptr_t)K & 15) == 0, "afk_x32_attention: key rows must be 16-byte aligned");
    AFK_REQUIRE(D % 4 == 0, "afk_x32_attention: head_dim %d is not a multiple of 4", D);
    const size_t lds = (size_t)4 * (((S + 3) & ~3) + D) * sizeof(float);
    AFK_REQUIRE(lds <= 160 * 1024, "afk_x32_attention: S + D = %d exceeds the LDS score rows of this verification kernel (160 KiB / 16 bytes)", S + D);
    static int once = hipFuncSetAttribute((const void*)x32_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : 1;
    (void)once;
    hipLaunchKernelGGL(x32_attention_kernel, dim3((unsigned)afk_cdiv(S, 4), (unsigned)Hq, (unsigned)B), dim3(256), lds, (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, O,
                       ldo, S, Hq, Hkv, D, scale, causal, kv_lo, kv_len);
    AFK_LAUNCH_CHECK("afk_x32_attention");
    return AFK_OK;
}

extern "C" int afk_x32_rope(float* x, int64_t ld, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int nheads, int D, void* stream) {
    AFK_REQUIRE(x && cos_tab && sin_tab && rows > 0 && S > 0 && nheads > 0 && D > 0 && D % 2 == 0, "afk_x32_rope: bad arguments");
    hipLaunchKernelGGL(x32_rope_kernel, dim3(grid1d(rows * nheads * (D / 2))), dim3(256), 0, (hipStream_t)stream, x, ld, cos_tab, sin_tab, rows, S, nheads, D);
    AFK_LAUNCH_CHECK("afk_x32_rope");
    return AFK_OK;
}

extern "C" int afk_x32_silu_mul(const float* gate_up, float* out, int64_t rows, int I, void* stream) {
    AFK_REQUIRE(gate_up && out && rows > 0 && I > 0, "afk_x32_silu_mul: bad arguments");
    hipLaunchKernelGGL(x32_silu_mul_kernel, dim3(grid1d(rows * I)), dim3(256), 0, (hipStream_t)stream, gate_up, out, rows, I);
    AFK_LAUNCH_CHECK("afk_x32_silu_mul");
    return AFK_OK;
}

extern "C" int afk_x32_conv3_gelu(const float* x, int x_cmajor, const void* w, const void* bias, const void* pos, float* y, int W, int C, int T, int E, int stride,
                                  void* stream) {
    AFK_REQUIRE(x && w && bias && y && W > 0 && C > 0 && T > 0 && E > 0 && (stride == 1 || stride == 2), "afk_x32_conv3_gelu: bad arguments");
    const int To = (T - 1) / stride + 1;
    hipLaunchKernelGGL(x32_conv_kernel, dim3(grid1d((int64_t)W * To * E)), dim3(256), 0, (hipStream_t)stream, x, x_cmajor, (const bf16*)w, (const bf16*)bias,
                       (const bf16*)pos, y, W, C, T, E, stride);
    AFK_LAUNCH_CHECK("afk_x32_conv3_gelu");
    return AFK_OK;
}

extern "C" int afk_x32_avgpool2(const float* x, float* y, int64_t W, int T, int E, void* stream) {
    AFK_REQUIRE(x && y && W > 0 && T > 1 && E > 0, "afk_x32_avgpool2: bad arguments");
    hipLaunchKernelGGL(x32_avgpool2_kernel, dim3(grid1d(W * (T / 2) * E)), dim3(256), 0, (hipStream_t)stream, x, y, W, T, E);
    AFK_LAUNCH_CHECK("afk_x32_avgpool2");
    return AFK_OK;
}

extern "C" int afk_x32_embed_scatter(const int64_t* ids, const int* src, const float* audio, const void* embed, float* out, int64_t rows, int H, void* stream) {
    AFK_REQUIRE(ids && embed && out && rows > 0 && H > 0 && (!src || audio), "afk_x32_embed_scatter: bad arguments");
    hipLaunchKernelGGL(x32_embed_scatter_kernel, dim3(grid1d(rows * H)), dim3(256), 0, (hipStream_t)stream, ids, src, audio, (const bf16*)embed, out, rows, H);
    AFK_LAUNCH_CHECK("afk_x32_embed_scatter");
    return AFK_OK;
}
