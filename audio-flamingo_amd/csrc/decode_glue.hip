// Decode-step glue: what follows each weight-streaming GEMV of a decoder layer when only a handful of rows are alive.
//
// A decode token runs four Linears per layer on the GEMV kernel (gemm.hip), which leaves fp32 split-K partials ws[split][M][N].  Done
// as separate kernels, "sum the partials", bias, RoPE, cache append, residual, RMSNorm and SwiGLU are nine ~5 us launches per layer -
// 1.5 ms of a 4 ms token.  Here each GEMV is followed by ONE kernel that sums its partials and applies everything up to the next GEMV's
// input, with the arithmetic (and the bf16 rounding points) of the stand-alone kernels:
//   qkv      -> + bias, bf16, rotate q and k (apply_rotary_pos_emb, modeling_qwen2.py:112-135), q to a buffer, k / v into the KV cache
//   o_proj   -> bf16, + residual (Qwen2DecoderLayer :284), RMSNorm of the sum (:294, Qwen2RMSNorm :247-252) -> (x2, h2)
//   gate|up  -> bf16, silu(gate) * up (Qwen2MLP :46-48)
//   down     -> bf16, + residual (:297), RMSNorm with the NEXT layer's input_layernorm (or the final norm) -> (x, h)
#include "common.h"
#include "../../include/afk.h"

namespace {

// fixed-order sum of the split-K partials (the order of gemm_splitk_reduce_kernel); the loads of up to 16 partials are issued together
__device__ __forceinline__ float sum_splits(const float* __restrict__ ws, int splits, int64_t stride, int64_t idx) {
    float v[16];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp) v[sp] = sp < splits ? ws[idx + sp * stride] : 0.f;
    float s = v[0];
#pragma unroll
    for (int sp = 1; sp < 16; ++sp)
        if (sp < splits) s += v[sp];
    for (int sp = 16; sp < splits; ++sp) s += ws[idx + sp * stride];
    return s;
}

// one thread = one rotation pair (d, d + D/2) of a q / k head, or two consecutive v features
__global__ __launch_bounds__(256) void decode_qkv_finish_kernel(const float* __restrict__ ws, int splits, int M, const bf16* __restrict__ bias,
                                                                const bf16* __restrict__ cos_t, const bf16* __restrict__ sin_t,
                                                                const int* __restrict__ pos, bf16* __restrict__ q_out, bf16* __restrict__ Kc,
                                                                int64_t kc_bs, bf16* __restrict__ Vt, int64_t vt_bs, int spad,
                                                                const int* __restrict__ start_dev, int Hq, int Hkv, int D) {
    const int half = D >> 1, N = (Hq + 2 * Hkv) * D, nq = Hq * D, nk = Hkv * D;
    const int pairs = N >> 1;
    const int64_t stride = (int64_t)M * N;
    const int start = *start_dev;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < M * pairs; t += gridDim.x * 256) {
        const int m = t / pairs, pi = t % pairs;
        const int head = pi / half, d = pi % half;  // heads: Hq q heads, Hkv k heads, then Hkv v "heads"
        if (head < Hq + Hkv) {
            const int c1 = head * D + d, c2 = c1 + half;
            const float a = rbf(sum_splits(ws, splits, stride, (int64_t)m * N + c1) + (float)bias[c1]);
            const float b = rbf(sum_splits(ws, splits, stride, (int64_t)m * N + c2) + (float)bias[c2]);
            const int p = pos[m];
            const float o1 = rbf(rbf_strict(a * (float)cos_t[(int64_t)p * D + d]) + rbf_strict(-b * (float)sin_t[(int64_t)p * D + d]));   // rbf_strict: contraction-proof (common.h)
            const float o2 = rbf(rbf_strict(b * (float)cos_t[(int64_t)p * D + half + d]) + rbf_strict(a * (float)sin_t[(int64_t)p * D + half + d]));
            if (head < Hq) {
                q_out[(int64_t)m * nq + c1] = (bf16)o1;
                q_out[(int64_t)m * nq + c2] = (bf16)o2;
            } else {
                bf16* kr = Kc + m * kc_bs + (int64_t)start * nk + (c1 - nq);
                kr[0] = (bf16)o1;
                kr[half] = (bf16)o2;
            }
        } else {
            const int c = nq + nk + (pi - (Hq + Hkv) * half) * 2;  // two consecutive v features
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v = sum_splits(ws, splits, stride, (int64_t)m * N + c + e) + (float)bias[c + e];
                Vt[m * vt_bs + (int64_t)(c + e - nq - nk) * spad + start] = (bf16)v;
            }
        }
    }
}

// one block = one row: x = bf16(sum) + residual ; h = w * bf16(x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(1024) void decode_residual_rmsnorm_kernel(const float* __restrict__ ws, int splits, int M, int N,
                                                                       const bf16* __restrict__ residual, const bf16* __restrict__ w, float eps,
                                                                       bf16* __restrict__ x_out, bf16* __restrict__ h_out) {
    __shared__ float red[16];
    const int m = blockIdx.x;
    const int64_t stride = (int64_t)M * N;
    float ss = 0.f;
    for (int c = threadIdx.x; c < N; c += 1024) {
        const float v = rbf(rbf(sum_splits(ws, splits, stride, (int64_t)m * N + c)) + (float)residual[(int64_t)m * N + c]);
        x_out[(int64_t)m * N + c] = (bf16)v;
        ss += v * v;
    }
    ss = block_sum<16>(ss, red);
    const float rstd = rsqrtf(ss * (1.f / (float)N) + eps);
    for (int c = threadIdx.x; c < N; c += 1024) {
        const float v = (float)x_out[(int64_t)m * N + c];
        h_out[(int64_t)m * N + c] = (bf16)((float)w[c] * rbf(v * rstd));
    }
}

__global__ __launch_bounds__(256) void decode_swiglu_kernel(const float* __restrict__ ws, int splits, int M, int I, bf16* __restrict__ a_out) {
    const int64_t stride = (int64_t)M * 2 * I;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < M * I; t += gridDim.x * 256) {
        const int m = t / I, c = t % I;
        const float g = rbf(sum_splits(ws, splits, stride, (int64_t)m * 2 * I + c));
        const float u = rbf(sum_splits(ws, splits, stride, (int64_t)m * 2 * I + I + c));
        a_out[(int64_t)m * I + c] = (bf16)(rbf(g * sigmoid_f(g)) * u);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int afk_decode_qkv_finish(const float* ws, int splits, int M, const void* bias, const void* cos_t, const void* sin_t, const int* pos,
                                     void* q_out, void* kcache, int64_t kc_bs, void* vtcache, int64_t vt_bs, int spad, const int* start_dev,
                                     int Hq, int Hkv, int D, void* stream) {
    AFK_REQUIRE(ws && bias && cos_t && sin_t && pos && q_out && kcache && vtcache && start_dev, "afk_decode_qkv_finish: null pointer");
    AFK_REQUIRE(splits >= 1 && M >= 1 && M <= 16 && Hq > 0 && Hkv > 0 && D % 2 == 0, "afk_decode_qkv_finish: bad shape");
    const int total = M * ((Hq + 2 * Hkv) * D / 2);
    hipLaunchKernelGGL(decode_qkv_finish_kernel, dim3((unsigned)afk_cdiv(total, 256)), dim3(256), 0, ST, ws, splits, M, (const bf16*)bias,
                       (const bf16*)cos_t, (const bf16*)sin_t, pos, (bf16*)q_out, (bf16*)kcache, kc_bs, (bf16*)vtcache, vt_bs, spad, start_dev, Hq,
                       Hkv, D);
    AFK_LAUNCH_CHECK("afk_decode_qkv_finish");
    return AFK_OK;
}

extern "C" int afk_decode_residual_rmsnorm(const float* ws, int splits, int M, int N, const void* residual, const void* w, float eps, void* x_out,
                                           void* h_out, void* stream) {
    AFK_REQUIRE(ws && residual && w && x_out && h_out && splits >= 1 && M >= 1 && M <= 16 && N > 0, "afk_decode_residual_rmsnorm: bad args");
    hipLaunchKernelGGL(decode_residual_rmsnorm_kernel, dim3((unsigned)M), dim3(1024), 0, ST, ws, splits, M, N, (const bf16*)residual, (const bf16*)w, eps,
                       (bf16*)x_out, (bf16*)h_out);
    AFK_LAUNCH_CHECK("afk_decode_residual_rmsnorm");
    return AFK_OK;
}

extern "C" int afk_decode_swiglu(const float* ws, int splits, int M, int I, void* a_out, void* stream) {
    AFK_REQUIRE(ws && a_out && splits >= 1 && M >= 1 && M <= 16 && I > 0, "afk_decode_swiglu: bad args");
    hipLaunchKernelGGL(decode_swiglu_kernel, dim3((unsigned)afk_cdiv((int64_t)M * I, 256)), dim3(256), 0, ST, ws, splits, M, I, (bf16*)a_out);
    AFK_LAUNCH_CHECK("afk_decode_swiglu");
    return AFK_OK;
}
