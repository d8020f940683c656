// Shared pieces of the NT GEMM kernels (argument block, tile->workgroup mapping, fused epilogue).
#pragma once
#include "common.h"
#include "../../include/afk.h"

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    void* C;
    void* C2;  // optional second output: pre-activation (bf16) when AFK_GEMM_GELU is set
    const bf16* bias;
    const bf16* R;
    int64_t lda, ldb, ldc, ldr;
    int M, N, K;
    int ntm, ntn;
    int flags;
    int res_mod;  // residual row = m % res_mod when > 0 (broadcast table, e.g. embed_positions)
    float alpha;
    int gm;  // M-tiles per rasterization group (L2 locality)
    int splits;  // split-K (128x128 kernel only): blockIdx.y = split, raw fp32 partial sums go to ws[split][M][N]
    float* ws;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// workgroup -> tile: XCD-contiguous (bijective remap of the round-robin dispatch) then grouped along M so that
// the tiles resident on one XCD share A and B panels through that XCD's private L2.
__device__ __forceinline__ void gemm_tile_of_block(const GemmArgs& p, int& tm, int& tn) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int swz = (p.gm & 0x80) ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bit 7 of gm: raw dispatch order (experiment)
    const int GM = (p.gm & 0x7f) > 0 ? (p.gm & 0x7f) : 4;
    const int per_group = GM * p.ntn;
    const int g = swz / per_group, rem = swz - g * per_group;
    const int first_m = g * GM;
    const int gsz = min(p.ntm - first_m, GM);
    tm = first_m + rem % gsz;
    tn = rem / gsz;
}

// epilogue for 4 consecutive n of one row m (values already hold the fp32 accumulators)
// order: *alpha, +bias[n] -> (round bf16, write preact, GELU-erf) -> (round bf16, +residual) -> (+C if ACCUM) -> store
__device__ __forceinline__ void gemm_epilogue_store4(const GemmArgs& p, int m, int n, float v[4]) {
    const int flags = p.flags;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
    if (flags & AFK_GEMM_BIAS) {
        const bf16x4 bv = *(const bf16x4*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
    }
    if (flags & AFK_GEMM_GELU) {
        // oracle applies GELU to the bf16-rounded Linear output
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
        if (p.C2) {
            bf16x4 pre;
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[e] = (bf16)v[e];
            *(bf16x4*)((bf16*)p.C2 + (int64_t)m * p.ldc + n) = pre;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
    }
    if (flags & AFK_GEMM_SWIGLU_BWD) {
        // v = d(silu(gate)*up) for columns n..n+3 of the SwiGLU output (this GEMM is the down-projection dgrad).  R = the saved
        // [rows, 2N] gate|up pre-activations; the two gradients go to C[m, n] (gate) and C[m, N + n] (up): the [rows, N] intermediate
        // never exists in HBM.  Same arithmetic as silu_mul_bwd_kernel on the bf16-rounded GEMM result (bit-identical to the
        // two-kernel form).
        const bf16x4 gv = *(const bf16x4*)(p.R + (int64_t)m * p.ldr + n);
        const bf16x4 uv = *(const bf16x4*)(p.R + (int64_t)m * p.ldr + p.N + n);
        bf16x4 og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gf = (float)gv[e], uf = (float)uv[e], df = rbf(v[e]);
            const float s = sigmoid_f(gf);
            og[e] = (bf16)(df * uf * (s * (1.f + gf * (1.f - s))));
            ou[e] = (bf16)(df * (gf * s));
        }
        bf16* cp = (bf16*)p.C + (int64_t)m * p.ldc + n;
        *(bf16x4*)cp = og;
        *(bf16x4*)(cp + p.N) = ou;
        return;
    }
    if (flags & AFK_GEMM_RESIDUAL) {
        const int rm = p.res_mod > 0 ? m % p.res_mod : m;
        const bf16x4 rv = *(const bf16x4*)(p.R + (int64_t)rm * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) + (float)rv[e];
    }
    if (flags & AFK_GEMM_OUT_F32) {
        float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
        f32x4 o;
        if (flags & AFK_GEMM_ACCUM) {
            o = *(const f32x4*)cp;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
        }
        *(f32x4*)cp = o;
    } else {
        bf16* cp = (bf16*)p.C + (int64_t)m * p.ldc + n;
        if (flags & AFK_GEMM_ACCUM) {
            const bf16x4 old = *(const bf16x4*)cp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)old[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *(bf16x4*)cp = o;
    }
}

// 256x256 ping-pong kernel (gemm256.hip)
int afk_launch_gemm256(const GemmArgs& p, hipStream_t st);
// transposed-operand variants (gemm256t.hip): NN (trans_a = 0) and TN (trans_a = 1); B is reduction-major in both
int afk_launch_gemm256t(const GemmArgs& p, int trans_a, hipStream_t st);
