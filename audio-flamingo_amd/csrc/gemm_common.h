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
    int splits;  // split-K: blockIdx.y = split, raw fp32 partial sums go to ws[split][M][N]
    float* ws;
    int wide;    // 16-byte epilogue accesses are legal (N % 8 == 0 and every epilogue pointer / leading dimension 16-byte aligned)
    // AFK_GEMM_ROPE (afk_gemm_nt_bf16_rope): rotate-half RoPE on output columns [0, rope_cols), heads of 128 columns
    const bf16* rope_cos;
    const bf16* rope_sin;
    const int* rope_pos;   // [M] or null (row % rope_S)
    int rope_S, rope_cols;
    const bf16* rope_cos_lanes;   // afk_rope_lanes_table(form 1) copies (nullable): element ((rb * 16 + c) * 32 + l31) * 8 + e = table[32 rb + l31][8 c + e]
    const bf16* rope_sin_lanes;
};

// Probe bits of GemmArgs::gm (bit 6 = kernel WITHOUT its epilogue: wrong results; bit 7 = raw dispatch order) and the three rejected 256x256
// schedules (gemm256w4 / gemm256f8 / gemm256p.hip, profiles/r02_gemm_probes.md) exist only in -DAFK_PROBES builds (make PROBES=1).  The
// default library compiles the branches away and afk_gemm_set_variant() rejects the values that select them.
#ifdef AFK_PROBES
#define AFK_GM_NOEPI(p) (((p).gm & 0x40) != 0)
#define AFK_GM_RAW(p) (((p).gm & 0x80) != 0)
#else
#define AFK_GM_NOEPI(p) false
#define AFK_GM_RAW(p) false
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// workgroup -> tile: XCD-contiguous (bijective remap of the round-robin dispatch) then grouped along M so that
// the tiles resident on one XCD share A and B panels through that XCD's private L2.
__device__ __forceinline__ void gemm_tile_of(const GemmArgs& p, int bid, int nwg, int& tm, int& tn) {
    const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int swz = AFK_GM_RAW(p) ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bit 7 of gm (probe builds): raw dispatch order
    const int GM = (p.gm & 0x3f) > 0 ? (p.gm & 0x3f) : 4;
    const int per_group = GM * p.ntn;
    const int g = swz / per_group, rem = swz - g * per_group;
    const int first_m = g * GM;
    const int gsz = min(p.ntm - first_m, GM);
    tm = first_m + rem % gsz;
    tn = rem / gsz;
}
__device__ __forceinline__ void gemm_tile_of_block(const GemmArgs& p, int& tm, int& tn) { gemm_tile_of(p, blockIdx.x, gridDim.x, tm, tn); }

// epilogue for W (4 or 8) consecutive n of one row m (values already hold the fp32 accumulators)
// order: *alpha, +bias[n] -> (round bf16, write preact, GELU-erf) -> (round bf16, +residual) -> (+C if ACCUM) -> store
// F >= 0: the epilogue flags are a compile-time constant (one kernel instantiation per epilogue the training step uses) - the kernel then
// carries ONLY that epilogue.  With the runtime form (F = -1) every one of the 16 store sites of a wave inlines all variants (erf GELU,
// SwiGLU backward with its divides, fp32 / accumulate forms): ~30 000 instructions, ~200 KB of code behind a 350-instruction K loop, and the
// plain path hops through it once per tile - the instruction cache (64 KB per two CUs) misses all the way (profiles/r02_gemm_probes.md §10).
template <int W, int F = -1>
__device__ __forceinline__ void gemm_epilogue_store(const GemmArgs& p, int m, int n, float* v) {
    typedef __attribute__((ext_vector_type(W))) __bf16 bvec;
    typedef __attribute__((ext_vector_type(W))) float fvec;
    const int flags = F >= 0 ? F : p.flags;
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] *= p.alpha;
    if (flags & AFK_GEMM_BIAS) {
        const bvec bv = *(const bvec*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] += (float)bv[e];
    }
    if (flags & AFK_GEMM_GELU) {
        // oracle applies GELU to the bf16-rounded Linear output
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = rbf(v[e]);
        if (p.C2) {
            bvec pre;
#pragma unroll
            for (int e = 0; e < W; ++e) pre[e] = (bf16)v[e];
            *(bvec*)((bf16*)p.C2 + (int64_t)m * p.ldc + n) = pre;
        }
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = gelu_f(v[e]);
    }
    if (flags & AFK_GEMM_SWIGLU_BWD) {
        // v = d(silu(gate)*up) for columns n.. of the SwiGLU output (this GEMM is the down-projection dgrad).  R = the saved
        // [rows, 2N] gate|up pre-activations; the two gradients go to C[m, n] (gate) and C[m, N + n] (up): the [rows, N] intermediate
        // never exists in HBM.  Same arithmetic as silu_mul_bwd_kernel on the bf16-rounded GEMM result (bit-identical to the
        // two-kernel form).
        const bvec gv = *(const bvec*)(p.R + (int64_t)m * p.ldr + n);
        const bvec uv = *(const bvec*)(p.R + (int64_t)m * p.ldr + p.N + n);
        bvec og, ou;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float gf = (float)gv[e], uf = (float)uv[e], df = rbf(v[e]);
            const float s = sigmoid_f(gf);
            og[e] = (bf16)(df * uf * (s * (1.f + gf * (1.f - s))));
            ou[e] = (bf16)(df * (gf * s));
        }
        bf16* cp = (bf16*)p.C + (int64_t)m * p.ldc + n;
        *(bvec*)cp = og;
        *(bvec*)(cp + p.N) = ou;
        return;
    }
    if (flags & AFK_GEMM_RESIDUAL) {
        const int rm = p.res_mod > 0 ? m % p.res_mod : m;
        const bvec rv = *(const bvec*)(p.R + (int64_t)rm * p.ldr + n);
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = rbf(v[e]) + (float)rv[e];
    }
    if (flags & AFK_GEMM_OUT_F32) {
        float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
        fvec o;
        if (flags & AFK_GEMM_ACCUM) {
            o = *(const fvec*)cp;
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += v[e];
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = v[e];
        }
        *(fvec*)cp = o;
    } else {
        bf16* cp = (bf16*)p.C + (int64_t)m * p.ldc + n;
        if (flags & AFK_GEMM_ACCUM) {
            const bvec old = *(const bvec*)cp;
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] += (float)old[e];
        }
        bvec o;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = (bf16)v[e];
        *(bvec*)cp = o;
    }
}
__device__ __forceinline__ void gemm_epilogue_store4(const GemmArgs& p, int m, int n, float v[4]) { gemm_epilogue_store<4, -1>(p, m, n, v); }

// One 32 (m) x 32 (n) MFMA result block of a wave: lane (l31, hi) holds row m and, in accumulator registers 4q..4q+3, the columns
// nb + 8q + 4hi + {0..3}.  Wide form (p.wide: every pointer / leading dimension of the epilogue keeps 16-byte alignment): the two lane
// halves trade registers through v_permlane32_swap so that each lane owns 8 CONSECUTIVE columns (nb + 16t + 8hi + 0..7) - 16-byte stores
// and 16-byte bias / residual loads, half as many memory instructions as the 8-byte form.  Split-K partials keep the 4-wide form.
// F >= 0 (specialised instantiation): the launcher guarantees p.wide and p.splits <= 1.  F = -2: split-K partial sums only.
template <int F>
__device__ __forceinline__ void gemm_store_block32_body(const GemmArgs& p, int m, int nb, int hi, const f32x16& acc);
// The runtime-flag form (F = -1: narrow stores, fp32 output, epilogues outside the training step) as ONE out-of-line function per kernel
// image: inlined at the 8 store sites of a 256x256 wave it made the generic instantiations of the 2-waves-per-SIMD kernels spill 730 VGPRs
// to scratch; the accumulator block travels by value in 16 VGPRs.
static __device__ __noinline__ void gemm_store_block32_generic(const GemmArgs& p, int m, int nb, int hi, f32x16 acc) {
    gemm_store_block32_body<-1>(p, m, nb, hi, acc);
}
template <int F = -1>
__device__ __forceinline__ void gemm_store_block32(const GemmArgs& p, int m, int nb, int hi, const f32x16& acc) {
    if constexpr (F == -1) gemm_store_block32_generic(p, m, nb, hi, acc);
    else gemm_store_block32_body<F>(p, m, nb, hi, acc);
}
template <int F>
__device__ __forceinline__ void gemm_store_block32_body(const GemmArgs& p, int m, int nb, int hi, const f32x16& acc) {
    if (m >= p.M) return;
    if (F == -2 || (F == -1 && p.splits > 1)) {  // F = -2: split-K launch, fp32 partial sums only
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nb + 8 * q + 4 * hi;
            if (n >= p.N) continue;
            const f32x4 o = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(p.ws + ((int64_t)blockIdx.y * p.M + m) * p.N + n) = o;
        }
        return;
    }
    if constexpr (F == -2) return;
    if (F >= 0 || p.wide) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * t + e]), __float_as_uint(acc[8 * t + 4 + e]), false, false);
                v[e] = __uint_as_float(r[0]);      // lo lanes: own q=2t ; hi lanes: partner's q=2t+1
                v[4 + e] = __uint_as_float(r[1]);  // lo lanes: partner's q=2t ; hi lanes: own q=2t+1
            }
            const int n = nb + 16 * t + 8 * hi;
            if (n < p.N) gemm_epilogue_store<8, F>(p, m, n, v);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q + 4 * hi;
        if (n >= p.N) continue;
        float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        gemm_epilogue_store<4, F>(p, m, n, v);
    }
}

// 256x256 ping-pong kernel (gemm256.hip)
int afk_launch_gemm256(const GemmArgs& p, hipStream_t st);
int afk_launch_gemm256p(const GemmArgs& p, hipStream_t st);  // persistent tile loop (tools/probes/gemm256p.hip)
int afk_launch_gemm256q(const GemmArgs& p, hipStream_t st);  // persistent tile loop, next tile's prologue ahead of the epilogue stores (tools/probes/gemm256q.hip)
// 256x256, K-step 32, eight free-running waves, ten-slot LDS ring (gemm256f8.hip); mode 1 = no-DMA timing probe
int afk_launch_gemm256f8(const GemmArgs& p, int mode, hipStream_t st);
// 256x256 four-wave kernel, 128x128 per wave, accumulators in AGPRs (gemm256w4.hip)
int afk_launch_gemm256w4(const GemmArgs& p, int mode, hipStream_t st);  // mode 1 / 2: timing probes (no DMA / no DMA + no fragment reads)
// transposed-operand variants (gemm256t.hip): NN (trans_a = 0) and TN (trans_a = 1); B is reduction-major in both
int afk_launch_gemm256t(const GemmArgs& p, int trans_a, hipStream_t st);
