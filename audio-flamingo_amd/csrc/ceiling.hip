// Measured ceiling of the matrix pipe under the socket power cap (round 6).
//
// The 256x256 GEMM sits at ~0.49 of the 2.5 PFLOP/s bf16 datasheet peak and the step runs power-limited (1.34-1.37 kW, 1.35-1.7 GHz instead of
// 2.4 GHz).  The datasheet peak is then not what the kernel can be held against: this file measures what the chip sustains when NOTHING but
// MFMAs run - v_mfma_f32_32x32x16_bf16 on N(0,1) operands, 8 waves per CU (two per SIMD, as the GEMM), operands resident in registers, no LDS and
// no global traffic inside the loop (mode 0) - and the same loop with the GEMM's LDS fragment reads added (mode 1: 24 ds_read_b128 per 32 MFMAs,
// the 256x256x64 ping-pong kernel's ratio, gemm256.hip).  bench.py divides the GEMM's achieved rate by the mode-0 figure
// (`roofline.frac_of_power_ceiling`); tools/measure_mfma_ceiling.py samples clock and socket power beside it -> profiles/r06_mfma_power_ceiling.md.
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr int SEG = 16;   // MFMAs per segment: 2 x 2 output tiles x 4 k-steps (one MFMA segment of the ping-pong GEMM)

template <int MODE>
__global__ __launch_bounds__(512, 1) void mfma_ceiling_kernel(const bf16* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // operands: 2 A rows-blocks x 4 k-steps and 2 B column-blocks x 4 k-steps, 8 bf16 per lane each = the 64 operand VGPRs of a GEMM segment
    bf16x8 a[2][4], b[2][4];
    const bf16x8* s8 = (const bf16x8*)src;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[i][k] = s8[((blockIdx.x * 8 + wave) * 16 + i * 4 + k) * 64 % 8192 + lane];
            b[i][k] = s8[((blockIdx.x * 8 + wave) * 16 + 8 + i * 4 + k) * 64 % 8192 + lane];
        }
    if (MODE == 1) {   // 64 KiB of N(0,1) bf16 into LDS: the fragment reads below walk it
        for (int i = tid; i < 4096; i += 512) ((bf16x8*)smem)[i] = s8[i];
        __syncthreads();
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint32_t lds = afk_lds_addr(smem) + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            // 12 fragment reads per 16 MFMAs (24 per 32: 8 B + 16 A fragments per K-tile per wave); fresh data every segment
            const uint32_t base = lds + ((it * 12 * 1024) & 0xffff & ~0x3fff);
            afk_static_for<4>([&](auto K) {
                constexpr int k = decltype(K)::value;
                a[0][k] = afk_lds_b128<k * 1024>(base);
                a[1][k] = afk_lds_b128<(4 + k) * 1024>(base);
                b[k & 1][k] = afk_lds_b128<(8 + k) * 1024>(base);
            });
            afk_lds_wait0(a[0][0], a[0][1], a[0][2], a[0][3], a[1][0], a[1][1], a[1][2], a[1][3], b[0][0], b[1][1], b[0][2], b[1][3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 123.456f) sink[0] = s;   // keeps the accumulators alive; never true in practice
}

// modes 2 / 3 (round 6): issue pacing of one wave per SIMD - 16 MFMAs per trip over four accumulators, round-robin (2: every MFMA depends on the one
// four back) or each accumulator four times in a row (3: every MFMA depends on its predecessor, the P.V order of the attention kernels of rounds 1-5).
// sink[0] = shader cycles per MFMA (s_memtime of wave 0, block 0).
template <bool CHAIN>
__global__ __launch_bounds__(256, 1) void mfma_pacing_kernel(const bf16* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* s8 = (const bf16x8*)src;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = s8[(wave * 8 + k) * 64 + lane];
        b[k] = s8[(wave * 8 + 4 + k) * 64 + lane];
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (CHAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) sink[0] = (float)(t1 - t0) / (16.f * iters);
    if (s == 123.456f) sink[1] = s;
}

}  // namespace

extern "C" int afk_mfma_ceiling(int mode, int nblocks, int iters, const void* operands, float* sink, double* host_flops, void* stream) {
    AFK_REQUIRE(mode >= 0 && mode <= 3, "afk_mfma_ceiling: mode %d (0 = register-resident operands, 1 = + LDS fragment reads, 2 / 3 = issue pacing of one wave per SIMD, "
                "independent / dependent accumulators: sink[0] = cycles per MFMA)", mode);
    AFK_REQUIRE(nblocks > 0 && iters > 0 && operands && sink, "afk_mfma_ceiling: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (mode >= 2) {
        if (mode == 2) mfma_pacing_kernel<false><<<nblocks, 256, 0, s>>>((const bf16*)operands, sink, iters);
        else mfma_pacing_kernel<true><<<nblocks, 256, 0, s>>>((const bf16*)operands, sink, iters);
        AFK_LAUNCH_CHECK("afk_mfma_ceiling");
        if (host_flops) *host_flops = (double)nblocks * 4.0 * iters * 16 * 2.0 * 32 * 32 * 16;
        return AFK_OK;
    }
    if (mode == 0) {
        mfma_ceiling_kernel<0><<<nblocks, 512, 0, s>>>((const bf16*)operands, sink, iters);
    } else {
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute((const void*)mfma_ceiling_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384);
            attr = true;
        }
        mfma_ceiling_kernel<1><<<nblocks, 512, 65536 + 16384, s>>>((const bf16*)operands, sink, iters);
    }
    AFK_LAUNCH_CHECK("afk_mfma_ceiling");
    if (host_flops) *host_flops = (double)nblocks * 8.0 * iters * SEG * 2.0 * 32 * 32 * 16;
    return AFK_OK;
}
