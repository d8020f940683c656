#ifdef AFK_PROBES  // round-4 probe (variant 14): persistent 256x256 ping-pong GEMM with the NEXT tile's prologue issued BEFORE the epilogue stores
// DESIGN.md §8 item 4 (round 3) / VERDICT r03 item 3a: the round-2 persistent loop (gemm256p.hip, variant 13) lost 7-10 us per tile boundary because
// the next tile's X0 Y0 X1 LDS-DMA pieces were issued BEHIND the epilogue stores and every counted wait of the first K-tile then waited for the
// stores as well.  This variant issues X0' Y0' X1' right after the K loop's final barrier (both LDS buffers are dead by then), THEN the 16 store
// instructions of the epilogue, and raises the waits of the first K-tile by those 16: queue = X0 Y0 X1 | S x 16 | Y1 | X2 ...
//     before the first barrier (X0 landed):    allowed Y0 + X1 + S      = 24
//     after MEM_a(0)  (Y0 landed):             allowed X1 + S           = 22
//     after MFMA_a(0) (Y1 issued):             allowed X1 + S + Y1      = 24
//     after MEM_b(0)  (X1 landed):             allowed S + Y1           = 18
//     after MFMA_b(0) (X2 issued):             allowed S + Y1 + X2      = 24
// From K-tile 1 on the standard ladder (6 / 8 / 2 / 8) applies: Y1 is younger than the stores, so its wait also waits for them.
// ASSUMPTION under test: vector-memory operations of one wave retire IN ORDER across loads and stores (if stores could retire ahead of older
// loads, vmcnt(24) would not prove that X0 has landed) - the bit-identity test against the one-tile-per-workgroup kernel is the check.
// Interior tiles only issue exactly 16 stores per wave; a ragged tile (fewer stores) drains with vmcnt(0) behind its epilogue, and so does the
// first tile of a workgroup behind its prologue (no stores in the queue yet).  Plain epilogue (flags 0) only: this is a measurement, not a product path.
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = 128;
constexpr int OP_BYTES = 256 * ROWB;
constexpr int BUF_BYTES = 2 * OP_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;

#define AFK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AFK_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AFK_BARRIER()                         \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_k256q(GemmArgs p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;

    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;
    const int b_row0 = OP_BYTES + (wn * 64 + l31) * ROWB;
    const int T = p_in.K / BK;   // >= 2 (launcher)
    const int ntiles = p_in.ntm * p_in.ntn;

    const bf16* xsrc[6];
    int xdst[6];
    const bf16* ysrc[2];
    int ydst[2];
    // LDS-DMA sources of tile (m0, n0): same unit lists as gemm256.hip
    auto sources = [&](int m0, int n0) {
        const GemmArgs* kp = (const GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const GemmArgs& p = *kp;
        const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int i = wave + 8 * j;
            const bool isB = i < 32;
            const int unit = isB ? i : ((i - 32) < 8 ? (i - 32) : (i - 32) + 8);
            const int rl = unit * 8 + lrow;
            const int chunk = pos ^ ((rl >> 1) & 7);
            if (isB) {
                const int r = min(n0 + rl, p.N - 1);
                xsrc[j] = p.B + (int64_t)r * p.ldb + chunk * 8;
            } else {
                const int r = min(m0 + rl, p.M - 1);
                xsrc[j] = p.A + (int64_t)r * p.lda + chunk * 8;
            }
            xdst[j] = (isB ? OP_BYTES : 0) + unit * 1024;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int jj = wave + 8 * j;
            const int unit = jj < 8 ? 8 + jj : 16 + jj;
            const int rl = unit * 8 + lrow;
            const int chunk = pos ^ ((rl >> 1) & 7);
            const int r = min(m0 + rl, p.M - 1);
            ysrc[j] = p.A + (int64_t)r * p.lda + chunk * 8;
            ydst[j] = unit * 1024;
        }
    };
    auto issue_prologue = [&]() {   // X0 Y0 X1 of the tile whose sources are loaded
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[j]), (lds_void*)(smem + xdst[j]), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)(ysrc[j]), (lds_void*)(smem + ydst[j]), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[j] + BK), (lds_void*)(smem + BUF_BYTES + xdst[j]), 16, 0, 0);
    };

    int tile = blockIdx.x;
    int tm, tn;
    gemm_tile_of(p_in, tile, ntiles, tm, tn);
    int m0 = tm * BM, n0 = tn * BN;
    sources(m0, n0);
    issue_prologue();
    AFK_VMCNT(0);   // first tile of the workgroup: no stores in the queue, the raised waits below would prove nothing - drain once

    f32x16 acc[4][2];
    bf16x8 bf[2][4], af[2][4];
#define AFK_MFMA4(ACC0, s)                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)          \
            acc[ACC0 + i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j_][s], af[i_][s], acc[ACC0 + i_][j_], 0, 0, 0); \
    } while (0)
    // one K-tile; FIRST = K-tile 0 of a tile (waits raised by the 16 epilogue stores that sit between X1 and Y1 in the queue)
    auto ktile = [&](int t, auto first_) {
        constexpr bool FIRST = decltype(first_)::value;
        const char* buf = smem + (t & 1) * BUF_BYTES;
        const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
        const int e1 = (t + 1) & 1, e2 = t & 1;
        const int64_t oy = (int64_t)t1 * BK, ox = (int64_t)t2 * BK;
        char* by = smem + e1 * BUF_BYTES;
        char* bx = smem + e2 * BUF_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = *(const bf16x8*)(buf + b_row0 + j * 32 * ROWB + koffb[s]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + i * 32 * ROWB + koffb[s]);
        AFK_LGKMCNT0();
        if constexpr (FIRST) AFK_VMCNT(22); else AFK_VMCNT(6);
        AFK_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        AFK_MFMA4(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(ysrc[0] + oy), (lds_void*)(by + ydst[0]), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        AFK_MFMA4(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(ysrc[1] + oy), (lds_void*)(by + ydst[1]), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        AFK_MFMA4(0, 2);
        AFK_MFMA4(0, 3);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (FIRST) AFK_VMCNT(24); else AFK_VMCNT(8);
        AFK_BARRIER();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + (i + 2) * 32 * ROWB + koffb[s]);
        AFK_LGKMCNT0();
        if constexpr (FIRST) AFK_VMCNT(18); else AFK_VMCNT(2);
        AFK_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0][s], af[i][s], acc[2 + i][0], 0, 0, 0);
                acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1][s], af[i][s], acc[2 + i][1], 0, 0, 0);
                const int piece = 2 * s + i;
                if (piece < 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[piece] + ox), (lds_void*)(bx + xdst[piece]), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (FIRST) AFK_VMCNT(24); else AFK_VMCNT(8);
        AFK_BARRIER();
    };

    while (true) {
        AFK_VMCNT(24);   // X0 of this tile has landed (first tile: everything has)
        AFK_BARRIER();
        if (wm == 1) AFK_BARRIER();  // group 1 runs one segment behind group 0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        ktile(0, std::true_type{});
        for (int t = 1; t < T; ++t) ktile(t, std::false_type{});
        AFK_VMCNT(0);                // the clamped (dead) prefetches of the tail have landed before the same units are staged again
        if (wm == 0) AFK_BARRIER();  // equalise barrier counts: every fragment read of this tile is retired on both groups

        const int em0 = m0, en0 = n0;
        tile += gridDim.x;
        const bool more = tile < ntiles;
        const GemmArgs* kp = (const GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const GemmArgs& pe = *kp;
        if (more) {
            gemm_tile_of(pe, tile, ntiles, tm, tn);
            m0 = tm * BM;
            n0 = tn * BN;
            sources(m0, n0);
            issue_prologue();        // 14 LDS-DMA pieces per wave, AHEAD of the stores
        }
        afk_static_for<8>([&](auto ij_) {
            constexpr int i = decltype(ij_)::value >> 1, j = decltype(ij_)::value & 1;
            gemm_store_block32_body<0>(pe, em0 + wm * 128 + i * 32 + l31, en0 + wn * 64 + j * 32, hi, acc[i][j]);
        });
        if (!more) break;
        if (em0 + BM > pe.M || en0 + BN > pe.N) AFK_VMCNT(0);   // ragged tile: fewer than 16 stores may have been issued - the raised waits need exactly 16
    }
#undef AFK_MFMA4
}

}  // namespace

int afk_launch_gemm256q(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    static int ncu = 256;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_bf16_k256q, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "gemm256q: cannot reserve %d bytes of LDS", LDS_BYTES);
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        attr_set = true;
    }
    if (p.flags != 0 || !p.wide || p.K < 2 * BK || p.splits > 1)
        return afk_set_error(AFK_ERR_UNSUPPORTED, "gemm256q (probe variant 14): plain bf16 epilogue, 16-byte stores, K >= 128, no split-K only");
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    hipLaunchKernelGGL(gemm_nt_bf16_k256q, dim3((unsigned)(nwg < ncu ? nwg : ncu)), dim3(512), LDS_BYTES, st, p);
    return AFK_OK;
}

#endif  // AFK_PROBES
