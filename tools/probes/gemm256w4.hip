#ifdef AFK_PROBES  // rejected schedule, kept for the probe tables of profiles/r02_gemm_probes.md: not part of the default libafk.so (make PROBES=1)
// bf16 NT GEMM, 256x256x64 tile, FOUR waves of 128x128 each - one wave per SIMD, accumulators in AGPRs (round 2).
//
// Why a second 256x256 kernel.  gemm256.hip (8 waves, two ping-pong groups of 128x64 per wave) keeps each SIMD's matrix pipe 70 % busy:
// its four segments per K-tile are 512 MFMA-cycles each and every one ends in a workgroup barrier, so ~200 cycles of barrier / wait /
// segment start-up are paid per 512 cycles of work, and the chip runs power-limited at that (PMC: profiles/r02_gemm_pmc.md).  This
// kernel removes the structure that needs those barriers:
//   * 128x128 per wave (4x4 MFMA tiles of 32x32x16, 256 accumulator registers -> AGPRs; 512-register budget at one wave per SIMD).
//     Fragment traffic per MFMA drops by a third (8 ds_read_b128 per 16 MFMAs instead of 12): less LDS power per flop.
//   * no partner wave: the wave hides its own memory work in the shadow of its own MFMAs (an MFMA occupies the pipe for 32 cycles, the
//     wave needs ~4 to issue it: up to 5 single-issue instructions fit per gap, MI355X guide "one wave per SIMD").  Per k-step of 16
//     MFMAs the stream carries 8 ds_read_b128 (next k-step's fragments, register double buffer) and at most 8 LDS-DMA pieces.
//   * ONE barrier per K-tile (2048 MFMA-cycles) instead of four.
//
// LDS: a ring of FIVE 32 KiB operand images (all 160 KiB of the CU): image 2t = A rows of K-tile t, image 2t+1 = B rows, image i lives in
// slot i % 5 (same XOR-swizzled 128-byte rows as gemm256.hip).  Per wave and K-tile: 16 LDS-DMA pieces (8 per image), ONE per four MFMAs:
//     k-step 0 : MFMA(t,0)  | read (t,1)   | pieces 4..7 of B(t+1)
//     k-step 1 : MFMA(t,1)  | read (t,2)   | pieces 0..3 of A(t+2)
//     k-step 2 : MFMA(t,2)  | read (t,3)   | pieces 4..7 of A(t+2)
//     --- lgkmcnt(0) (my reads of tile t are back) ; vmcnt(8) (all but the 8 pieces of A(t+2) have landed: tile t+1 is complete) ; X(t) ---
//     k-step 3 : MFMA(t,3)  | read (t+1,0) | pieces 0..3 of B(t+2)
// RAW: tile t+1 = A(t+1) (issued during tile t-1) + B(t+1) (issued right after X(t-1) .. k-step 0 of t); each wave waits for its own
// pieces before X(t); the first read of tile t+1 comes after X(t).  WAR: A(t+2) and B(t+1) reuse the slots of B(t-1) / A(t-1) (free
// since X(t-1)), B(t+2) reuses the slot of A(t) (free since X(t)).  Landing slack: B pieces >= 1024 MFMA-cycles, A pieces >= 2048.
// The spread (4 pieces per k-step instead of 8 in two of them) matters: an LDS-DMA instruction blocks its wave's issue for ~60 cycles
// (r02 probe), and a lone wave has only the 64 cycles of its two queued MFMAs to hide in.
// Past the last tile the prefetch index is clamped (redundant loads into dead slots): the loop body is branch-free.
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = 128;              // bytes per LDS row
constexpr int IMG_BYTES = 256 * ROWB;  // 32 KiB per operand image
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * IMG_BYTES;  // 160 KiB

#define W4_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define W4_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W4_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
// scheduling-group masks (llvm.amdgcn.sched.group.barrier): 0x008 MFMA, 0x100 DS read, 0x020 VMEM read
#define W4_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int OFF>
__device__ __forceinline__ bf16x8 w4_lds_read(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// JT = 32-column MFMA tiles per wave along N:
//   JT = 4: FOUR waves of 128x128 (one per SIMD, 256 accumulators in AGPRs, 512-register budget)                       "w4"
//   JT = 2: EIGHT waves of 128x64 (two per SIMD, free-running: no ping-pong barriers - when one wave of a SIMD is blocked in an
//           LDS-DMA issue or a counted wait, its sibling's MFMAs own the matrix pipe), same ring, same single barrier per K-tile  "w8f"
// MODE 0: the kernel.  MODE 1 / 2 / 3: timing probes (wrong results): no LDS-DMA in the loop / neither LDS-DMA nor fragment reads /
// LDS-DMA issued but never waited for (separates the issue cost of the DMA instructions from the time spent waiting for them to land).
template <int JT, int MODE>
__global__ __launch_bounds__(JT == 4 ? 256 : 512, JT == 4 ? 1 : 2) void gemm_nt_bf16_w4(GemmArgs p) {
    constexpr int NWAVE = 16 / JT;   // 4 or 8
    constexpr int PI = 32 / NWAVE;   // LDS-DMA pieces per operand image and wave: 8 or 4
    constexpr int NG = 2 * JT;       // groups of 2 MFMAs per k-step: 8 or 4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = JT == 4 ? wave >> 1 : wave >> 2, wn = JT == 4 ? wave & 1 : wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA sources: unit u of an operand image = rows [8u, 8u+8) = 1 KiB; piece j of wave w stages unit w + NWAVE*j (j < PI) of an image
    const bf16* srcA[PI];
    const bf16* srcB[PI];
    {
        const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const int rl = (wave + NWAVE * j) * 8 + lrow;
            const int chunk = pos ^ ((rl >> 1) & 7);
            srcA[j] = p.A + (int64_t)min(m0 + rl, p.M - 1) * p.lda + chunk * 8;
            srcB[j] = p.B + (int64_t)min(n0 + rl, p.N - 1) * p.ldb + chunk * 8;
        }
    }
    const int dst0 = wave * 1024;  // + j*PSTEP within an image
    constexpr int PSTEP = NWAVE * 1024;

    // ---- fragment offsets (bytes within an operand image); the swizzle term is lane-constant
    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;  // + i*32*ROWB, i = 0..3   (within the A image)
    const int b_row0 = (wn * 32 * JT + l31) * ROWB;  // + j*32*ROWB, j < JT   (within the B image)

    f32x16 acc[4][JT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int T = p.K / BK;
    // ---- prologue: A0 B0 A1 + pieces 0..3 of B1 (slots 0 1 2 3)
    {
        const int64_t k1 = (int64_t)min(1, T - 1) * BK;
#pragma unroll
        for (int j = 0; j < PI; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)srcA[j], (lds_void*)(smem + dst0 + j * PSTEP), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < PI; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)srcB[j], (lds_void*)(smem + IMG_BYTES + dst0 + j * PSTEP), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < PI; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(srcA[j] + k1), (lds_void*)(smem + 2 * IMG_BYTES + dst0 + j * PSTEP), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < PI / 2; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(srcB[j] + k1), (lds_void*)(smem + 3 * IMG_BYTES + dst0 + j * PSTEP), 16, 0, 0);
    }
    if constexpr (JT == 4) W4_VMCNT(12); else W4_VMCNT(6);   // A0 B0 landed; A1 and half of B1 may still fly
    W4_BARRIER();

    bf16x8 fa[2][4], fb[2][JT];  // [k-step parity][row tile]
    // One k-step, written out in issue order: 8 groups of { 2 MFMAs ; 1 ds_read_b128 of the NEXT k-step's fragments } and one LDS-DMA piece
    // behind every second group, each pinned by scheduling fences.  Fragment reads are OPAQUE asm ds_read_b128 with hand-counted lgkmcnt
    // waits: left to the compiler, every k-step opened with s_waitcnt lgkmcnt(0), i.e. waited for the read issued one instruction earlier
    // (ISA inspected).  LDS returns in order, so with the fetch order B0 A0 B1 B2 B3 A1 A2 A3 of the previous k-step and one new read
    // issued per group of this one, group g may start when at most WAIT[g] reads are outstanding:
    //     g0 (needs B0 A0 B1) 5 | g1 (B2 B3) 4 | g2 (A1) 4 | g4 (A2) 5 | g6 (A3) 6 ;  every fragment has >= 5 groups (320 MFMA-cycles) of lead.
    //   CUR: fragment register set in use (the other is being filled);  ra / rb: LDS addresses (A / B image + lane offset + k-step swizzle)
    //   the reads come from;  dsrc / dko / dbase: the 4 DMA pieces of this k-step = dsrc[DJ0 .. DJ0+3] + dko -> LDS dbase + dst0 + j*4096.
    //   JT = 2 (w8f): 8 MFMAs, 6 reads (fetch order B0 B1 A0 A1 A2 A3, issued 2 2 1 1 over the four groups), 2 DMA pieces (groups 2, 3);
    //                 waits g0 (B0 B1 A0) 3 | g1 (A1) 4 | g2 (A2) 5 | g3 (A3) 5.
    auto kstep = [&](auto cur_, uint32_t ra, uint32_t rb, const bf16* const (&dsrc)[PI], auto dj0_, int64_t dko, char* dbase) {
        constexpr int CUR = decltype(cur_)::value, NXT = CUR ^ 1, DJ0 = decltype(dj0_)::value;
        afk_static_for<NG>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            constexpr int n0_ = 2 * g, n1_ = 2 * g + 1;
            if constexpr (MODE < 2 && JT == 4) {
                if constexpr (g == 0) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fb[CUR][0], fa[CUR][0], fb[CUR][1]); }
                if constexpr (g == 1) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fb[CUR][2], fb[CUR][3]); }
                if constexpr (g == 2) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fa[CUR][1]); }
                if constexpr (g == 4) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][2]); }
                if constexpr (g == 6) { asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); afk_lds_tie(fa[CUR][3]); }
            }
            if constexpr (MODE < 2 && JT == 2) {
                if constexpr (g == 0) { asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); afk_lds_tie(fb[CUR][0], fb[CUR][1], fa[CUR][0]); }
                if constexpr (g == 1) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fa[CUR][1]); }
                if constexpr (g == 2) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][2]); }
                if constexpr (g == 3) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][3]); }
            }
            acc[n0_ / JT][n0_ % JT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][n0_ % JT], fa[CUR][n0_ / JT], acc[n0_ / JT][n0_ % JT], 0, 0, 0);
            acc[n1_ / JT][n1_ % JT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][n1_ % JT], fa[CUR][n1_ / JT], acc[n1_ / JT][n1_ % JT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE < 2 && JT == 4) {  // fetch order B0 A0 B1 B2 B3 A1 A2 A3
                if constexpr (g == 0) fb[NXT][0] = w4_lds_read<0>(rb);
                if constexpr (g == 1) fa[NXT][0] = w4_lds_read<0>(ra);
                if constexpr (g == 2) fb[NXT][1] = w4_lds_read<1 * 32 * ROWB>(rb);
                if constexpr (g == 3) fb[NXT][2] = w4_lds_read<2 * 32 * ROWB>(rb);
                if constexpr (g == 4) fb[NXT][3] = w4_lds_read<3 * 32 * ROWB>(rb);
                if constexpr (g == 5) fa[NXT][1] = w4_lds_read<1 * 32 * ROWB>(ra);
                if constexpr (g == 6) fa[NXT][2] = w4_lds_read<2 * 32 * ROWB>(ra);
                if constexpr (g == 7) fa[NXT][3] = w4_lds_read<3 * 32 * ROWB>(ra);
            }
            if constexpr (MODE < 2 && JT == 2) {  // fetch order B0 B1 | A0 A1 | A2 | A3
                if constexpr (g == 0) { fb[NXT][0] = w4_lds_read<0>(rb); fb[NXT][1] = w4_lds_read<1 * 32 * ROWB>(rb); }
                if constexpr (g == 1) { fa[NXT][0] = w4_lds_read<0>(ra); fa[NXT][1] = w4_lds_read<1 * 32 * ROWB>(ra); }
                if constexpr (g == 2) fa[NXT][2] = w4_lds_read<2 * 32 * ROWB>(ra);
                if constexpr (g == 3) fa[NXT][3] = w4_lds_read<3 * 32 * ROWB>(ra);
            }
            constexpr bool dma_here = JT == 4 ? (g & 1) == 1 : g >= 2;
            constexpr int dj = DJ0 + (JT == 4 ? (g >> 1) : g - 2);
            if constexpr ((MODE == 0 || MODE == 3) && dma_here)
                __builtin_amdgcn_global_load_lds((gbl_void*)(dsrc[dj] + dko), (lds_void*)(dbase + dst0 + dj * PSTEP), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IH = std::integral_constant<int, PI / 2>;   // second half of an image's pieces

    // fragments of (tile 0, k-step 0), in the fetch order the wait ladder assumes
    const uint32_t lds0 = afk_lds_addr(smem);
    {
        const uint32_t ra = lds0 + a_row0 + koffb[0], rb = lds0 + IMG_BYTES + b_row0 + koffb[0];
        if constexpr (JT == 4) {
            fb[0][0] = w4_lds_read<0>(rb);
            fa[0][0] = w4_lds_read<0>(ra);
            fb[0][1] = w4_lds_read<1 * 32 * ROWB>(rb);
            fb[0][2] = w4_lds_read<2 * 32 * ROWB>(rb);
            fb[0][3] = w4_lds_read<3 * 32 * ROWB>(rb);
        } else {
            fb[0][0] = w4_lds_read<0>(rb);
            fb[0][1] = w4_lds_read<1 * 32 * ROWB>(rb);
            fa[0][0] = w4_lds_read<0>(ra);
        }
        fa[0][1] = w4_lds_read<1 * 32 * ROWB>(ra);
        fa[0][2] = w4_lds_read<2 * 32 * ROWB>(ra);
        fa[0][3] = w4_lds_read<3 * 32 * ROWB>(ra);
    }
    if constexpr (JT == 4) __builtin_amdgcn_s_setprio(1);
    // ring positions (scalar): image 2t (A of tile t) sits in slot sa, B of tile t in sa+1, ... all modulo 5
    int sa = 0;
    for (int t = 0; t < T; ++t) {
        const int s_a0 = sa;                                   // A(t)
        const int s_b0 = sa + 1 >= NSLOT ? sa + 1 - NSLOT : sa + 1;   // B(t)
        const int s_a1 = sa + 2 >= NSLOT ? sa + 2 - NSLOT : sa + 2;   // A(t+1)
        const int s_b1 = sa + 3 >= NSLOT ? sa + 3 - NSLOT : sa + 3;   // B(t+1)
        const int s_a2 = sa + 4 >= NSLOT ? sa + 4 - NSLOT : sa + 4;   // A(t+2)
        const int s_b2 = sa;                                   // B(t+2): image 2t+5 -> the slot of A(t)
        const int64_t o1 = (int64_t)min(t + 1, T - 1) * BK, o2 = (int64_t)min(t + 2, T - 1) * BK;
        const uint32_t ra0 = lds0 + s_a0 * IMG_BYTES + a_row0, rb0 = lds0 + s_b0 * IMG_BYTES + b_row0;
        const uint32_t ra1 = lds0 + s_a1 * IMG_BYTES + a_row0, rb1 = lds0 + s_b1 * IMG_BYTES + b_row0;
        kstep(I0{}, ra0 + koffb[1], rb0 + koffb[1], srcB, IH{}, o1, smem + s_b1 * IMG_BYTES);   // MFMA(t,0) | read (t,1) | B(t+1) second half
        kstep(I1{}, ra0 + koffb[2], rb0 + koffb[2], srcA, I0{}, o2, smem + s_a2 * IMG_BYTES);   // MFMA(t,1) | read (t,2) | A(t+2) first half
        kstep(I0{}, ra0 + koffb[3], rb0 + koffb[3], srcA, IH{}, o2, smem + s_a2 * IMG_BYTES);   // MFMA(t,2) | read (t,3) | A(t+2) second half
        W4_LGKMCNT0();
        if constexpr (MODE == 3) {
        } else if constexpr (MODE != 0) W4_VMCNT(0); else if constexpr (JT == 4) W4_VMCNT(8); else W4_VMCNT(4);
        W4_BARRIER();                                                                             // X(t)
        kstep(I1{}, ra1 + koffb[0], rb1 + koffb[0], srcB, I0{}, o2, smem + s_b2 * IMG_BYTES);   // MFMA(t,3) | read (t+1,0) | B(t+2) pieces 0..3
        sa = s_a1;
    }
    if constexpr (JT == 4) __builtin_amdgcn_s_setprio(0);
    W4_LGKMCNT0();  // the last k-step fetched fragments of a tile that does not exist: retire them before the registers are reused
    W4_VMCNT(0);    // no LDS-DMA may be in flight when the workgroup releases its LDS

    // ---- epilogue: lane holds row m = ..+l31 and n = ..+8q+4hi+{0..3} of each 32x32 block
    // (one 32x32 block at a time, fenced: left alone the compiler hoists all accumulator reads to the top and spills ~160 VGPRs)
    afk_static_for<4 * JT>([&](auto ij_) {
        constexpr int i = decltype(ij_)::value / JT, j = decltype(ij_)::value % JT;
        gemm_store_block32(p, m0 + wm * 128 + i * 32 + l31, n0 + wn * 32 * JT + j * 32, hi, acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

}  // namespace

// mode: 0 = w4, 1 / 2 = its timing probes; 3 = w8f (8 free-running waves), 4 / 5 = its probes
int afk_launch_gemm256w4(const GemmArgs& p, int mode, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        const void* ks[7] = {(const void*)gemm_nt_bf16_w4<4, 0>, (const void*)gemm_nt_bf16_w4<4, 1>, (const void*)gemm_nt_bf16_w4<4, 2>,
                             (const void*)gemm_nt_bf16_w4<2, 0>, (const void*)gemm_nt_bf16_w4<2, 1>, (const void*)gemm_nt_bf16_w4<2, 2>,
                             (const void*)gemm_nt_bf16_w4<2, 3>};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
                return afk_set_error(AFK_ERR_LAUNCH, "gemm256w4: cannot reserve %d bytes of LDS", LDS_BYTES);
        attr_set = true;
    }
    const dim3 grid((unsigned)((int64_t)p.ntm * p.ntn));
    switch (mode) {
        case 1: hipLaunchKernelGGL((gemm_nt_bf16_w4<4, 1>), grid, dim3(256), LDS_BYTES, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_nt_bf16_w4<4, 2>), grid, dim3(256), LDS_BYTES, st, p); break;
        case 3: hipLaunchKernelGGL((gemm_nt_bf16_w4<2, 0>), grid, dim3(512), LDS_BYTES, st, p); break;
        case 4: hipLaunchKernelGGL((gemm_nt_bf16_w4<2, 1>), grid, dim3(512), LDS_BYTES, st, p); break;
        case 5: hipLaunchKernelGGL((gemm_nt_bf16_w4<2, 2>), grid, dim3(512), LDS_BYTES, st, p); break;
        case 6: hipLaunchKernelGGL((gemm_nt_bf16_w4<2, 3>), grid, dim3(512), LDS_BYTES, st, p); break;
        default: hipLaunchKernelGGL((gemm_nt_bf16_w4<4, 0>), grid, dim3(256), LDS_BYTES, st, p); break;
    }
    return AFK_OK;
}

#endif  // AFK_PROBES
