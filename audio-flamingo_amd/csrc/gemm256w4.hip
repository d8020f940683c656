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
// Schedule (tile t lives in LDS buffer t&1, 64 KiB = A image | B image, same XOR-swizzled rows as gemm256.hip):
//     k-step 0 : MFMA(t,0)  | read (t,1)   | DMA pieces 8..15 of tile t+1 -> buffer (t+1)&1
//     k-step 1 : MFMA(t,1)  | read (t,2)
//     k-step 2 : MFMA(t,2)  | read (t,3)
//     --- lgkmcnt(0) (my reads of buffer t&1 are back) ; vmcnt(0) (my 16 pieces of tile t+1 have landed) ; s_barrier  X(t) ---
//     k-step 3 : MFMA(t,3)  | read (t+1,0) | DMA pieces 0..7 of tile t+2 -> buffer t&1
// RAW: pieces of tile t+1 were issued after X(t-1), every wave waits for its own before X(t), the first read of tile t+1 comes after
// X(t).  WAR: buffer t&1 is re-staged only after X(t), when every wave's last read of it (k-step 3 fragments) has returned.  The last
// piece of a tile is issued one k-step (>= 512 cycles) after X and waited for three k-steps later: >= 1024 cycles to land, the first
// ones 1536.  Past the last tile the prefetch index is clamped (redundant loads into dead buffers): the loop body is branch-free.
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = 128;              // bytes per LDS row
constexpr int OP_BYTES = 256 * ROWB;   // 32 KiB per operand image
constexpr int BUF_BYTES = 2 * OP_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;  // 128 KiB

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

__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_w4(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA sources: unit u of an operand image = rows [8u, 8u+8) = 1 KiB.  64 units per K-tile (A 0..31, B 32..63), 16 per wave:
    // piece j of wave w stages unit w + 4j  (pieces 0..7 -> A, 8..15 -> B).
    const bf16* src[16];
    int dst[16];
    {
        const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int u = wave + 4 * j;
            const bool isB = u >= 32;
            const int unit = isB ? u - 32 : u;
            const int rl = unit * 8 + lrow;
            const int chunk = pos ^ ((rl >> 1) & 7);
            if (isB) {
                const int r = min(n0 + rl, p.N - 1);
                src[j] = p.B + (int64_t)r * p.ldb + chunk * 8;
            } else {
                const int r = min(m0 + rl, p.M - 1);
                src[j] = p.A + (int64_t)r * p.lda + chunk * 8;
            }
            dst[j] = (isB ? OP_BYTES : 0) + unit * 1024;
        }
    }

    // ---- fragment offsets (bytes within an operand image); the swizzle term is lane-constant
    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;            // + i*32*ROWB, i = 0..3
    const int b_row0 = OP_BYTES + (wn * 128 + l31) * ROWB;  // + j*32*ROWB, j = 0..3

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int T = p.K / BK;
    // ---- prologue: tile 0 (16 pieces) and the first half of tile 1
#pragma unroll
    for (int j = 0; j < 16; ++j) __builtin_amdgcn_global_load_lds((gbl_void*)src[j], (lds_void*)(smem + dst[j]), 16, 0, 0);
    {
        const int64_t k1 = (int64_t)min(1, T - 1) * BK;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src[j] + k1), (lds_void*)(smem + BUF_BYTES + dst[j]), 16, 0, 0);
    }
    W4_VMCNT(8);
    W4_BARRIER();

    bf16x8 fa[2][4], fb[2][4];  // [k-step parity][row tile]
    // One k-step, written out in issue order: 8 groups of { 2 MFMAs ; 1 ds_read_b128 of the NEXT k-step's fragments ; optionally 1 LDS-DMA
    // piece }, each pinned by scheduling fences (the compiler still places the counted lgkmcnt waits).  Fragments are fetched in the order the
    // next k-step consumes them (B0 A0 B1 B2 B3 A1 A2 A3: its first four MFMAs need B0..B3 and A0).
    //   CUR / NXT: fragment register set in use / being filled;  rbuf: LDS buffer the reads come from;  S: k-step (0..3) being READ;
    //   dma0: first DMA piece of this k-step (-1: none), pieces go to dbuf at k-offset dko.
    // Fragment reads are OPAQUE asm ds_read_b128 with hand-counted lgkmcnt waits: left to the compiler, every k-step opened with
    // s_waitcnt lgkmcnt(0), i.e. waited for the read issued one instruction earlier (ISA inspected).  LDS returns in order, so with the
    // fetch order B0 A0 B1 B2 B3 A1 A2 A3 of the previous k-step and one new read issued per group of this one, group g may start when at
    // most WAIT[g] reads are outstanding:  g0 (needs B0 A0 B1) 5 | g1 (B2 B3) 4 | g2 (A1) 4 | g4 (A2) 5 | g6 (A3) 6 ; every fragment has
    // >= 5 groups (320 MFMA-cycles) between issue and first use.
    auto kstep = [&](auto cur_, auto s_, uint32_t rbuf, auto dma0_, char* dbuf, int64_t dko) {
        constexpr int CUR = decltype(cur_)::value, NXT = CUR ^ 1, S = decltype(s_)::value, DMA0 = decltype(dma0_)::value;
        const uint32_t ra = rbuf + a_row0 + koffb[S];
        const uint32_t rb = rbuf + b_row0 + koffb[S];
        afk_static_for<8>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            constexpr int n0_ = 2 * g, n1_ = 2 * g + 1;
            if constexpr (g == 0) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fb[CUR][0], fa[CUR][0], fb[CUR][1]); }
            if constexpr (g == 1) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fb[CUR][2], fb[CUR][3]); }
            if constexpr (g == 2) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fa[CUR][1]); }
            if constexpr (g == 4) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][2]); }
            if constexpr (g == 6) { asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); afk_lds_tie(fa[CUR][3]); }
            acc[n0_ >> 2][n0_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][n0_ & 3], fa[CUR][n0_ >> 2], acc[n0_ >> 2][n0_ & 3], 0, 0, 0);
            acc[n1_ >> 2][n1_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][n1_ & 3], fa[CUR][n1_ >> 2], acc[n1_ >> 2][n1_ & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // fetch order B0 A0 B1 B2 B3 A1 A2 A3
            if constexpr (g == 0) fb[NXT][0] = w4_lds_read<0>(rb);
            if constexpr (g == 1) fa[NXT][0] = w4_lds_read<0>(ra);
            if constexpr (g == 2) fb[NXT][1] = w4_lds_read<1 * 32 * ROWB>(rb);
            if constexpr (g == 3) fb[NXT][2] = w4_lds_read<2 * 32 * ROWB>(rb);
            if constexpr (g == 4) fb[NXT][3] = w4_lds_read<3 * 32 * ROWB>(rb);
            if constexpr (g == 5) fa[NXT][1] = w4_lds_read<1 * 32 * ROWB>(ra);
            if constexpr (g == 6) fa[NXT][2] = w4_lds_read<2 * 32 * ROWB>(ra);
            if constexpr (g == 7) fa[NXT][3] = w4_lds_read<3 * 32 * ROWB>(ra);
            if constexpr (DMA0 >= 0)
                __builtin_amdgcn_global_load_lds((gbl_void*)(src[DMA0 + g] + dko), (lds_void*)(dbuf + dst[DMA0 + g]), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I8 = std::integral_constant<int, 8>;
    using IN = std::integral_constant<int, -1>;

    // fragments of (tile 0, k-step 0), in the fetch order the wait ladder assumes
    const uint32_t lds0 = afk_lds_addr(smem);
    {
        const uint32_t ra = lds0 + a_row0 + koffb[0], rb = lds0 + b_row0 + koffb[0];
        fb[0][0] = w4_lds_read<0>(rb);
        fa[0][0] = w4_lds_read<0>(ra);
        fb[0][1] = w4_lds_read<1 * 32 * ROWB>(rb);
        fb[0][2] = w4_lds_read<2 * 32 * ROWB>(rb);
        fb[0][3] = w4_lds_read<3 * 32 * ROWB>(rb);
        fa[0][1] = w4_lds_read<1 * 32 * ROWB>(ra);
        fa[0][2] = w4_lds_read<2 * 32 * ROWB>(ra);
        fa[0][3] = w4_lds_read<3 * 32 * ROWB>(ra);
    }
    __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < T; ++t) {
        const uint32_t rown = lds0 + (t & 1) * BUF_BYTES, roth = lds0 + ((t + 1) & 1) * BUF_BYTES;
        char* own = smem + (t & 1) * BUF_BYTES;
        char* oth = smem + ((t + 1) & 1) * BUF_BYTES;
        const int64_t o1 = (int64_t)min(t + 1, T - 1) * BK, o2 = (int64_t)min(t + 2, T - 1) * BK;
        kstep(I0{}, I1{}, rown, I8{}, oth, o1);   // MFMA(t,0) | read (t,1) | DMA pieces 8..15 of tile t+1
        kstep(I1{}, I2{}, rown, IN{}, oth, o1);   // MFMA(t,1) | read (t,2)
        kstep(I0{}, I3{}, rown, IN{}, oth, o1);   // MFMA(t,2) | read (t,3)
        W4_LGKMCNT0();
        W4_VMCNT(0);
        W4_BARRIER();                             // X(t)
        kstep(I1{}, I0{}, roth, I0{}, own, o2);   // MFMA(t,3) | read (t+1,0) | DMA pieces 0..7 of tile t+2 into the buffer just retired
    }
    W4_LGKMCNT0();  // the last k-step fetched fragments of a tile that does not exist: retire them before the registers are reused
    __builtin_amdgcn_s_setprio(0);
    W4_VMCNT(0);  // no LDS-DMA may be in flight when the workgroup releases its LDS

    // ---- epilogue: lane holds row m = ..+l31 and n = ..+8q+4hi+{0..3} of each 32x32 block
    afk_static_for<16>([&](auto ij_) {
        constexpr int i = decltype(ij_)::value >> 2, j = decltype(ij_)::value & 3;
        gemm_store_block32(p, m0 + wm * 128 + i * 32 + l31, n0 + wn * 128 + j * 32, hi, acc[i][j]);
    });
}

}  // namespace

int afk_launch_gemm256w4(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_bf16_w4, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "gemm256w4: cannot reserve %d bytes of LDS", LDS_BYTES);
        attr_set = true;
    }
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    hipLaunchKernelGGL(gemm_nt_bf16_w4, dim3((unsigned)nwg), dim3(256), LDS_BYTES, st, p);
    return AFK_OK;
}
