#ifdef AFK_PROBES  // rejected schedule, kept for the probe tables of profiles/r02_gemm_probes.md: not part of the default libafk.so (make PROBES=1)
// bf16 NT GEMM, 256x256 output tile, K-step 32, EIGHT free-running waves of 128x64, ten-slot LDS ring (round 2).
//
// What the round-2 probes showed (tools/bench_gemm.py variants 2..9, profiles/r02_gemm_probes.md), on random operands where the chip is
// power-limited (the same kernel runs 1.35-1.4x faster on zero-filled operands at ~2.3 GHz instead of ~1.85):
//   * the 8-wave ping-pong kernel (gemm256.hip) loses ~30 % of the matrix pipe to its four barriers per 64-wide K-tile;
//   * one wave per SIMD (gemm256w4.hip, 128x128 per wave) cannot hide its own LDS-DMA issue (a lone wave has only one queued MFMA of cover);
//   * two FREE-RUNNING waves per SIMD (no ping-pong barriers; when one is blocked in a DMA issue or a counted wait its sibling's MFMAs own
//     the pipe) with ONE barrier per K-tile reach the MFMA-only ceiling as soon as the loads never have to be waited for: with a 64-wide
//     K-tile and five 32 KiB slots the tightest load had 1024 cycles to land and the kernel spent ~20 % of its time in vmcnt waits
//     (probe "DMA issued, never waited" = "no DMA at all" on random data).
// Hence this kernel: the same free-running 8 waves, but K-tiles of 32 (16 KiB operand images, 64-byte LDS rows) in a ring of TEN slots
// (all 160 KiB of the CU): tile t reads images 2t (A) and 2t+1 (B); the slots it frees at its barrier X(t) are refilled with tile t+5,
// which is first read after X(t+4): every load has >= 3.5 K-tiles (~3 600 MFMA-cycles, ~1.8 us) to land.
//
// Per wave and K-tile (2 k-steps of 16): 16 MFMAs 32x32x16, 12 ds_read_b128 (next k-step's fragments, register double buffer, opaque asm
// reads with a hand-counted lgkmcnt ladder), 4 LDS-DMA pieces (1 KiB = 16 rows x 64 B), one barrier:
//     k-step 0 : MFMA(t,0) | read (t,1)   | B(t+4): 2 pieces  (slot freed at X(t-1))
//     --- lgkmcnt(0) (my reads of tile t are back) ; vmcnt(12) (only tiles t+2, t+3, t+4 may still fly: tile t+1 is complete) ; X(t) ---
//     k-step 1 : MFMA(t,1) | read (t+1,0) | A(t+5): 2 pieces  (slot of A(t), free since X(t))
// LDS rows are 64 B = four 16-byte chunks; chunk c of row r is stored at position c ^ ((r >> 2) & 3): every ds_read_b128 lane group
// (16 lanes) then covers 16 distinct 16-byte slots of the 256-byte bank row (conflict-free; the permutation is applied to the LDS-DMA source
// address, the destination stays lane-linear).  Past the last tile the prefetch index is clamped: the loop body is branch-free.
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROWB = 64;               // bytes per LDS row
constexpr int IMG_BYTES = 256 * ROWB;  // 16 KiB per operand image
constexpr int NSLOT = 10;
constexpr int LDS_BYTES = NSLOT * IMG_BYTES;  // 160 KiB
constexpr int AHEAD = 5;               // tile t's slots are refilled with tile t + AHEAD

#define F8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define F8_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define F8_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

template <int OFF>
__device__ __forceinline__ bf16x8 f8_lds_read(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// MODE 0: the kernel.  MODE 1: timing probe (wrong results): no LDS-DMA in the loop.  MODE 2: the kernel + s_memtime stamps around the
// per-tile wait: every wave adds up the cycles it spends in [lgkmcnt(0) + vmcnt] and in the barrier and writes
// {loop cycles, vm-wait cycles, barrier cycles, tiles} as 4 floats to ((float*)p.C2)[(block * 8 + wave) * 4] (tools/gemm_waits.py).
template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_f8(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 128 rows x 64 columns per wave
    const int hi = lane >> 5, l31 = lane & 31;

    uint64_t rt_entry = 0, rt_pro = 0, rt_loop = 0;
    if constexpr (MODE == 2) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt_entry)::"memory");
    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA sources: unit u of an operand image = rows [16u, 16u+16) = 1 KiB; piece j (0, 1) of wave w stages unit w + 8j
    const bf16* srcA[2];
    const bf16* srcB[2];
    {
        const int lrow = lane >> 2, pos = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = (wave + 8 * j) * 16 + lrow;
            const int chunk = pos ^ ((rl >> 2) & 3);
            srcA[j] = p.A + (int64_t)min(m0 + rl, p.M - 1) * p.lda + chunk * 8;
            srcB[j] = p.B + (int64_t)min(n0 + rl, p.N - 1) * p.ldb + chunk * 8;
        }
    }
    const int dst0 = wave * 1024;  // + j*8192 within an image

    // ---- fragment offsets (bytes within an operand image); the swizzle term is lane-constant (row bases are multiples of 32)
    const int swz_l = (l31 >> 2) & 3;
    const int koff0 = ((0 + hi) ^ swz_l) << 4, koff1 = ((2 + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;  // + i*32*ROWB, i = 0..3
    const int b_row0 = (wn * 64 + l31) * ROWB;   // + j*32*ROWB, j = 0..1

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int T = p.K / BK;
    // ---- prologue: tiles 0..3 and A of tile 4 (images 0..8 -> slots 0..8), 18 pieces per wave
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) {
        const int64_t ko = (int64_t)min(t, T - 1) * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(srcA[j] + ko), (lds_void*)(smem + (2 * t) * IMG_BYTES + dst0 + j * 8192), 16, 0, 0);
        if (t < AHEAD - 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void*)(srcB[j] + ko), (lds_void*)(smem + (2 * t + 1) * IMG_BYTES + dst0 + j * 8192), 16, 0, 0);
        }
    }
    F8_VMCNT(14);  // tile 0 (the first 4 pieces) has landed
    F8_BARRIER();
    if constexpr (MODE == 2) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt_pro)::"memory");

    bf16x8 fa[2][4], fb[2][2];  // [k-step parity][row tile]
    // One k-step in issue order: 4 groups of 2 MFMAs; fragments of the NEXT k-step are fetched B0 B1 | A0 A1 | A2 | A3 over the four groups
    // (what the next k-step's first MFMAs need comes first), the two DMA pieces ride behind groups 2 and 3.  LDS returns in order, one
    // k-step's 6 reads are all issued before the next k-step starts, so group g may start when at most WAIT[g] reads are outstanding:
    //     g0 (needs B0 B1 A0) 3 | g1 (A1) 4 | g2 (A2) 5 | g3 (A3) 5
    auto kstep = [&](auto cur_, uint32_t ra, uint32_t rb, const bf16* const (&dsrc)[2], int64_t dko, char* dbase) {
        constexpr int CUR = decltype(cur_)::value, NXT = CUR ^ 1;
        afk_static_for<4>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 0) { asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); afk_lds_tie(fb[CUR][0], fb[CUR][1], fa[CUR][0]); }
            if constexpr (g == 1) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); afk_lds_tie(fa[CUR][1]); }
            if constexpr (g == 2) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][2]); }
            if constexpr (g == 3) { asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); afk_lds_tie(fa[CUR][3]); }
            acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][0], fa[CUR][g], acc[g][0], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][1], fa[CUR][g], acc[g][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == 0) { fb[NXT][0] = f8_lds_read<0>(rb); fb[NXT][1] = f8_lds_read<32 * ROWB>(rb); }
            if constexpr (g == 1) { fa[NXT][0] = f8_lds_read<0>(ra); fa[NXT][1] = f8_lds_read<32 * ROWB>(ra); }
            if constexpr (g == 2) fa[NXT][2] = f8_lds_read<2 * 32 * ROWB>(ra);
            if constexpr (g == 3) fa[NXT][3] = f8_lds_read<3 * 32 * ROWB>(ra);
            if constexpr ((MODE == 0 || MODE == 2) && g >= 2)
                __builtin_amdgcn_global_load_lds((gbl_void*)(dsrc[g - 2] + dko), (lds_void*)(dbase + dst0 + (g - 2) * 8192), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    const uint32_t lds0 = afk_lds_addr(smem);
    {   // fragments of (tile 0, k-step 0), in the fetch order the wait ladder assumes
        const uint32_t ra = lds0 + a_row0 + koff0, rb = lds0 + IMG_BYTES + b_row0 + koff0;
        fb[0][0] = f8_lds_read<0>(rb);
        fb[0][1] = f8_lds_read<32 * ROWB>(rb);
        fa[0][0] = f8_lds_read<0>(ra);
        fa[0][1] = f8_lds_read<32 * ROWB>(ra);
        fa[0][2] = f8_lds_read<2 * 32 * ROWB>(ra);
        fa[0][3] = f8_lds_read<3 * 32 * ROWB>(ra);
    }
    // ring positions (scalar): A(t) sits in slot sa, B(t) in sa + 1 (sa is even), A(t+1) in sa + 2, ... modulo 10
    int sa = 0;
    uint64_t c_start = 0, c0 = 0, c1 = 0, c2 = 0, sum_vm = 0, sum_bar = 0;
    if constexpr (MODE == 2) {
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c_start)::"memory");
    }
    for (int t = 0; t < T; ++t) {
        const int s_next = sa + 2 >= NSLOT ? 0 : sa + 2;                        // A(t+1)
        const int s_b4 = sa + 9 >= NSLOT ? sa + 9 - NSLOT : sa + 9;             // B(t+4): image 2t+9
        const int64_t o4 = (int64_t)min(t + AHEAD - 1, T - 1) * BK, o5 = (int64_t)min(t + AHEAD, T - 1) * BK;
        const uint32_t ra0 = lds0 + sa * IMG_BYTES + a_row0, rb0 = ra0 - a_row0 + IMG_BYTES + b_row0;
        const uint32_t ra1 = lds0 + s_next * IMG_BYTES + a_row0, rb1 = ra1 - a_row0 + IMG_BYTES + b_row0;
        kstep(I0{}, ra0 + koff1, rb0 + koff1, srcB, o4, smem + s_b4 * IMG_BYTES);   // MFMA(t,0) | read (t,1)   | B(t+4)
        F8_LGKMCNT0();
        if constexpr (MODE == 2) asm volatile("s_memtime %0" : "=s"(c0)::"memory");
        if constexpr (MODE == 0 || MODE == 2) F8_VMCNT(12); else F8_VMCNT(0);
        if constexpr (MODE == 2) asm volatile("s_memtime %0" : "=s"(c1)::"memory");
        F8_BARRIER();                                                                 // X(t)
        if constexpr (MODE == 2) {
            asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c2)::"memory");
            sum_vm += c1 - c0;
            sum_bar += c2 - c1;
        }
        kstep(I1{}, ra1 + koff0, rb1 + koff0, srcA, o5, smem + sa * IMG_BYTES);     // MFMA(t,1) | read (t+1,0) | A(t+5) into the slot of A(t)
        sa = s_next;
    }
    F8_LGKMCNT0();  // the last k-step fetched fragments of a tile that does not exist: retire them before the registers are reused
    F8_VMCNT(0);    // no LDS-DMA may be in flight when the workgroup releases its LDS
    if constexpr (MODE == 2) {
        uint64_t c_end;
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c_end), "=s"(rt_loop)::"memory");
        if (lane == 0 && p.C2) {
            float* o = (float*)p.C2 + ((int64_t)blockIdx.x * 8 + wave) * 4;
            o[0] = (float)(c_end - c_start);
            o[1] = (float)sum_vm;
            o[2] = (float)sum_bar;
            o[3] = (float)T;
        }
    }

    // ---- epilogue: lane holds row m = ..+l31 and n = ..+8q+4hi+{0..3} of each 32x32 block
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gemm_store_block32(p, m0 + wm * 128 + i * 32 + l31, n0 + wn * 64 + j * 32, hi, acc[i][j]);
    if constexpr (MODE == 2) {
        // block timeline in s_memrealtime ticks (100 MHz): entry, prologue done, loop done, stores retired; + hardware id (CU) of the block
        uint64_t rt_end;
        F8_VMCNT(0);
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
        if (tid == 0 && p.C2) {
            const uint32_t hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
            double* o = (double*)((float*)p.C2 + (int64_t)gridDim.x * 8 * 4) + (int64_t)blockIdx.x * 6;
            o[0] = (double)rt_entry; o[1] = (double)rt_pro; o[2] = (double)rt_loop; o[3] = (double)rt_end; o[4] = (double)hwid; o[5] = (double)xcc;
        }
    }
}

}  // namespace

// mode 0: the kernel; 1: its no-DMA timing probe
int afk_launch_gemm256f8(const GemmArgs& p, int mode, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_bf16_f8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)gemm_nt_bf16_f8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)gemm_nt_bf16_f8<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "gemm256f8: cannot reserve %d bytes of LDS", LDS_BYTES);
        attr_set = true;
    }
    const dim3 grid((unsigned)((int64_t)p.ntm * p.ntn));
    if (mode == 1) hipLaunchKernelGGL(gemm_nt_bf16_f8<1>, grid, dim3(512), LDS_BYTES, st, p);
    else if (mode == 2) hipLaunchKernelGGL(gemm_nt_bf16_f8<2>, grid, dim3(512), LDS_BYTES, st, p);
    else hipLaunchKernelGGL(gemm_nt_bf16_f8<0>, grid, dim3(512), LDS_BYTES, st, p);
    return AFK_OK;
}

#endif  // AFK_PROBES
