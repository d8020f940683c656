#ifdef AFK_PROBES  // rejected schedule, kept for the probe tables of profiles/r02_gemm_probes.md: not part of the default libafk.so (make PROBES=1)
// bf16 NT GEMM, 256x256x64 ping-pong kernel of gemm256.hip as a PERSISTENT tile loop: one workgroup per CU walks the rasterised tile list
// (tile = blockIdx.x + i * gridDim.x) instead of one workgroup per tile.
//
// What it buys: the workgroup turnover between tiles (launch of 8 waves + 128 KiB of LDS: ~2.4 us per tile measured from the block
// timeline, profiles/r02_gemm_probes.md table 8) and the store tail of the epilogue (the wave retires only when its stores are acknowledged)
// no longer sit between two K loops: the next tile's prologue loads are issued right behind the epilogue stores.
// Why that is safe without knowing how vmcnt retires stores against loads on gfx950: every counted wait of the K loop only relies on the
// LOADS of one wave returning in order among themselves.  "at most n operations outstanding" then implies "at most n loads outstanding"
// whatever the stores do, so with stores of the previous tile still in flight the ladder can only over-wait (during the first K-tile).
// LDS hand-over between tiles: each wave stages fixed 1-KiB units; it passes s_waitcnt vmcnt(0) after its K loop, so its own (dead) clamped
// prefetches of the finished tile have landed before it issues the next tile's pieces into the same units; the group-equalising barrier
// behind the K loop guarantees the other group has finished its last fragment reads before anybody writes the buffers again.
#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = 128;              // bytes per LDS row
constexpr int OP_BYTES = 256 * ROWB;   // 32 KiB per operand image
constexpr int BUF_BYTES = 2 * OP_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;  // 128 KiB

#define AFK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AFK_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AFK_BARRIER()                         \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_k256p(GemmArgs p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // wm doubles as the ping-pong group
    const int hi = lane >> 5, l31 = lane & 31;

    // ---- fragment offsets (bytes within an operand image); the swizzle term is lane-constant
    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;            // + i*32*ROWB, i = 0..3 (i<2: MEM_a, i>=2: MEM_b)
    const int b_row0 = OP_BYTES + (wn * 64 + l31) * ROWB;  // + j*32*ROWB, j = 0..1
    const int T = p_in.K / BK;
    const int ntiles = p_in.ntm * p_in.ntn;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // the ~40 argument words are only needed at the two ends of a tile (source addresses, epilogue): re-read them from the kernarg
    // segment there instead of carrying them through the K loop (kept live across the tile loop they spilled 37 SGPRs into VGPR lanes)
    const GemmArgs* kp = (const GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const GemmArgs& p = *kp;
    int tm, tn;
    gemm_tile_of(p, tile, ntiles, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA sources.  unit u of an operand image = rows [8u, 8u+8).  X: 6 units per wave, Y: 2 units per wave.
    // X list (48): i<32 -> B unit i ; i>=32 -> A unit (i-32 < 8 ? i-32 : i-32+8)   (A rows 0..63, 128..191)
    // Y list (16): j<8  -> A unit 8+j ; else A unit 16+j                            (A rows 64..127, 192..255)
    const bf16* xsrc[6];
    int xdst[6];
    const bf16* ysrc[2];
    int ydst[2];
    {
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
    }
    auto issue_x = [&](int t) {
        char* base = smem + (t & 1) * BUF_BYTES;
        const int koff = t * BK;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[j] + koff), (lds_void*)(base + xdst[j]), 16, 0, 0);
    };
    auto issue_y = [&](int t) {
        char* base = smem + (t & 1) * BUF_BYTES;
        const int koff = t * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(ysrc[j] + koff), (lds_void*)(base + ydst[j]), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: X0 Y0 X1, then make X0 visible to everyone
    issue_x(0);
    issue_y(0);
    if (T > 1) {
        issue_x(1);
        AFK_VMCNT(8);
    } else {
        AFK_VMCNT(2);
    }
    AFK_BARRIER();
    if (wm == 1) AFK_BARRIER();  // group 1 runs one segment behind group 0

    bf16x8 bf[2][4], af[2][4];
    // one LDS-DMA piece slotted behind every group of MFMAs (issue cost hides in the MFMA shadow)
#define AFK_MFMA4(ACC0, s)                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)          \
            acc[ACC0 + i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j_][s], af[i_][s], acc[ACC0 + i_][j_], 0, 0, 0); \
    } while (0)
    // Branch-free steady state: past the last K-tile the prefetch index is clamped to T-1, i.e. the tail re-loads the
    // last tile into slots nobody reads again (dead by the same lifetime argument), so the vmcnt ladder never changes.
    for (int t = 0; t < T; ++t) {
        const char* buf = smem + (t & 1) * BUF_BYTES;
        const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
        const int e1 = (t + 1) & 1, e2 = t & 1;  // destination buffer parity follows the UNclamped tile index
        const int64_t oy = (int64_t)t1 * BK, ox = (int64_t)t2 * BK;
        char* by = smem + e1 * BUF_BYTES;
        char* bx = smem + e2 * BUF_BYTES;
        // ================= MEM_a(t): B fragments (whole tile) + A rows 0..63
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = *(const bf16x8*)(buf + b_row0 + j * 32 * ROWB + koffb[s]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + i * 32 * ROWB + koffb[s]);
        AFK_LGKMCNT0();
        AFK_VMCNT(6);
        AFK_BARRIER();
        // ================= MFMA_a(t) (+ Y(t+1): 2 pieces)
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
        AFK_VMCNT(8);
        AFK_BARRIER();
        // ================= MEM_b(t): A rows 64..127
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + (i + 2) * 32 * ROWB + koffb[s]);
        AFK_LGKMCNT0();
        AFK_VMCNT(2);
        AFK_BARRIER();
        // ================= MFMA_b(t) (+ X(t+2): 6 pieces)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0][s], af[i][s], acc[2 + i][0], 0, 0, 0);
                acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1][s], af[i][s], acc[2 + i][1], 0, 0, 0);
                const int piece = 2 * s + i;  // 0..7, pieces 0..5 carry a DMA
                if (piece < 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[piece] + ox), (lds_void*)(bx + xdst[piece]), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        AFK_VMCNT(8);
        AFK_BARRIER();
    }
    AFK_VMCNT(0);  // no LDS-DMA may be in flight when the workgroup releases its LDS
#undef AFK_MFMA4
    if (wm == 0) AFK_BARRIER();  // equalise barrier counts
    asm volatile("" : "+s"(kp));
    const GemmArgs& pe = *kp;     // epilogue arguments: loaded here, not held across the K loop

    // ---- epilogue: lane holds row m = ..+l31 and n = ..+8q+4hi+{0..3}
    if (!AFK_GM_NOEPI(pe))  // bit 6 of gm (afk_gemm_set_variant(13 + 256 * 0x40)): timing probe WITHOUT the epilogue (wrong results) - profiles/r02_gemm_probes.md §9
    afk_static_for<8>([&](auto ij_) {  // compile-time indices: inside the tile loop a #pragma unroll was not honoured and acc went through scratch
        constexpr int i = decltype(ij_)::value >> 1, j = decltype(ij_)::value & 1;
        gemm_store_block32(pe, m0 + wm * 128 + i * 32 + l31, n0 + wn * 64 + j * 32, hi, acc[i][j]);
    });
  }  // tile loop
}

}  // namespace

int afk_launch_gemm256p(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    static int ncu = 256;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_bf16_k256p, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "gemm256p: cannot reserve %d bytes of LDS", LDS_BYTES);
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        attr_set = true;
    }
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    hipLaunchKernelGGL(gemm_nt_bf16_k256p, dim3((unsigned)(nwg < ncu ? nwg : ncu)), dim3(512), LDS_BYTES, st, p);
    return AFK_OK;
}

#endif  // AFK_PROBES
