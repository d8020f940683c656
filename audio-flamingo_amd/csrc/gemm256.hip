// bf16 NT GEMM, 256x256x64 "ping-pong" kernel for gfx950 (the high-throughput variant of gemm.hip).
//
// Why: the 128x128 kernel needs 64 B/clk/CU of L2->LDS traffic at full MFMA rate - more than the ~56 B/clk/CU the
// L2s deliver - and tops out at ~35 % of peak.  A 256x256 tile halves the traffic per flop; the price is one
// workgroup per CU (128 KiB LDS), so latency must be hidden inside the workgroup.
//
// Structure.  8 waves: wave w -> M half wm = w>>2 (128 rows), N quarter wn = w&3 (64 cols); waves w and w+4 share a
// SIMD, so the two M-halves form two GROUPS with one wave of each on every SIMD.  The groups run the same program
// one barrier apart: while group 0 is in an MFMA segment (16 x v_mfma_f32_32x32x16_bf16 = 512 cycles of that SIMD's
// matrix pipe, s_setprio 1) group 1 is in a MEMORY segment (ds_read_b128 fragments + LDS-DMA issue) and vice versa,
// so every SIMD's matrix pipe always has an owner.  Per K-tile (64) a wave runs
//     MEM_a : read B (64 rows x 64 k = 8 frags, kept for the whole tile) + A rows 0..63 of its half (8 frags)
//     MFMA_a: 2x2 tiles x 4 k-steps                       -> output rows 0..63
//     MEM_b : read A rows 64..127 (8 frags)
//     MFMA_b: 2x2 tiles x 4 k-steps                       -> output rows 64..127
// each followed by one s_barrier (4 per K-tile).  Accumulators: 4x2 tiles x 16 = 128 VGPRs; operands 64 VGPRs.
//
// LDS: two 64 KiB buffers (tile parity), each = A image (256 rows x 128 B) + B image.  Row r stores 16-B chunk c at
// position c ^ ((r>>1)&7) (conflict-free ds_read_b128 for the 32x32x16 fragment; the permutation is applied on the
// SOURCE address of the lane-linear LDS-DMA and again on the read, guide rule 21).
// Prefetch schedule (LDS-DMA, 1 KiB per wave-instruction, 64 per K-tile = 8 per wave): rows die in two classes,
//     X = B rows + A rows {0..63, 128..191}   last read in MEM_a   -> restaged for tile t+2 inside MFMA_b(t) (6 per wave)
//     Y =          A rows {64..127, 192..255} last read in MEM_b   -> restaged for tile t+1 inside MFMA_a(t) (2 per wave)
// (the LDS-DMA pieces are slotted between MFMAs: their ~100-cycle issue cost disappears in the MFMA shadow)
// so every load has >= 4 segments (~2000 cycles) to land.  Per wave the issue order is X0 Y0 X1 | Y1 X2 | Y2 X3 ...
// and counted waits before the barriers (vmcnt 6 / 8 / 2 / 8 after MEM_a / MFMA_a / MEM_b / MFMA_b) retire exactly the
// class the NEXT segment of the other group reads (derivation in DESIGN.md §3); never vmcnt(0) in the steady state.  ds_reads are retired
// (lgkmcnt(0)) before the barrier that ends a MEM segment, so a slot is never restaged under an outstanding read.
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

// EPI: compile-time epilogue flags (gemm_common.h), -1 = runtime flags / narrow stores / split-K partials
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_k256(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // wm doubles as the ping-pong group
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
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
                // SWIGLU_FWD: N-tile tn = 128 gate rows [128 tn, +128) followed by the 128 up rows [I + 128 tn, +128) of the fused weight,
                // so that one workgroup holds gate AND up of the same 128 intermediate features
                const int r = (EPI == AFK_GEMM_SWIGLU_FWD) ? (rl < 128 ? tn * 128 + rl : (p.N >> 1) + tn * 128 + rl - 128) : min(n0 + rl, p.N - 1);
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

    // ---- fragment offsets (bytes within an operand image); the swizzle term is lane-constant
    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * ROWB;            // + i*32*ROWB, i = 0..3 (i<2: MEM_a, i>=2: MEM_b)
    const int b_row0 = OP_BYTES + (wn * 64 + l31) * ROWB;  // + j*32*ROWB, j = 0..1

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int T = p.K / BK;
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

    if constexpr (EPI == AFK_GEMM_SWIGLU_FWD) {
        // waves wn = 0, 1 hold gate columns, wn = 2, 3 the up columns of the SAME 64 features and rows: the up waves park their
        // bf16 pieces in the (now idle) operand buffers, the gate waves pick them up and write h = bf16(bf16(silu(g)) * u) beside g
        // (same arithmetic as silu_mul_fwd_kernel on the bf16-rounded GEMM results: bit-identical to the two-kernel form).
        typedef __attribute__((ext_vector_type(8))) __bf16 b8;
        const int I = p.N >> 1;
        const bool up = wn >= 2;
        const int feat0 = tn * 128 + (wn & 1) * 64;                 // first of this wave's 64 intermediate features
        char* xch = smem + (wm * 2 + (wn & 1)) * 16384 + lane * 16;  // piece q of the pair at + q * 1024
        b8 pc[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    b8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * t + e]), __float_as_uint(acc[i][j][8 * t + 4 + e]), false, false);
                        o[e] = (bf16)(__uint_as_float(r[0]) * p.alpha);
                        o[4 + e] = (bf16)(__uint_as_float(r[1]) * p.alpha);
                    }
                    pc[(i * 2 + j) * 2 + t] = o;
                }
        if (up) {
#pragma unroll
            for (int q = 0; q < 16; ++q) *(b8*)(xch + q * 1024) = pc[q];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = (i * 2 + j) * 2 + t;
                    const int f = feat0 + j * 32 + 16 * t + 8 * hi;   // feature (= column of h) of this piece
                    if (!up) {
                        const b8 u = *(const b8*)(xch + q * 1024);
                        b8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gf = (float)pc[q][e];
                            h[e] = (bf16)(rbf(gf * sigmoid_f(gf)) * (float)u[e]);
                        }
                        if (m < p.M) *(b8*)((bf16*)p.C2 + (int64_t)m * I + f) = h;
                    }
                    if (m < p.M) *(b8*)((bf16*)p.C + (int64_t)m * p.ldc + (up ? I : 0) + f) = pc[q];
                }
        }
        return;
    }
    if constexpr (EPI >= 0 && (EPI & AFK_GEMM_ROPE) != 0) {
        // qkv projection + rotary embedding (afk_gemm_nt_bf16_rope).  A 256-column tile holds two heads of 128 columns: waves wn = 0, 1 (2, 3) own the first /
        // second 64 columns of head 0 (1) for the same rows - each wave rounds its Linear output (+ bias) to bf16, parks the 16 pieces in the idle operand
        // buffers, and after the barrier takes its partner's (wn ^ 1: the rotate-half column d +- 64 of the same row) and applies rope_kernel's arithmetic
        // (elementwise.hip, sign = +1): x1' = bf16(bf16(x1 c) + bf16(-x2 s)), x2' = bf16(bf16(x2 c) + bf16(x1 s)).  Tiles right of rope_cols (the v heads) skip
        // the exchange.  Bit-identical to the two-launch form.
        typedef __attribute__((ext_vector_type(8))) __bf16 b8;
        const bool rot = n0 < p.rope_cols;                               // block-uniform
        char* mine = smem + (wm * 4 + wn) * 16384 + lane * 16;           // piece q at + q * 1024
        const char* theirs = smem + (wm * 4 + (wn ^ 1)) * 16384 + lane * 16;
        b8 pc[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = n0 + wn * 64 + j * 32 + 16 * t + 8 * hi;
                    b8 bv;
                    if constexpr ((EPI & AFK_GEMM_BIAS) != 0) bv = *(const b8*)(p.bias + n);
                    b8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * t + e]), __float_as_uint(acc[i][j][8 * t + 4 + e]), false, false);
                        float v0 = __uint_as_float(r[0]) * p.alpha, v1 = __uint_as_float(r[1]) * p.alpha;
                        if constexpr ((EPI & AFK_GEMM_BIAS) != 0) {
                            v0 += (float)bv[e];
                            v1 += (float)bv[4 + e];
                        }
                        o[e] = (bf16)v0;
                        o[4 + e] = (bf16)v1;
                    }
                    pc[(i * 2 + j) * 2 + t] = o;
                }
        if (rot) {
#pragma unroll
            for (int q = 0; q < 16; ++q) *(b8*)(mine + q * 1024) = pc[q];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + l31;
            const int mc = min(m, p.M - 1);
            const int64_t pr = p.rope_pos ? p.rope_pos[mc] : mc % p.rope_S;
            // lane-major tables (S % 32 == 0, no explicit positions): the 32 rows of this accumulator block are positions pr0 .. pr0 + 31 of one table block
            const bool lanes = p.rope_cos_lanes != nullptr;
            const int64_t lbase = lanes ? (((int64_t)((m0 + wm * 128 + i * 32) % p.rope_S) >> 5) * 16 * 32 + l31) * 8 : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = (i * 2 + j) * 2 + t;
                    const int cl = wn * 64 + j * 32 + 16 * t + 8 * hi;   // column inside the tile
                    b8 o = pc[q];
                    if (rot) {
                        const int d = cl & 127;                            // column inside the head (this wave's half: d < 64 for wn even)
                        const b8 other = *(const b8*)(theirs + q * 1024);
                        const b8 c = lanes ? *(const b8*)(p.rope_cos_lanes + lbase + (d >> 3) * 256) : *(const b8*)(p.rope_cos + pr * 128 + d);
                        const b8 sn = lanes ? *(const b8*)(p.rope_sin_lanes + lbase + (d >> 3) * 256) : *(const b8*)(p.rope_sin + pr * 128 + d);
                        const float sg = (wn & 1) ? 1.f : -1.f;            // first half: x1 c - x2 s; second half: x2 c + x1 s
#pragma unroll
                        for (int e = 0; e < 8; ++e)   // rbf_strict: see common.h (fp-contract would fuse one product into the add)
                            o[e] = (bf16)(rbf_strict((float)pc[q][e] * (float)c[e]) + rbf_strict(sg * (float)other[e] * (float)sn[e]));
                    }
                    if (m < p.M) *(b8*)((bf16*)p.C + (int64_t)m * p.ldc + n0 + cl) = o;
                }
        }
        return;
    }
    // ---- epilogue: lane holds row m = ..+l31 and n = ..+8q+4hi+{0..3}
    if (AFK_GM_NOEPI(p)) return;  // -DAFK_PROBES builds only: timing probe (afk_gemm_set_variant(2 + 256 * 0x40), wrong results): the kernel without its epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gemm_store_block32<EPI>(p, m0 + wm * 128 + i * 32 + l31, n0 + wn * 64 + j * 32, hi, acc[i][j]);
}

}  // namespace

// the epilogues of the AF3 training step get their own instantiation; anything else (SwiGLU backward, fp32 output, narrow stores) the generic one
#define AFK_EPI_LIST(X) X(0) X(AFK_GEMM_BIAS) X(AFK_GEMM_RESIDUAL) X(AFK_GEMM_BIAS | AFK_GEMM_RESIDUAL) X(AFK_GEMM_BIAS | AFK_GEMM_GELU) \
    X(AFK_GEMM_BIAS | AFK_GEMM_GELU | AFK_GEMM_RESIDUAL) X(AFK_GEMM_ACCUM) X(AFK_GEMM_SWIGLU_FWD) X(AFK_GEMM_SWIGLU_BWD) X(AFK_GEMM_ROPE) X(AFK_GEMM_ROPE | AFK_GEMM_BIAS) X(-1)

int afk_launch_gemm256(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
#define AFK_SET(F)                                                                                                                       \
    if (hipFuncSetAttribute((const void*)gemm_nt_bf16_k256<(F)>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) \
        return afk_set_error(AFK_ERR_LAUNCH, "gemm256: cannot reserve %d bytes of LDS", LDS_BYTES);
        AFK_EPI_LIST(AFK_SET)
#undef AFK_SET
        attr_set = true;
    }
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    const int f = (p.wide && p.splits <= 1) ? p.flags : -1;
    switch (f) {
#define AFK_CASE(F)                                                                                          \
    case (F):                                                                                                \
        if ((F) == -1) afk_count(AFK_CNT_GEMM_GENERIC);                                                      \
        hipLaunchKernelGGL(gemm_nt_bf16_k256<(F)>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);        \
        break;
        AFK_EPI_LIST(AFK_CASE)
#undef AFK_CASE
        default:   // an epilogue outside the list: runtime-flag instantiation
            afk_count(AFK_CNT_GEMM_GENERIC);
            hipLaunchKernelGGL(gemm_nt_bf16_k256<-1>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
    }
    return AFK_OK;
}
