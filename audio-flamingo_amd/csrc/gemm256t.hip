// 256x256x64 ping-pong GEMM, transposed-operand variants (companion of gemm256.hip - read its header first).
//
//   NN  (dgrad):  C[M,N] = A[M,K] . Bt[K,N]        A k-contiguous,  B stored reduction-major  (dX = dY . W, W as nn.Linear stores it)
//   TN  (wgrad):  C[M,N] = At[K,M]^T . Bt[K,N]     both operands reduction-major             (dW = dY^T . X, activations as they lie)
//
// so the backward of a Linear needs NO transposed copy of anything: the W^T shadows and the activation transposes of
// round-1's first version (31 ms/step of pure HBM traffic, 15 GB of HBM) are gone.
//
// A reduction-major operand is staged as the image [64 k-rows][256 cols] (512-byte rows, LDS-DMA reads full 512-byte
// lines) and its MFMA fragment "col x 8 consecutive k" is fetched with two ds_read_b64_tr_b16 (4 k-rows x 16 cols per
// 16-lane group, lane i receives column i; rows 8hi..8hi+3 and 8hi+4..8hi+7 so that the k order matches the
// ds_read_b128 fragment of a k-contiguous operand).  16-byte chunk c of row r is stored at c ^ (((r&3)<<2)|((r>>2)&3)):
// the four rows of a tr-read block fall into four different 64-byte windows (conflict-free).
//
// Ping-pong schedule, LDS-DMA lifetime classes and the counted-vmcnt ladder are those of gemm256.hip with
//   NN: X = Bt image (32 pieces) + A rows {0..63,128..191} (16)  -> 6 per wave;  Y = A rows {64..127,192..255} -> 2 per wave
//       phases split by output rows (m halves); ladder vmcnt 6 / 8 / 2 / 8
//   TN: phases split by K (k-steps 0,1 | 2,3): X = k-rows 0..31 of both images (4 per wave), Y = k-rows 32..63 (4 per
//       wave); ladder vmcnt 4 / 8 / 4 / 8.  K need not be a multiple of 64: out-of-range k-rows are read clamped (finite
//       data) and the A fragments of the last tile are zeroed for k >= K.
#include "gemm_common.h"

namespace {

constexpr int BK = 64;
constexpr int OP_BYTES = 32768;
constexpr int BUF_BYTES = 2 * OP_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;
constexpr int TROW = 512;  // bytes per row of a transposed image

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define AFK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AFK_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AFK_BARRIER()                         \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
#define AFK_DMA_PTR(SRCPTR, DSTPTR)                                                                       \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        __builtin_amdgcn_global_load_lds((gbl_void*)(SRCPTR), (lds_void*)(DSTPTR), 16, 0, 0);             \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ int tswz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// fragment of a transposed image: lane -> column 32*tile32 + (lane&31), k = 16*s + 8*hi + 0..7.
// The swizzle term of a tr-read address does not depend on the k-step s (16*s leaves r&3 and (r>>2)&3 unchanged), so the
// two byte offsets of a (tile, piece) pair are lane constants: they are computed once per kernel and every read is
// "base + immediate" (s*8192 fits the 16-bit ds offset field) - no VALU in the MEM segments.
__device__ __forceinline__ int tr_off(int tile32, int pc, int lane) {
    const int g = lane >> 4, i = lane & 15, hi = g >> 1;
    const int chunk = 4 * tile32 + 2 * (g & 1) + ((i & 3) >> 1);
    const int r = 8 * hi + (i >> 2) + 4 * pc;  // + 16*s
    return r * TROW + ((chunk ^ tswz(r)) << 4) + 8 * (i & 1);
}
// opaque-asm reads (common.h: the builtin form drags an s_waitcnt vmcnt(0) into the loop); every MEM segment ends with
// AFK_LGKMCNT0 + a scheduling barrier before the first MFMA, which is the wait these reads need
template <int S>
__device__ __forceinline__ bf16x8 tr_frag(uint32_t buf, int off0, int off1) {
    return afk_lds_tr_frag<S * 16 * TROW>(buf + off0, buf + off1);
}

// per-lane source of one LDS-DMA piece (1 KiB = 2 k-rows x 512 B) of a transposed image
struct TSrc {
    const bf16* base;  // + column offset (already clamped into the row)
    int krow;          // k-row inside the tile (0..63)
};
__device__ __forceinline__ TSrc tsrc(const bf16* mat, int unit, int col0, int ncols, int lane) {
    const int krow = 2 * unit + (lane >> 5);
    const int pos = lane & 31;
    const int chunk = pos ^ tswz(krow);
    const int col = min(col0 + chunk * 8, ncols - 8);
    TSrc s;
    s.base = mat + col;
    s.krow = krow;
    return s;
}

// EPI: compile-time epilogue flags (gemm_common.h: one small epilogue instead of every variant inlined at each store site), -1 = runtime
template <bool AT, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_xt_bf16_k256(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    // split-K (blockIdx.y = split): this block reduces K-tiles [t0, t1) of its output tile; operands are re-based so that the body
    // below runs unchanged on the local reduction length KL
    int KL = p.K;
    const bf16* Abase = p.A;
    const bf16* Bbase = p.B;
    if (p.splits > 1) {
        const int tall = (p.K + BK - 1) / BK;
        const int t0 = (int)((int64_t)tall * blockIdx.y / p.splits), t1 = (int)((int64_t)tall * (blockIdx.y + 1) / p.splits);
        const int kbeg = t0 * BK;
        KL = min(p.K, t1 * BK) - kbeg;
        Abase += AT ? (int64_t)kbeg * p.lda : (int64_t)kbeg;
        Bbase += (int64_t)kbeg * p.ldb;
    }
    const int T = (KL + BK - 1) / BK;
    constexpr int NX = AT ? 4 : 6, NY = AT ? 4 : 2;

    // ------------------------------------------------------------------ LDS-DMA piece table
    // normal A image (NN only): unit = 8 rows x 128 B, chunk permutation c ^ ((r>>1)&7)
    const bf16* an_src[NX + NY];  // k-contiguous sources (advance by t*64 elements)
    TSrc tr_src[NX + NY];         // reduction-major sources (advance by t*64 rows)
    int64_t tr_ld[NX + NY];
    int dst[NX + NY];
#pragma unroll
    for (int j = 0; j < NX + NY; ++j) {
        const bool isx = j < NX;
        const int jj = isx ? j : j - NX;
        an_src[j] = nullptr;
        tr_src[j].base = nullptr;
        tr_src[j].krow = 0;
        tr_ld[j] = 0;
        if (AT) {
            // X: k-rows 0..31 (units 0..15) of A (jj<2) then B (jj>=2); Y: k-rows 32..63 (units 16..31) likewise
            const bool isA = jj < 2;
            const int unit = (isx ? 0 : 16) + wave + 8 * (jj & 1);
            tr_src[j] = isA ? tsrc(Abase, unit, m0, p.M, lane) : tsrc(Bbase, unit, n0, p.N, lane);
            tr_ld[j] = isA ? p.lda : p.ldb;
            dst[j] = (isA ? 0 : OP_BYTES) + unit * 1024;
        } else if (isx && jj < 4) {
            const int unit = wave + 8 * jj;  // Bt image, 32 pieces
            tr_src[j] = tsrc(Bbase, unit, n0, p.N, lane);
            tr_ld[j] = p.ldb;
            dst[j] = OP_BYTES + unit * 1024;
        } else {
            // A (k-contiguous): X (jj = 4,5) -> units {0..7, 16..23}, Y (jj = 0,1) -> units {8..15, 24..31}
            const int k = wave + 8 * (isx ? jj - 4 : jj);
            const int unit = isx ? (k < 8 ? k : k + 8) : (k < 8 ? 8 + k : 16 + k);
            const int rl = unit * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
            const int r = min(m0 + rl, p.M - 1);
            an_src[j] = Abase + (int64_t)r * p.lda + chunk * 8;
            dst[j] = unit * 1024;
        }
    }
    // reduction-major pieces: per-lane base already at the piece's k-row, advanced per K-tile by a wave-uniform (scalar) offset - one 64-bit
    // add per DMA.  The row clamp (k-rows beyond the reduction length) only exists in a ragged LAST tile: that one takes the general form.
    // (The general form on every piece - per-lane min + 64-bit multiply, ~10 VALU between two MFMAs, eight times per K-tile - cost the TN
    // kernel ~8 % against the NT kernel: profiles/r02_gemm_probes.md §12.)
    const bf16* tbase[NX + NY];
#pragma unroll
    for (int j = 0; j < NX + NY; ++j) tbase[j] = (AT || j < 4) ? tr_src[j].base + (int64_t)tr_src[j].krow * tr_ld[j] : nullptr;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool ragged = (KL % BK) != 0;
  // the whole pipeline twice: RAG = false (K a multiple of 64: every training shape) never clamps, RAG = true is the general form
  auto pipeline = [&](auto rag_) {
    constexpr bool RAG = decltype(rag_)::value;
    auto src_of = [&](int j, int t) -> const bf16* {
        const bool tr = AT || j < 4;  // compile-time after unrolling
        if (tr) {
            if constexpr (RAG) {
                const int row = min(t * BK + tr_src[j].krow, KL - 1);
                return tr_src[j].base + (int64_t)row * tr_ld[j];
            } else {
                return tbase[j] + (int64_t)t * BK * tr_ld[j];
            }
        }
        return an_src[j] + (int64_t)t * BK;
    };
    auto issue = [&](int j, int t_src, int parity) { AFK_DMA_PTR(src_of(j, t_src), smem + parity * BUF_BYTES + dst[j]); };

    // lane-constant tr-read offsets: B tiles wn*2+j, A tiles wm*4+i (TN)
    int tob[2][2], toa[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) tob[j][pc] = OP_BYTES + tr_off(wn * 2 + j, pc, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) toa[i][pc] = AT ? tr_off(wm * 4 + i, pc, lane) : 0;
    // fragment offsets for the k-contiguous A image (NN)
    const int swz_l = (lane >> 1) & 7;
    int koffb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koffb[s] = ((2 * s + hi) ^ swz_l) << 4;
    const int a_row0 = (wm * 128 + l31) * 128;

    // ------------------------------------------------------------------ prologue: X0 Y0 X1
#pragma unroll
    for (int j = 0; j < NX; ++j) issue(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NY; ++j) issue(NX + j, 0, 0);
    {
        const int t1 = min(1, T - 1);
#pragma unroll
        for (int j = 0; j < NX; ++j) issue(j, t1, 1);
    }
    if (AT) AFK_VMCNT(8); else AFK_VMCNT(8);  // NY + NX outstanding allowed: X0 has landed
    AFK_BARRIER();
    if (wm == 1) AFK_BARRIER();

    bf16x8 bf[2][4], af[AT ? 4 : 2][AT ? 2 : 4];
    const uint32_t lds0 = afk_lds_addr(smem);
    for (int t = 0; t < T; ++t) {
        const char* buf = smem + (t & 1) * BUF_BYTES;
        const uint32_t lbuf = lds0 + (t & 1) * BUF_BYTES;
        const int t1 = min(t + 1, T - 1), t2 = min(t + 2, T - 1);
        const int e1 = (t + 1) & 1, e2 = t & 1;
        if (!AT) {
            // ================= NN  MEM_a: Bt fragments (whole tile) + A rows 0..63
#pragma unroll
            for (int j = 0; j < 2; ++j)
                afk_static_for<4>([&](auto s_) { constexpr int s = decltype(s_)::value; bf[j][s] = tr_frag<s>(lbuf, tob[j][0], tob[j][1]); });
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + i * 32 * 128 + koffb[s]);
            AFK_LGKMCNT0();
            AFK_VMCNT(6);
            AFK_BARRIER();
            // ================= MFMA_a (+ Y(t+1))
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MFMA(bf[j][s], af[i][s], acc[i][j]);
                if (s < 2) issue(NX + s, t1, e1);
            }
            __builtin_amdgcn_s_setprio(0);
            AFK_VMCNT(8);
            AFK_BARRIER();
            // ================= MEM_b: A rows 64..127
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i][s] = *(const bf16x8*)(buf + a_row0 + (i + 2) * 32 * 128 + koffb[s]);
            AFK_LGKMCNT0();
            AFK_VMCNT(2);
            AFK_BARRIER();
            // ================= MFMA_b (+ X(t+2))
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[2 + i][0] = MFMA(bf[0][s], af[i][s], acc[2 + i][0]);
                    acc[2 + i][1] = MFMA(bf[1][s], af[i][s], acc[2 + i][1]);
                    const int piece = 2 * s + i;
                    if (piece < 6) issue(piece, t2, e2);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            AFK_VMCNT(8);
            AFK_BARRIER();
        } else {
            const int kvalid = KL - t * BK;  // < 64 only on a ragged last tile
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                // ================= TN  MEM: fragments of k-steps {2ph, 2ph+1} of both images
                afk_static_for<2>([&](auto s_) {
                    constexpr int s = decltype(s_)::value;
                    if (ph == 0) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) bf[j][s] = tr_frag<s>(lbuf, tob[j][0], tob[j][1]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[i][s] = tr_frag<s>(lbuf, toa[i][0], toa[i][1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 2; ++j) bf[j][s] = tr_frag<2 + s>(lbuf, tob[j][0], tob[j][1]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[i][s] = tr_frag<2 + s>(lbuf, toa[i][0], toa[i][1]);
                    }
                });
                AFK_LGKMCNT0();
                if (RAG && kvalid < BK) {  // block-uniform, ragged last tile only (the RAG = false pipeline has no such tile): zero the A contribution of k >= K
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const bool dead = 16 * (2 * ph + s) + 8 * hi + e >= kvalid;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (dead) af[i][s][e] = (bf16)0.f;
                        }
                }
                AFK_VMCNT(4);
                AFK_BARRIER();
                // ================= MFMA (+ Y(t+1) in phase 0, X(t+2) in phase 1: 4 pieces each)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][0] = MFMA(bf[0][s], af[i][s], acc[i][0]);
                        acc[i][1] = MFMA(bf[1][s], af[i][s], acc[i][1]);
                        const int piece = 4 * s + i;  // 0..7, even pieces carry a DMA
                        if ((piece & 1) == 0) {
                            if (ph == 0) issue(NX + (piece >> 1), t1, e1);
                            else issue(piece >> 1, t2, e2);
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                AFK_VMCNT(8);
                AFK_BARRIER();
            }
        }
    }
    AFK_VMCNT(0);
    if (wm == 0) AFK_BARRIER();
  };
    if (ragged) pipeline(std::true_type{});
    else pipeline(std::false_type{});

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gemm_store_block32<EPI>(p, m0 + wm * 128 + i * 32 + l31, n0 + wn * 64 + j * 32, hi, acc[i][j]);
}

}  // namespace

// weight gradients write or accumulate plain bf16 (flags 0 / ACCUM); the generic instantiation serves split-K partials and the rest
#define AFK_EPI_LIST(X) X(0) X(AFK_GEMM_ACCUM) X(-2) X(-1)

int afk_launch_gemm256t(const GemmArgs& p, int trans_a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
#define AFK_SET(F)                                                                                                                                \
    if (hipFuncSetAttribute((const void*)gemm_xt_bf16_k256<false, (F)>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess || \
        hipFuncSetAttribute((const void*)gemm_xt_bf16_k256<true, (F)>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)    \
        return afk_set_error(AFK_ERR_LAUNCH, "gemm256t: cannot reserve %d bytes of LDS", LDS_BYTES);
        AFK_EPI_LIST(AFK_SET)
#undef AFK_SET
        attr_set = true;
    }
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    const unsigned ns = (unsigned)(p.splits > 1 ? p.splits : 1);
    const int f = p.splits > 1 ? -2 : (p.wide ? p.flags : -1);  // -2: split-K partial sums (the fold kernel applies the epilogue)
    const dim3 grid((unsigned)nwg, ns);
    switch (f) {
#define AFK_CASE(F)                                                                                                   \
    case (F):                                                                                                         \
        if ((F) == -1) afk_count(AFK_CNT_GEMM_GENERIC);                                                               \
        if (trans_a) hipLaunchKernelGGL((gemm_xt_bf16_k256<true, (F)>), grid, dim3(512), LDS_BYTES, st, p);          \
        else hipLaunchKernelGGL((gemm_xt_bf16_k256<false, (F)>), grid, dim3(512), LDS_BYTES, st, p);                 \
        break;
        AFK_EPI_LIST(AFK_CASE)
#undef AFK_CASE
        default:   // an epilogue outside the list: runtime-flag instantiation
            afk_count(AFK_CNT_GEMM_GENERIC);
            if (trans_a) hipLaunchKernelGGL((gemm_xt_bf16_k256<true, -1>), grid, dim3(512), LDS_BYTES, st, p);
            else hipLaunchKernelGGL((gemm_xt_bf16_k256<false, -1>), grid, dim3(512), LDS_BYTES, st, p);
    }
    return AFK_OK;
}
