// Flash attention v2 for gfx950: LDS-staged K/V (forward, dQ) and Q/dO (dK/dV) tiles, transposed MFMA operands taken
// straight from the row-major LDS image with ds_read_b64_tr_b16 - no transposed copies in HBM at all.
//
// Same math / register dataflow as attention.hip (see its header): S^T = K.Q^T with the probabilities landing in the
// k-operand layout of the next MFMA.  What changes is where the streamed operand comes from:
//   * a 256-thread block (4 waves x 32 rows) shares each 64-row tile of the streamed tensors through LDS, filled by
//     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), double buffered, one barrier per tile;
//   * row fragments (K, V, Q, dO as "row x 8 consecutive d") are ds_read_b128;
//   * transposed fragments (V^T for P.V, K^T for dS.K, dO^T for P^T.dO, Q^T for dS^T.Q: "d x 8 keys") are two
//     ds_read_b64_tr_b16 each: within a 16-lane group lane i receives column i of the 4x16 block whose rows the lanes
//     address (semantics probed on hardware: profiles/r01_probe_ds_read_b64_tr_b16.txt).  The two reads fetch keys
//     {16s+4hi+0..3} and {16s+8+4hi+0..3} - exactly the keys a lane's probability registers 8s..8s+7 belong to.
// LDS image: [64 rows][D] bf16, 16-byte chunk c of row r stored at c ^ f(r) with
//     D=128 (16 chunks, row = one 256-B bank row): f = ((r&3)<<2) | ((r>>2)&3)
//     D=64  ( 8 chunks, two rows per bank row):    f = (((r>>1)&1)<<2) | ((r>>2)&3)
// which makes BOTH access patterns conflict-free: a ds_read_b128 lane group sees 16 distinct slots, and the four rows
// of a tr-read block land in four different 64-byte windows.  (The permutation is applied to the LDS-DMA source
// address, the destination stays lane-linear.)
#include "common.h"
#include "../../include/afk.h"

// Timing probes (tools/attn_waits.py, profiles/r02_attn_probes.md) produce WRONG results on purpose (skipped kernels, skipped MFMA bodies).
// They exist only in -DAFK_PROBES builds (make PROBES=1); the default build compiles every probe branch away.
#ifdef AFK_PROBES
#define AFK_DBG(p) ((p).dbg)
#else
#define AFK_DBG(p) 0
#endif

namespace {

struct AttnArgs2 {
    const bf16* Q; int64_t q_bs, q_hs, q_rs;
    const bf16* K; int64_t k_bs, k_hs, k_rs;
    const bf16* V; int64_t v_bs, v_hs, v_rs;
    bf16* O; int64_t o_bs, o_hs, o_rs;
    const bf16* dO; int64_t do_bs, do_hs, do_rs;
    bf16* dQ; int64_t dq_bs, dq_hs, dq_rs;
    bf16* dK; int64_t dk_bs, dk_hs, dk_rs;
    bf16* dV; int64_t dv_bs, dv_hs, dv_rs;
    float* LSE;          // [B, Hq, Spad]
    const float* delta;  // [B, Hq, Spad]
    const int* kv_len;
    const int* kv_lo;    // causal only: first visible key per sample (left-padded batches); keys [kv_lo[b], kv_len[b])
    int B, Hq, Hkv, S, Spad;
    float scale;
    int causal;
    int dbg;          // AFK_ATTN_DBG experiments: read only in -DAFK_PROBES builds (AFK_DBG below), ignored otherwise
    int split_heads;  // dK/dV sweep (GQA): P > 0 = P blocks per kv head, each sweeping ~group / P query heads into one partial dK/dV, reduced afterwards; 0 = one block per kv head
    int wide;         // 16-byte epilogue stores are legal (every output pointer / stride keeps 16-byte alignment)
    int xcd_map;      // forward / dQ: 1-D grid with the XCD-aware block -> (sample, head, query block) map below; 0 = linear (heads, batch, blocks) order
    int nz;           // forward / dQ: number of 128-query blocks
    // afk_attn2_bwd_fused_rope: the backward of the rotary embedding (rotation by -theta, the oracle's bf16 rounding points: afk_rope_inplace backward) applied to
    // dQ in the dQ kernel's epilogue and to dK where its final value is formed (GQA reduce / the sweep's epilogue) - no separate pass over dq | dk
    const bf16* rope_cos;   // [positions, D] bf16 tables (null: no rotation)
    const bf16* rope_sin;
    const int* rope_pos;    // [B * S] position of every row (null: row % S)
    // lane-major copies of the tables (nullable; only read when rope_pos is null): element ((((rb * 2 + h2) * (D / 16) + j) * 2 + hi) * 32 + l31) * 4 + e =
    // table[32 rb + l31][h2 * D/2 + 8 j + 4 hi + e] - what lane (l31, hi) of a wave on rows 32 rb .. 32 rb + 31 needs for accumulator piece j is 8 bytes next
    // to its neighbours' (a row-per-lane read of the [positions, D] table touches 32 cache lines per instruction and re-requests every line 8 times)
    const bf16* rope_cos_lanes;
    const bf16* rope_sin_lanes;
    int paired;   // attn_fwd_persist_kernel: 1 = no work queue - block k runs item k, then item total - 1 - k (the paired-tile causal schedule of VERDICT r04 / r05:
                  // with items ordered longest first every block sweeps the same number of key tiles), grid = ceil(total / 2)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr float NEG_INF = -INFINITY;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define ROW_OF(r, hi) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hi))

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// exchange with lane^32 through v_permlane32_swap (no LDS round trip): returns the partner half's value
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = {lo: x_lo, hi: x_lo}, r[1] = {lo: x_hi, hi: x_hi}
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// Epilogue store of one 32 (rows) x 32 (columns) MFMA result block, row-per-lane: lane (l31, hi) holds its row's columns 8q + 4hi + {0..3} in
// accumulator registers 4q..4q+3, i.e. four 8-byte pieces 16 bytes apart.  WIDE: the two lane halves (same row) trade registers through
// v_permlane32_swap so that each lane owns 8 CONSECUTIVE columns (16t + 8hi + 0..7): two 16-byte stores instead of four 8-byte ones.  The
// store tail of these short blocks is issue-bound (MI355X_MICROARCH: row-per-lane stores at a row stride; halving the instruction count at
// equal bytes halves it), and at S = 1024 a block lives for only 2-16 key tiles.  rowp = &out[row][32-column block].
template <bool WIDE>
__device__ __forceinline__ void store_block32(bf16* rowp, const f32x16& acc, float mul, int hi) {
    if constexpr (WIDE) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * t + e]), __float_as_uint(acc[8 * t + 4 + e]), false, false);
                o[e] = (bf16)(__uint_as_float(r[0]) * mul);       // lo lanes: own q = 2t;       hi lanes: partner's q = 2t + 1
                o[4 + e] = (bf16)(__uint_as_float(r[1]) * mul);   // lo lanes: partner's q = 2t; hi lanes: own q = 2t + 1
            }
            *(bf16x8*)(rowp + 16 * t + 8 * hi) = o;
        }
    } else {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[4 * qd + e] * mul);
            *(bf16x4*)(rowp + 8 * qd + 4 * hi) = o;
        }
    }
}

// RoPE backward on the DT accumulator blocks of ONE row held row-per-lane (store_block32's layout: block dt, register 4 qd + e <-> column 32 dt + 8 qd + 4 hi + e;
// the rotate-half partner d + D/2 is block dt + DT/2, same register, same lane).  Rounding points of rope_kernel (elementwise.hip) with sign = -1: the incoming
// gradient is a bf16 tensor (acc x mul rounded), every product is rounded, the sum is rounded - so the accumulators leave as exactly-representable bf16 values
// and the caller stores them with mul = 1.  ctab / stab = the row's cos / sin table rows (D entries each).
// LANES: ctab / stab point at the wave's 32-row block of the lane-major tables (AttnArgs2::rope_cos_lanes) and l31 selects the lane's row
template <int DT, bool LANES = false>
__device__ __forceinline__ void rope_bwd_rows(f32x16 (&acc)[DT], float mul, const bf16* ctab, const bf16* stab, int hi, int l31 = 0) {
    constexpr int HB = DT / 2;   // blocks per half head
#pragma unroll
    for (int dt = 0; dt < HB; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int d0 = 32 * dt + 8 * qd + 4 * hi;
            bf16x4 c1, s1, c2, s2;
            if constexpr (LANES) {
                const int o1 = (((dt * 4 + qd) * 2 + hi) * 32 + l31) * 4, o2 = o1 + 2 * DT * 2 * 32 * 4;   // h2 = 1: + (D / 16) pieces x 2 x 32 x 4
                c1 = *(const bf16x4*)(ctab + o1); s1 = *(const bf16x4*)(stab + o1);
                c2 = *(const bf16x4*)(ctab + o2); s2 = *(const bf16x4*)(stab + o2);
            } else {
                c1 = *(const bf16x4*)(ctab + d0); s1 = *(const bf16x4*)(stab + d0);
                c2 = *(const bf16x4*)(ctab + 16 * DT + d0); s2 = *(const bf16x4*)(stab + 16 * DT + d0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = rbf_strict(acc[dt][4 * qd + e] * mul), b = rbf_strict(acc[dt + HB][4 * qd + e] * mul);   // rbf_strict: see common.h (fp-contract)
                acc[dt][4 * qd + e] = rbf_strict(rbf_strict(a * (float)c1[e]) + rbf_strict(b * (float)s1[e]));
                acc[dt + HB][4 * qd + e] = rbf_strict(rbf_strict(b * (float)c2[e]) + rbf_strict(-a * (float)s2[e]));
            }
        }
}

// Buffer parity of a tile: DynPar = run-time (boundary tiles, the odd tile in front of the unrolled interior loop), StaticPar<P> = compile time
// - the interior loops are unrolled by two so that EVERY LDS address of a tile is "lane-constant register + immediate" (the buffer offset folded
// into the 16-bit ds offset field): the run-time form paid 16 v_add_u32 per 64-key tile for it (round 4, ISA count).
struct DynPar { int par; };
template <int P> struct StaticPar { static constexpr int par = P; };
template <typename T> struct is_static_par : std::false_type {};
template <int P> struct is_static_par<StaticPar<P>> : std::true_type {};
template <typename P>
__host__ __device__ constexpr int par_static_off(int bytes) {   // compile-time part of the buffer offset (0 for DynPar)
    if constexpr (is_static_par<P>::value) return P::par * bytes;
    else return 0;
}
template <typename P>
__device__ __forceinline__ uint32_t par_dyn_off(const P& p, int bytes) {   // run-time part (0 for StaticPar)
    if constexpr (is_static_par<P>::value) return 0u;
    else return (uint32_t)(p.par * bytes);
}

template <int D>
__device__ __forceinline__ int swz(int r) {
    if (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

template <int D>
struct Tile {
    static constexpr int RS = 2 * D;             // row stride, bytes
    static constexpr int BYTES = 64 * RS;        // one 64-row image
    static constexpr int UNITS = BYTES / 1024;   // LDS-DMA pieces per image (16 / 8)
    static constexpr int RPU = 1024 / RS;        // rows per piece (4 / 8)
    static constexpr int CPR = RS / 16;          // 16-byte chunks per row (16 / 8)
    static constexpr int KS = D / 16, DT = D / 32;

    // stage rows [row0, row0+64) of a [rows][D] tensor (row stride rs elements) into the image at `img`
    static __device__ __forceinline__ void stage(char* img, const bf16* base, int64_t rs, int row0, int max_row, int wave, int lane) {
        const int lrow = lane / CPR, pos = lane % CPR;
#pragma unroll
        for (int u0 = 0; u0 < UNITS / 4; ++u0) {
            const int u = wave + 4 * u0;
            const int r = u * RPU + lrow;
            const int chunk = pos ^ swz<D>(r);
            const int gr = min(row0 + r, max_row);
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + (int64_t)gr * rs + chunk * 8), (lds_void*)(img + u * 1024), 16, 0, 0);
        }
    }
    // Lane-constant fragment offsets.  Neither swizzle term depends on the 32-row half (kt2) or on the 16-key step (s4):
    // adding 32 (resp. 16) rows leaves r&3, (r>>1)&1 and (r>>2)&3 unchanged, so every fragment address is
    // "precomputed lane offset + compile-time immediate" - the loops issue ds_reads with no address VALU at all
    // (computing the swizzles per read cost more issue slots than the MFMAs they fed).
    struct Offs {
        int row[KS];      // row fragment (row = lane&31 of half 0), k-step ks
        int tr[DT][2];    // transposed fragment, d-tile dt, piece pc (s4 = 0)
    };
    static __device__ __forceinline__ Offs make_offs(int lane) {
        Offs o;
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) o.row[ks] = l31 * RS + (((2 * ks + hi) ^ swz<D>(l31)) << 4);
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int chunk = 4 * dt + 2 * (g & 1) + ((i & 3) >> 1);
                const int r = 4 * hi + (i >> 2) + 8 * pc;
                o.tr[dt][pc] = r * RS + ((chunk ^ swz<D>(r)) << 4) + 8 * (i & 1);
            }
        return o;
    }
    // row fragment: row 32*kt2 + (lane&31), d = 16*ks + 8*hi .. +8
    static __device__ __forceinline__ bf16x8 row_frag(const char* img, const Offs& o, int kt2, int ks) {
        return *(const bf16x8*)(img + o.row[ks] + kt2 * 32 * RS);
    }
    // transposed fragment: lane holds d = 32*dt + (lane&31), rows (keys) {16*s4 + 4hi + 0..3, 16*s4 + 8 + 4hi + 0..3}
    static __device__ __forceinline__ bf16x8 tr_frag(const char* img, const Offs& o, int dt, int s4) {
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o.tr[dt][0] + s4 * 16 * RS));
        const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o.tr[dt][1] + s4 * 16 * RS));
        bf16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        return v;
    }
};

// ------------------------------------------------------------------------------------------ block -> work map of the forward and dQ kernels (round 6)
// The hardware deals workgroups to the 8 XCDs round-robin in linear block order and every XCD has its own 4 MB L2.  With the grid (heads, batch, query
// blocks) of rounds 2-5 the heads that SHARE a K / V stream - the Hq / Hkv query heads of a GQA group, the query blocks of one head - landed on different XCDs:
// at the AF3 decoder shape every XCD streamed all 32 (sample, kv head) pairs (16 MB through a 4 MB L2), i.e. nearly every K / V tile load of the launch
// - 528 MB, 5.3 TB/s over the forward's 100 us - missed L2 and crossed the fabric; the encoder forward re-read each head's K / V 12 times (737 MB).
// Here block L runs on XCD L % 8 in slot L / 8, and the map gives every XCD a CONTIGUOUS chunk of the work sorted by what it streams:
//   causal (blocks differ in length): level-major - all blocks of query-block level z (longest first, the order that balances the grid), and inside a level
//     the heads of one (sample, kv head) group side by side: an XCD owns whole groups (AF3: 4 groups = 2 MB of K / V, L2-resident for the whole launch;
//     B = 1: one group per XCD pair, levels alternating), the group's heads run in lockstep over the same tiles;
//   non-causal (equal blocks): head-major - the query blocks of a head back to back on one XCD, running concurrently over the same K / V tiles.
// When the counts do not divide by 8 the tail (or, non-causal, everything) keeps the linear order.  lvl = dispatch level (0 first).
struct QBlock { int b, h, z; };
__device__ __forceinline__ QBlock attn_qblock(const AttnArgs2& p) {
    const int nz = p.nz, n = p.Hq * p.B, gsz = p.Hq / p.Hkv;
    int L, item, lvl;
    if (!p.xcd_map) {
        return QBlock{(int)blockIdx.y, (int)blockIdx.x, p.causal ? nz - 1 - (int)blockIdx.z : (int)blockIdx.z};
    }
    L = blockIdx.x;
    const int T = n * nz;
    if (!p.causal) {
        if ((T & 7) == 0) {
            const int idx = (L & 7) * (T >> 3) + (L >> 3);
            item = idx / nz;
            lvl = idx - item * nz;
        } else {
            lvl = L / n;
            item = L - lvl * n;
        }
    } else {
        const int k = (n & 7) == 0 ? 1 : (n & 3) == 0 ? 2 : (n & 1) == 0 ? 4 : 8;   // levels per super-level: k n divisible by 8
        const int full = nz / k, c = (k * n) >> 3;                                   // complete super-levels; items per XCD and super-level
        if (L < full * k * n) {
            const int slot = L >> 3, sup = slot / c, r = slot - sup * c;
            const int idx = (L & 7) * c + r;                                         // position in the super-level's (group, level, head) order
            const int per_group = k * gsz, group = idx / per_group, rem = idx - group * per_group;
            const int li = rem / gsz;
            lvl = sup * k + li;
            item = (group / p.Hkv) * p.Hq + (group % p.Hkv) * gsz + (rem - li * gsz);
        } else {
            const int L2 = L - full * k * n;
            lvl = full * k + L2 / n;
            item = L2 % n;
        }
    }
    const int b = item / p.Hq;
    return QBlock{b, item - b * p.Hq, p.causal ? nz - 1 - lvl : lvl};
}

// ------------------------------------------------------------------------------------------ forward
// One block = 128 query rows (4 waves x 32), two blocks per CU.  Per 64-key tile and wave:
//     S^T = K.Q^T (2 x KS MFMAs, two independent accumulators)  ->  one online-softmax step over all 64 keys
//     ->  O^T += V^T.P (4 x DT MFMAs, V^T fragments by ds_read_b64_tr_b16)
// The key loop is split into an INTERIOR part (every key of the tile visible to every row of the block: no mask, no liveness
// test - the only branches are the loop edge and the rare rescale) and a BOUNDARY part (causal diagonal / ragged tail): a taken
// branch costs an instruction-buffer refill, and the first version of this kernel spent ~30 % of its cycles on them.
// Lazy rescale: the running max only moves (and O, l are only rescaled) when some row's max grew by more than 2^8; until then
// probabilities are taken against the stale max (<= 2^8, exact in fp32 / same relative precision in bf16).  The row sum is kept
// per lane half and folded once at the end.  The V^T reads are issued (opaque asm, common.h) BEFORE the softmax arithmetic so
// their latency hides under it; the K(j+1)/V(j+1) LDS-DMA is in flight during the whole tile (vmcnt(0) only at the barrier).
#define AFK_ATTN_BARRIER()                                    \
    do {                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);                    \
        __builtin_amdgcn_s_barrier();                         \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)
// probe builds only (AFK_ATTN_DBG & 2, forward / dQ): the tile barrier WITHOUT the wait for the prefetch - tiles are read before they have landed (wrong
// results); the time it saves is the time the loop spends waiting for LDS-DMA
#define AFK_ATTN_BARRIER_P(p_)                                \
    do {                                                      \
        if (AFK_DBG(p_) & 16) {   /* no block barrier at all: how much do the four waves cost each other? (wrong results) */ \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
            __builtin_amdgcn_sched_barrier(0);                \
        } else if (AFK_DBG(p_) & 2) {                         \
            __builtin_amdgcn_sched_barrier(0);                \
            __builtin_amdgcn_s_barrier();                     \
            __builtin_amdgcn_sched_barrier(0);                \
        } else {                                              \
            AFK_ATTN_BARRIER();                               \
        }                                                     \
    } while (0)
constexpr float RESCALE_THR = 8.f;  // log2 domain

// LM (round 4): the row sum l comes out of the matrix pipe - one more accumulator block fed with an all-ones A fragment against the SAME
// probability fragments (l = P^T . 1: every row of the block equals the per-query sum over the tile's keys, it accumulates over tiles and takes
// the lazy rescale like O).  4 MFMAs per tile replace 29 VALU adds, and the normaliser is the sum of the bf16-ROUNDED probabilities that P.V uses.
// SCH (round 6): 0 = the K row fragments are plain C++ LDS loads (the compiler schedules them just in time: two ds_read_b128, s_waitcnt lgkmcnt(0..1), two
// MFMAs - every MFMA pair of the S^T phase starts by waiting out an LDS round trip) and the next tile's LDS-DMA pieces are issued in one burst in front of the
// tile; 1 = explicit two-deep ring of opaque ds_read_b128 groups (4 fragments = 4 MFMAs per group, the next group in flight behind the one being consumed,
// counted lgkmcnt) with the DMA pieces of the next tile dealt out between the MFMA groups (issue cost of a piece ~60-180 cycles: under the matrix pipe
// instead of in front of it).  Same MFMAs on the same operands in the same order: bit-identical results.
template <int D, bool LM, int SCH>
__global__ __launch_bounds__(256, 2) void attn_fwd_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][K image | V image]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    // grid = (heads, batch, position blocks): the hardware deals workgroups to the 8 XCDs round-robin in linear block order, so the FASTEST
    // grid dimension must not be the one the work per block depends on - with the 8 causal position blocks of S = 1024 in x, XCD k received
    // every block of length class k and the kernel ran as long as the XCD holding the 16-tile blocks (profiles/r02_attn_probes.md).
    // Position blocks are the slowest dimension, longest first.
    const QBlock blk = attn_qblock(p);   // causal: late query blocks sweep the most keys - they are dispatched first so the grid drains evenly
    const int b = blk.b, h = blk.h, hk = h / (p.Hq / p.Hkv);
    const int qb0 = blk.z * 128;
    const int q0 = qb0 + wave * 32;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const int kv_lo = p.kv_lo ? min(max(p.kv_lo[b], 0), kv_len) : 0;  // keys before it are padding (left-padded sample)

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qp + ks * 16);

    const bf16* Kbase = p.K + b * p.k_bs + hk * p.k_hs;
    const bf16* Vbase = p.V + b * p.v_bs + hk * p.v_hs;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
    float m = NEG_INF, l = 0.f;  // running (possibly stale) max in the log2 domain; row sum of THIS lane half (LM: lacc instead)
    f32x16 lacc = zero16();
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.f;
    const float c2 = p.scale * LOG2E;

    const int kv_end_blk = p.causal ? min(kv_len, min(qb0 + 128, p.S)) : kv_len;  // keys any row of the block can see
    const int ntiles = (kv_end_blk + 63) >> 6;
    // interior tiles: all 64 keys valid and visible to every row of the block
    const int n_int = p.causal ? min(qb0, kv_len) >> 6 : kv_len >> 6;
    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;  // this wave's horizon (boundary tiles)
    const uint32_t lds0 = afk_lds_addr(smem);
    uint32_t vtr[DT][2];  // V^T fragment addresses in buffer 0
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) vtr[dt][pc] = lds0 + T::BYTES + offs.tr[dt][pc];
    uint32_t krow[KS];   // SCH 1: K row-fragment addresses in buffer 0 (kt2 = 1: + 32 rows)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) krow[ks] = lds0 + offs.row[ks];

    auto stage = [&](int j) {  // general form (row clamp): prologue and boundary tiles
        char* buf = smem + (j & 1) * 2 * T::BYTES;
        T::stage(buf, Kbase, p.k_rs, j * 64, p.S - 1, wave, lane);
        T::stage(buf + T::BYTES, Vbase, p.v_rs, j * 64, p.S - 1, wave, lane);
    };
    // interior form: lane-constant source pointers of tile 0, advanced by whole tiles (no per-tile address arithmetic beyond one
    // 64-bit add per piece); only legal while all 64 rows of the staged tile exist
    // per-lane part of every source address = a tile-independent 32-bit byte offset (row r of the tile, swizzled chunk); the tile part is wave-uniform:
    // scalar base + VGPR offset addressing, no per-tile address VALU (the 64-bit per-lane pointers of rounds 1-3 cost 16 v_lshl_add_u64 per tile)
    constexpr int NP = T::UNITS / 4;
    uint32_t koff[NP], voff[NP];
#pragma unroll
    for (int u0 = 0; u0 < NP; ++u0) {
        const int u = wave + 4 * u0, r = u * T::RPU + lane / T::CPR, chunk = (lane % T::CPR) ^ swz<D>(r);
        koff[u0] = (uint32_t)(r * (int)p.k_rs + chunk * 8) * 2u;
        voff[u0] = (uint32_t)(r * (int)p.v_rs + chunk * 8) * 2u;
    }
    auto stage_fast = [&](int j) {
        const uint32_t buf = lds0 + (j & 1) * 2 * T::BYTES + wave * 1024;
        const bf16* k0 = Kbase + (int64_t)j * 64 * p.k_rs;
        const bf16* v0 = Vbase + (int64_t)j * 64 * p.v_rs;
#pragma unroll
        for (int u0 = 0; u0 < NP; ++u0) {
            // piece u0 = rows 4 RPU u0 further down with the SAME swizzled chunk (neither swizzle term depends on u0): one per-lane offset, the piece
            // part goes into the scalar base
            afk_dma16_saddr(k0 + (int64_t)u0 * 4 * T::RPU * p.k_rs, koff[0], buf + 4 * u0 * 1024);
            afk_dma16_saddr(v0 + (int64_t)u0 * 4 * T::RPU * p.v_rs, voff[0], buf + T::BYTES + 4 * u0 * 1024);
        }
    };
    const int n_full = p.S >> 6;  // tiles whose 64 rows all exist

    // one 64-key tile.  MASKED: per-element visibility (key < kv_len, causal key <= q) is applied to the scores.  par_: DynPar / StaticPar<P>.
    // DMA piece i (0 .. 2 NP - 1: K pieces even, V pieces odd) of tile jn into buffer jn & 1 - the pointer form of stage_fast, one piece at a time
    auto dma_piece = [&](int jn, auto i_) {
        constexpr int i = decltype(i_)::value, u0 = i >> 1;
        constexpr bool isv = (i & 1) != 0;
        const uint32_t dst = lds0 + (jn & 1) * 2 * T::BYTES + wave * 1024 + (isv ? T::BYTES : 0) + 4 * u0 * 1024;
        if constexpr (isv) afk_dma16_saddr(Vbase + ((int64_t)jn * 64 + u0 * 4 * T::RPU) * p.v_rs, voff[0], dst);
        else afk_dma16_saddr(Kbase + ((int64_t)jn * 64 + u0 * 4 * T::RPU) * p.k_rs, koff[0], dst);
    };
    // dma_: std::true_type = this tile issues the LDS-DMA of tile j + 1 (a full tile) itself, dealt out between its MFMA groups (SCH 1 only)
    auto tile = [&](int j, auto masked_, auto par_, auto dma_) {
        constexpr bool MASKED = decltype(masked_)::value;
        constexpr bool DMA = decltype(dma_)::value;
        constexpr int POFF = par_static_off<decltype(par_)>(2 * T::BYTES);           // compile-time buffer offset (folded into the ds offset fields)
        const uint32_t boff = par_dyn_off(par_, 2 * T::BYTES);                        // run-time buffer offset
        const char* kimg = smem + POFF + boff;
        f32x16 st[2] = {zero16(), zero16()};
        if constexpr (SCH == 0) {
            static_assert(!DMA || SCH == 1, "interleaved DMA belongs to the ring schedule");
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[0] = MFMA(T::row_frag(kimg, offs, 0, ks), qf[ks], st[0]);
                st[1] = MFMA(T::row_frag(kimg, offs, 1, ks), qf[ks], st[1]);
            }
        } else {
            constexpr int NGK = KS / 2, PPG = 2 * NP / NGK;   // K fragment groups (2 k-steps x 2 row halves each); DMA pieces per group
            bf16x8 ra[4], rb[4];
            auto issue_k = [&](auto g_, bf16x8(&dst)[4]) {
                constexpr int g = decltype(g_)::value;
                const uint32_t a0 = krow[2 * g] + boff, a1 = krow[2 * g + 1] + boff;
                dst[0] = afk_lds_b128<POFF>(a0);
                dst[1] = afk_lds_b128<POFF + 32 * T::RS>(a0);
                dst[2] = afk_lds_b128<POFF>(a1);
                dst[3] = afk_lds_b128<POFF + 32 * T::RS>(a1);
            };
            issue_k(std::integral_constant<int, 0>{}, ra);
            if constexpr (NGK > 1) issue_k(std::integral_constant<int, 1>{}, rb);
            afk_static_for<NGK>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                bf16x8(&cur)[4] = (g & 1) ? rb : ra;
                if constexpr (g + 1 < NGK) afk_lgkmcnt<4>();   // LDS returns in order: group g has landed, g + 1 may still be in flight
                else afk_lgkmcnt<0>();
                afk_lds_tie(cur[0], cur[1], cur[2], cur[3]);
                st[0] = MFMA(cur[0], qf[2 * g], st[0]);
                st[1] = MFMA(cur[1], qf[2 * g], st[1]);
                st[0] = MFMA(cur[2], qf[2 * g + 1], st[0]);
                st[1] = MFMA(cur[3], qf[2 * g + 1], st[1]);
                if constexpr (g + 2 < NGK) issue_k(std::integral_constant<int, g + 2>{}, cur);
                if constexpr (DMA)
                    if (!(AFK_DBG(p) & 1)) afk_static_for<PPG>([&](auto i_) { dma_piece(j + 1, std::integral_constant<int, g * PPG + decltype(i_)::value>{}); });
            });
        }
        // V^T fragments of d-tile 0: in flight during the softmax; the other d-tiles follow through the two-deep ring below
        bf16x8 fa[4], fb[4];
        auto issue = [&](auto g_, bf16x8(&dst)[4]) {
            constexpr int dt = decltype(g_)::value;
            const uint32_t a0 = vtr[dt][0] + boff, a1 = vtr[dt][1] + boff;
            afk_static_for<4>([&](auto s_) { constexpr int s4 = decltype(s_)::value; dst[s4] = afk_lds_tr_frag<POFF + s4 * 16 * T::RS>(a0, a1); });
        };
        issue(std::integral_constant<int, 0>{}, fa);
        if (MASKED) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j * 64 + kt2 * 32 + ROW_OF(r, hi);
                    const bool dead = (key >= kv_len) || (key < kv_lo) || (p.causal && key > q);
                    st[kt2][r] = dead ? NEG_INF : st[kt2][r];
                }
        }
        // (no inline asm on MFMA results: the compiler's hazard recogniser does not see through asm and would not insert the
        //  MFMA-write -> VALU-read wait states; this file is built with -fno-honor-nans so the maxima fold to bare v_max3_f32)
        bf16x8 pb[4];
        if (AFK_DBG(p) & 32) {   // probe builds only (wrong results): no softmax arithmetic - the scores go to the P.V MFMAs as they are.  What the serial VALU block costs.
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[2 * kt2 + (r >> 3)][r & 7] = (bf16)st[kt2][r];
            m = 0.f;
        } else {
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);
        mx = fmaxf(mx, other_half(mx)) * c2;  // c2 > 0
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > m + RESCALE_THR) != 0, 0)) {  // first tile (m = -inf), then rare
            asm volatile("" ::);  // keeps this a real (almost never taken) branch: the compiler would otherwise speculate the O rescale
            const float m_new = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - ((m_new == NEG_INF) ? 0.f : m_new));  // m = -inf -> 0
            if (LM) lacc[0] *= alpha;   // only register 0 of the all-equal-rows block is ever read
            else l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m = m_new;
        }
        const float nm = (m == NEG_INF) ? 0.f : -m;
        const f32x2 c2v = {c2, c2}, nmv = {nm, nm};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {  // packed fp32: v_pk_fma_f32 / v_pk_add_f32
                const f32x2 s2 = {st[kt2][r], st[kt2][r + 1]};
                const f32x2 t2 = __builtin_elementwise_fma(s2, c2v, nmv);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                if (!LM) rs2 += p2;
                pb[2 * kt2 + (r >> 3)][r & 7] = (bf16)p2[0];
                pb[2 * kt2 + (r >> 3)][(r & 7) + 1] = (bf16)p2[1];
            }
        if (!LM) l += rs2[0] + rs2[1];
        }
        if (LM) {   // l += sum over the tile's 64 keys of the (bf16) probabilities: rows of the ones fragment x P - ahead of the V^T ring, no LDS operand
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) lacc = MFMA(ones, pb[s4], lacc);
        }
        afk_frag_ring<DT>(issue, [&](auto g_, bf16x8(&f)[4]) {
            constexpr int dt = decltype(g_)::value;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) oacc[dt] = MFMA(f[s4], pb[s4], oacc[dt]);
        }, fa, fb);
    };

    // left-padded sample: tiles before j0 hold no visible key; the tile that contains kv_lo is a boundary tile ahead of the interior run
    const int j0 = kv_lo >> 6, j_int0 = (kv_lo + 63) >> 6;
    if (ntiles > j0) stage(j0);
    AFK_ATTN_BARRIER();
    // Round 6 (ISA reading): the Q rows were requested by plain global loads and their first use sits inside the tile loops.  The compiler's waitcnt
    // pass does not see the asm `vmcnt(0)` of AFK_ATTN_BARRIER, so it planted `s_waitcnt vmcnt(0)` in front of the FIRST MFMA of every loop body that
    // could be the first consumer - i.e. every other interior tile waited for the K / V prefetch it had issued a few instructions earlier.  Naming the
    // registers as asm operands HERE (all loads have landed: the barrier above drained them) puts the compiler's wait where it costs nothing.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    int j = j0;
    constexpr std::false_type NO_DMA{};
    constexpr std::integral_constant<bool, SCH == 1> IN_TILE{};   // SCH 1: the fast tiles issue the next tile's DMA themselves
    for (; j < min(j_int0, ntiles); ++j) {
        stage(min(j + 1, ntiles - 1));
        if (q0 < p.S && j * 64 < kv_end) tile(j, std::true_type{}, DynPar{j & 1}, NO_DMA);
        AFK_ATTN_BARRIER();
    }
    const int n_fast = min(n_int, n_full - 1);  // tile j+1 must be a full tile for the pointer form of the prefetch
    if (j < n_fast && (j & 1)) {                // odd tile in front of the unrolled pairs
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, DynPar{1}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
        ++j;
    }
    for (; j + 1 < n_fast; j += 2) {            // interior pairs: buffer parity is a compile-time constant
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, StaticPar<0>{}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
        if constexpr (SCH == 0) stage_fast(j + 2);
        tile(j + 1, std::false_type{}, StaticPar<1>{}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
    }
    for (; j < n_fast; ++j) {
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, DynPar{j & 1}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
    }
    for (; j < n_int; ++j) {
        stage(min(j + 1, ntiles - 1));  // a redundant re-stage of the last tile lands in the other buffer and is never read
        tile(j, std::false_type{}, DynPar{j & 1}, NO_DMA);
        AFK_ATTN_BARRIER();
    }
    for (; j < ntiles; ++j) {
        stage(min(j + 1, ntiles - 1));
        if (q0 < p.S && j * 64 < kv_end) tile(j, std::true_type{}, DynPar{j & 1}, NO_DMA);  // wave-uniform
        AFK_ATTN_BARRIER();
    }
    if (q < p.S) {
        if (LM) l = lacc[0];              // the full-row sum already (the MFMA reduced over all 64 keys of a tile): no lane-half fold
        else l += other_half(l);
        const float inv = (l > 0.f) ? 1.f / l : 0.f;
        bf16* Op = p.O + b * p.o_bs + h * p.o_hs + (int64_t)q * p.o_rs;
        if (p.wide) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) store_block32<true>(Op + dt * 32, oacc[dt], inv, hi);
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) store_block32<false>(Op + dt * 32, oacc[dt], inv, hi);
        }
        // internal to the v2 kernels: MINUS the log-sum-exp in SCORE units, P = 2^(c2 (S + LSE)); +inf for a row that saw no key
        if (hi == 0 && p.LSE) p.LSE[((int64_t)b * p.Hq + h) * p.Spad + q] = (l > 0.f) ? -(m + __log2f(l)) / c2 : INFINITY;
    } else if (hi == 0 && p.LSE && q < p.Spad) {
        p.LSE[((int64_t)b * p.Hq + h) * p.Spad + q] = 0.f;   // padding tail [S, Spad) of a ragged last tile: the dK/dV sweep reads whole 64-query strips (no host-side fill)
    }
}

// ------------------------------------------------------------------------------------------ forward, PERSISTENT form (round 5, opt-in: AFK_ATTN_PERSIST=1)
// At S = 1024 a forward block lives for 2-16 key tiles and pays ~6.8 us of fixed cost per block against 2.3 us per tile at long S: Q rows and
// the first K / V tile arrive cold, the block is dispatched, the store tail drains.  Here min(items, slots) blocks stay resident and pull work items -
// (sample, head, 128-query block), heavy first, the order the grid of attn_fwd_lds_kernel dispatches - from an atomic queue.  The hand-over between two
// items hides the cold loads: the next item's first K / V tile is DMA'd into buffer 0 at the START of the current item's last tile (that tile lives in
// buffer 1: the tile count is even), its Q rows are requested right after the last tile (qf is dead there), BEFORE the O / LSE stores - so the stores of item
// i drain under tile 0 of item i + 1 (the compiler's counted vmcnt for qf skips the younger stores; nothing waits for vmcnt(0) until the barrier after tile 0).
// Restrictions (the launcher falls back to attn_fwd_lds_kernel otherwise): no kv_len / kv_lo, S % 128 == 0 (every tile full, an even tile count per item).
// Same arithmetic per item, bit for bit (same tile function, same order).  queue[0] = next item, queue[1] = blocks that have left; the last block resets both.
template <int D, bool LM>
__global__ __launch_bounds__(256, 2) void attn_fwd_persist_kernel(AttnArgs2 p, int* __restrict__ queue, int total_items) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_next;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const int nz = p.S >> 7, HB = p.Hq * p.B, group = p.Hq / p.Hkv;
    const float c2 = p.scale * LOG2E;
    const uint32_t lds0 = afk_lds_addr(smem);
    uint32_t vtr[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) vtr[dt][pc] = lds0 + T::BYTES + offs.tr[dt][pc];
    constexpr int NP = T::UNITS / 4;
    uint32_t koff[NP], voff[NP];
#pragma unroll
    for (int u0 = 0; u0 < NP; ++u0) {
        const int u = wave + 4 * u0, r = u * T::RPU + lane / T::CPR, chunk = (lane % T::CPR) ^ swz<D>(r);
        koff[u0] = (uint32_t)(r * (int)p.k_rs + chunk * 8) * 2u;
        voff[u0] = (uint32_t)(r * (int)p.v_rs + chunk * 8) * 2u;
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.f;

    struct Item { int b, h, qb0; const bf16 *Kb, *Vb; };
    auto decode = [&](int it) -> Item {   // wave-uniform
        const int zr = it / HB, rem = it - zr * HB;
        Item x;
        x.h = rem % p.Hq;
        x.b = rem / p.Hq;
        x.qb0 = (p.causal ? nz - 1 - zr : zr) * 128;
        x.Kb = p.K + x.b * p.k_bs + (x.h / group) * p.k_hs;
        x.Vb = p.V + x.b * p.v_bs + (x.h / group) * p.v_hs;
        return x;
    };
    auto stage_tile = [&](const Item& x, int j, int par) {
        const uint32_t buf = lds0 + par * 2 * T::BYTES + wave * 1024;
        const bf16* k0 = x.Kb + (int64_t)j * 64 * p.k_rs;
        const bf16* v0 = x.Vb + (int64_t)j * 64 * p.v_rs;
#pragma unroll
        for (int u0 = 0; u0 < NP; ++u0) {
            // piece u0 = rows 4 RPU u0 further down with the SAME swizzled chunk (neither swizzle term depends on u0): one per-lane offset, the piece
            // part goes into the scalar base
            afk_dma16_saddr(k0 + (int64_t)u0 * 4 * T::RPU * p.k_rs, koff[0], buf + 4 * u0 * 1024);
            afk_dma16_saddr(v0 + (int64_t)u0 * 4 * T::RPU * p.v_rs, voff[0], buf + T::BYTES + 4 * u0 * 1024);
        }
    };
    bf16x8 qf[KS];
    auto load_q = [&](const Item& x) {
        const bf16* Qp = p.Q + x.b * p.q_bs + x.h * p.q_hs + (int64_t)(x.qb0 + wave * 32 + l31) * p.q_rs + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qp + ks * 16);
    };

    Item cur = decode(blockIdx.x);
    load_q(cur);
    stage_tile(cur, 0, 0);
    AFK_ATTN_BARRIER();
    int pass = 0;
    while (true) {
        if (threadIdx.x == 0) {   // read >= 2 barriers later
            if (p.paired) s_next = (pass == 0 && total_items - 1 - (int)blockIdx.x != (int)blockIdx.x) ? total_items - 1 - (int)blockIdx.x : total_items;
            else s_next = (int)gridDim.x + __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++pass;
        const int q = cur.qb0 + wave * 32 + l31;
        const int ntiles = p.causal ? (cur.qb0 + 128) >> 6 : p.S >> 6;
        const int n_int = p.causal ? cur.qb0 >> 6 : ntiles;
        const int kv_end = p.causal ? cur.qb0 + wave * 32 + 32 : p.S;   // this wave's horizon
        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
        float m = NEG_INF, l = 0.f;
        f32x16 lacc = zero16();

        auto tile = [&](int j, auto masked_, auto par_) {
            constexpr bool MASKED = decltype(masked_)::value;
            constexpr int POFF = par_static_off<decltype(par_)>(2 * T::BYTES);
            const uint32_t boff = par_dyn_off(par_, 2 * T::BYTES);
            const char* kimg = smem + POFF + boff;
            f32x16 st[2] = {zero16(), zero16()};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[0] = MFMA(T::row_frag(kimg, offs, 0, ks), qf[ks], st[0]);
                st[1] = MFMA(T::row_frag(kimg, offs, 1, ks), qf[ks], st[1]);
            }
            bf16x8 fa[4], fb[4];
            auto issue = [&](auto g_, bf16x8(&dst)[4]) {
                constexpr int dt = decltype(g_)::value;
                const uint32_t a0 = vtr[dt][0] + boff, a1 = vtr[dt][1] + boff;
                afk_static_for<4>([&](auto s_) { constexpr int s4 = decltype(s_)::value; dst[s4] = afk_lds_tr_frag<POFF + s4 * 16 * T::RS>(a0, a1); });
            };
            issue(std::integral_constant<int, 0>{}, fa);
            if (MASKED) {
#pragma unroll
                for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j * 64 + kt2 * 32 + ROW_OF(r, hi);
                        st[kt2][r] = (p.causal && key > q) ? NEG_INF : st[kt2][r];
                    }
            }
            float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);
            mx = fmaxf(mx, other_half(mx)) * c2;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > m + RESCALE_THR) != 0, 0)) {
                asm volatile("" ::);
                const float m_new = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - ((m_new == NEG_INF) ? 0.f : m_new));
                if (LM) lacc[0] *= alpha;
                else l *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m = m_new;
            }
            const float nm = (m == NEG_INF) ? 0.f : -m;
            const f32x2 c2v = {c2, c2}, nmv = {nm, nm};
            bf16x8 pb[4];
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 s2 = {st[kt2][r], st[kt2][r + 1]};
                    const f32x2 t2 = __builtin_elementwise_fma(s2, c2v, nmv);
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                    if (!LM) rs2 += p2;
                    pb[2 * kt2 + (r >> 3)][r & 7] = (bf16)p2[0];
                    pb[2 * kt2 + (r >> 3)][(r & 7) + 1] = (bf16)p2[1];
                }
            if (!LM) l += rs2[0] + rs2[1];
            if (LM) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) lacc = MFMA(ones, pb[s4], lacc);
            }
            afk_frag_ring<DT>(issue, [&](auto g_, bf16x8(&f)[4]) {
                constexpr int dt = decltype(g_)::value;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) oacc[dt] = MFMA(f[s4], pb[s4], oacc[dt]);
            }, fa, fb);
        };

        // tiles in pairs (tile j in buffer j & 1; ntiles and n_int are even): every tile prefetches its successor, the LAST tile of the item (odd index:
        // buffer 1) prefetches tile 0 of the NEXT item into buffer 0 instead - that buffer is free once the barrier behind tile ntiles - 2 has passed
        int next = 0;
        bool has_next = false;
        Item nx = cur;
        auto second_prefetch = [&](int j) {   // issued in front of tile j + 1
            if (j + 2 < ntiles) {
                stage_tile(cur, j + 2, 0);
            } else {
                next = s_next;
                has_next = next < total_items;
                if (has_next) {
                    nx = decode(next);
                    stage_tile(nx, 0, 0);
                }
            }
        };
        int j = 0;
        for (; j < n_int; j += 2) {          // every key visible to every row
            stage_tile(cur, j + 1, 1);
            tile(j, std::false_type{}, StaticPar<0>{});
            AFK_ATTN_BARRIER();
            second_prefetch(j);
            tile(j + 1, std::false_type{}, StaticPar<1>{});
            AFK_ATTN_BARRIER();
        }
        if (j < ntiles) {                     // causal: the diagonal pair (the last one)
            stage_tile(cur, j + 1, 1);
            if (j * 64 < kv_end) tile(j, std::true_type{}, StaticPar<0>{});
            AFK_ATTN_BARRIER();
            second_prefetch(j);
            if ((j + 1) * 64 < kv_end) tile(j + 1, std::true_type{}, StaticPar<1>{});
            AFK_ATTN_BARRIER();
        }
        // epilogue of `cur`; the next item's Q rows are requested first
        const int qcur = q, bcur = cur.b, hcur = cur.h;
        if (has_next) {
            cur = nx;
            load_q(cur);
        }
        {
            if (LM) l = lacc[0];
            else l += other_half(l);
            const float inv = (l > 0.f) ? 1.f / l : 0.f;
            bf16* Op = p.O + bcur * p.o_bs + hcur * p.o_hs + (int64_t)qcur * p.o_rs;
            if (p.wide) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) store_block32<true>(Op + dt * 32, oacc[dt], inv, hi);
            } else {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) store_block32<false>(Op + dt * 32, oacc[dt], inv, hi);
            }
            if (hi == 0 && p.LSE) p.LSE[((int64_t)bcur * p.Hq + hcur) * p.Spad + qcur] = (l > 0.f) ? -(m + __log2f(l)) / c2 : INFINITY;
        }
        if (!has_next) break;
    }
    if (threadIdx.x == 0 && !p.paired) {
        const int gone = __hip_atomic_fetch_add(queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == (int)gridDim.x - 1) {   // every block has taken its last ticket: leave the queue ready for the next launch (stream order does the rest)
            __hip_atomic_store(queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(queue + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ
// Same geometry and loop structure as the forward (interior / boundary key tiles, opaque tr-reads issued ahead of the VALU block,
// K/V prefetch in flight for the whole tile).  Per 64-key tile and wave: S^T = K.Q^T and dP^T = V.dO^T (4 x KS MFMAs on four
// independent accumulators), dS = P o (dP - delta) with P = 2^(S*c2 - lse), dQ^T += K^T.dS (4 x DT MFMAs).
// SCH (round 6): 0 = compiler-scheduled row-fragment reads.  At head_dim 128 that kernel sat at 256 VGPRs with two spilled registers, and the allocator had
// serialised the first phase to ONE ds_read_b128 in flight: 32 x (read, s_waitcnt lgkmcnt(0), MFMA) per tile - the matrix pipe waited out an LDS round trip
// per MFMA.  1 = the 64 keys of a tile are processed as two 32-key halves (S^T and dP^T accumulators: 2 x 16 registers instead of 4 x 16), row fragments
// through an explicit two-deep ring of opaque ds_read_b128 groups (K, V of two k-steps per group) with counted lgkmcnt, the next tile's LDS-DMA pieces
// dealt out between the MFMA groups.  Same MFMAs on the same operands, same accumulation order per accumulator: bit-identical dQ.
template <int D, int SCH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const QBlock blk = attn_qblock(p);   // the forward kernel's block -> (sample, head, query block) map
    const int b = blk.b, h = blk.h, hk = h / (p.Hq / p.Hkv);
    const int qb0 = blk.z * 128;
    const int q0 = qb0 + wave * 32;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const int kv_lo = p.kv_lo ? min(max(p.kv_lo[b], 0), kv_len) : 0;  // keys before it are padding (left-padded sample)

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    const bf16* dOp = p.dO + b * p.do_bs + h * p.do_hs + (int64_t)qc * p.do_rs + hi * 8;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const bf16x8*)(Qp + ks * 16);
        dof[ks] = *(const bf16x8*)(dOp + ks * 16);
    }
    const float c2 = p.scale * LOG2E;
    const float nlse = p.LSE[((int64_t)b * p.Hq + h) * p.Spad + qc] * c2;  // stored negated, in score units (see the forward epilogue)
    float ndlt;
    if (p.O != nullptr) {
        // afk_attn2_bwd_fused: delta = rowsum(dO o O) is computed HERE - this wave holds its queries' dO rows already - and published for the
        // dK/dV sweep, which is launched behind this kernel: the separate pass over O and dO (attn2_delta_kernel) disappears.
        // Lane (l31, hi) covers d = 16 ks + 8 hi + 0..7 of row q: the two lane halves together cover the whole row.
        const bf16* Op = p.O + b * p.o_bs + h * p.o_hs + (int64_t)qc * p.o_rs + hi * 8;
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 o8 = *(const bf16x8*)(Op + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)o8[e] * (float)dof[ks][e];
        }
        acc += other_half(acc);
        ndlt = -acc;   // stored negated: the backward kernels start the dP accumulators from it
        if (hi == 0 && q < p.Spad) const_cast<float*>(p.delta)[((int64_t)b * p.Hq + h) * p.Spad + q] = (q < p.S) ? ndlt : 0.f;   // the tail [S, Spad) must read zero
    } else {
        ndlt = p.delta[((int64_t)b * p.Hq + h) * p.Spad + qc];      // stored negated (attn2_delta_kernel)
    }
    const bf16* Kbase = p.K + b * p.k_bs + hk * p.k_hs;
    const bf16* Vbase = p.V + b * p.v_bs + hk * p.v_hs;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = zero16();

    const int kv_end_blk = p.causal ? min(kv_len, min(qb0 + 128, p.S)) : kv_len;
    const int ntiles = (kv_end_blk + 63) >> 6;
    const int n_int = p.causal ? min(qb0, kv_len) >> 6 : kv_len >> 6;
    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;
    const int n_full = p.S >> 6;
    const uint32_t lds0 = afk_lds_addr(smem);
    uint32_t ktr[DT][2];  // K^T fragment addresses in buffer 0
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) ktr[dt][pc] = lds0 + offs.tr[dt][pc];

    auto stage = [&](int j) {
        char* buf = smem + (j & 1) * 2 * T::BYTES;
        T::stage(buf, Kbase, p.k_rs, j * 64, p.S - 1, wave, lane);
        T::stage(buf + T::BYTES, Vbase, p.v_rs, j * 64, p.S - 1, wave, lane);
    };
    constexpr int NP = T::UNITS / 4;   // scalar base + per-lane 32-bit offset (see the forward kernel)
    uint32_t koff[NP], voff[NP];
#pragma unroll
    for (int u0 = 0; u0 < NP; ++u0) {
        const int u = wave + 4 * u0, r = u * T::RPU + lane / T::CPR, chunk = (lane % T::CPR) ^ swz<D>(r);
        koff[u0] = (uint32_t)(r * (int)p.k_rs + chunk * 8) * 2u;
        voff[u0] = (uint32_t)(r * (int)p.v_rs + chunk * 8) * 2u;
    }
    auto stage_fast = [&](int j) {
        const uint32_t buf = lds0 + (j & 1) * 2 * T::BYTES + wave * 1024;
        const bf16* k0 = Kbase + (int64_t)j * 64 * p.k_rs;
        const bf16* v0 = Vbase + (int64_t)j * 64 * p.v_rs;
#pragma unroll
        for (int u0 = 0; u0 < NP; ++u0) {
            // piece u0 = rows 4 RPU u0 further down with the SAME swizzled chunk (neither swizzle term depends on u0): one per-lane offset, the piece
            // part goes into the scalar base
            afk_dma16_saddr(k0 + (int64_t)u0 * 4 * T::RPU * p.k_rs, koff[0], buf + 4 * u0 * 1024);
            afk_dma16_saddr(v0 + (int64_t)u0 * 4 * T::RPU * p.v_rs, voff[0], buf + T::BYTES + 4 * u0 * 1024);
        }
    };

    uint32_t krow[KS];   // SCH 1: K row-fragment addresses in buffer 0 (V: + T::BYTES, second 32-key half: + 32 rows)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) krow[ks] = lds0 + offs.row[ks];
    auto dma_piece = [&](int jn, auto i_) {   // piece i (K even, V odd) of tile jn into buffer jn & 1, as in the forward kernel
        constexpr int i = decltype(i_)::value, u0 = i >> 1;
        constexpr bool isv = (i & 1) != 0;
        const uint32_t dst = lds0 + (jn & 1) * 2 * T::BYTES + wave * 1024 + (isv ? T::BYTES : 0) + 4 * u0 * 1024;
        if constexpr (isv) afk_dma16_saddr(Vbase + ((int64_t)jn * 64 + u0 * 4 * T::RPU) * p.v_rs, voff[0], dst);
        else afk_dma16_saddr(Kbase + ((int64_t)jn * 64 + u0 * 4 * T::RPU) * p.k_rs, koff[0], dst);
    };
    auto tile = [&](int j, auto masked_, auto par_, auto dma_) {
        constexpr bool MASKED = decltype(masked_)::value;
        constexpr bool DMA = decltype(dma_)::value;
        constexpr int POFF = par_static_off<decltype(par_)>(2 * T::BYTES);
        const uint32_t boff = par_dyn_off(par_, 2 * T::BYTES);
        const char* kimg = smem + POFF + boff;
        const char* vimg = kimg + T::BYTES;
        bf16x8 fa[4], fb[4];  // K^T fragments: d-tile 0 in flight during the VALU block, the rest through the ring
        auto issue = [&](auto g_, bf16x8(&dst)[4]) {
            constexpr int dt = decltype(g_)::value;
            const uint32_t a0 = ktr[dt][0] + boff, a1 = ktr[dt][1] + boff;
            afk_static_for<4>([&](auto s_) { constexpr int s4 = decltype(s_)::value; dst[s4] = afk_lds_tr_frag<POFF + s4 * 16 * T::RS>(a0, a1); });
        };
        const f32x2 c2v = {c2, c2}, nl = {nlse, nlse}, nd = {ndlt, ndlt};
        bf16x8 dsb[4];
        // dS of one 32-key half (kt2) from its score / dP accumulators
        auto ds_half = [&](int kt2, const f32x16& sh, const f32x16& dh) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 s2 = {sh[r], sh[r + 1]};
                const f32x2 t2 = __builtin_elementwise_fma(s2, c2v, nl);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                const f32x2 d2 = {dh[r], dh[r + 1]};
                f32x2 ds2 = p2 * (d2 + nd);
                if (MASKED) {
                    const int key = j * 64 + kt2 * 32 + ROW_OF(r, hi);  // r even: rows key, key+1
                    if ((key >= kv_len) || (key < kv_lo) || (p.causal && key > q)) ds2[0] = 0.f;
                    if ((key + 1 >= kv_len) || (key + 1 < kv_lo) || (p.causal && key + 1 > q)) ds2[1] = 0.f;
                }
                dsb[2 * kt2 + (r >> 3)][r & 7] = (bf16)ds2[0];
                dsb[2 * kt2 + (r >> 3)][(r & 7) + 1] = (bf16)ds2[1];
            }
        };
        if constexpr (SCH == 0) {
            f32x16 st[2] = {zero16(), zero16()}, dp[2] = {zero16(), zero16()};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[0] = MFMA(T::row_frag(kimg, offs, 0, ks), qf[ks], st[0]);
                st[1] = MFMA(T::row_frag(kimg, offs, 1, ks), qf[ks], st[1]);
                dp[0] = MFMA(T::row_frag(vimg, offs, 0, ks), dof[ks], dp[0]);
                dp[1] = MFMA(T::row_frag(vimg, offs, 1, ks), dof[ks], dp[1]);
            }
            issue(std::integral_constant<int, 0>{}, fa);
            ds_half(0, st[0], dp[0]);
            ds_half(1, st[1], dp[1]);
        } else {
            constexpr int NGK = KS / 2, NT = 2 * NGK, PPG = (2 * NP + NT - 1) / NT;   // groups per half, groups per tile, DMA pieces per group
            bf16x8 ra[4], rb[4];
            auto issue_r = [&](auto G_, bf16x8(&dst)[4]) {
                constexpr int G = decltype(G_)::value, h = G / NGK, g = G % NGK;
                const uint32_t a0 = krow[2 * g] + boff, a1 = krow[2 * g + 1] + boff;
                dst[0] = afk_lds_b128<POFF + h * 32 * T::RS>(a0);
                dst[1] = afk_lds_b128<POFF + T::BYTES + h * 32 * T::RS>(a0);
                dst[2] = afk_lds_b128<POFF + h * 32 * T::RS>(a1);
                dst[3] = afk_lds_b128<POFF + T::BYTES + h * 32 * T::RS>(a1);
            };
            issue_r(std::integral_constant<int, 0>{}, ra);
            issue_r(std::integral_constant<int, 1>{}, rb);
            afk_static_for<2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                f32x16 sh = zero16(), dh = zero16();
                afk_static_for<NGK>([&](auto g_) {
                    constexpr int g = decltype(g_)::value, G = h * NGK + g;
                    bf16x8(&cur)[4] = (G & 1) ? rb : ra;
                    if constexpr (G + 1 < NT) afk_lgkmcnt<4>();   // in-order LDS returns: group G has landed, G + 1 may be in flight
                    else afk_lgkmcnt<0>();
                    afk_lds_tie(cur[0], cur[1], cur[2], cur[3]);
                    sh = MFMA(cur[0], qf[2 * g], sh);
                    dh = MFMA(cur[1], dof[2 * g], dh);
                    sh = MFMA(cur[2], qf[2 * g + 1], sh);
                    dh = MFMA(cur[3], dof[2 * g + 1], dh);
                    if constexpr (G + 2 < NT) issue_r(std::integral_constant<int, G + 2>{}, cur);
                    if constexpr (DMA)
                        if (!(AFK_DBG(p) & 1)) afk_static_for<PPG>([&](auto i_) {
                            constexpr int i = G * PPG + decltype(i_)::value;
                            if constexpr (i < 2 * NP) dma_piece(j + 1, std::integral_constant<int, i>{});
                        });
                });
                if constexpr (h == 1) issue(std::integral_constant<int, 0>{}, fa);   // every row fragment consumed: the K^T ring starts under the VALU block
                ds_half(h, sh, dh);
            });
        }
        afk_frag_ring<DT>(issue, [&](auto g_, bf16x8(&f)[4]) {
            constexpr int dt = decltype(g_)::value;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) dqacc[dt] = MFMA(f[s4], dsb[s4], dqacc[dt]);
        }, fa, fb);
    };

    const int j0 = kv_lo >> 6, j_int0 = (kv_lo + 63) >> 6;  // left-padded sample, as in the forward kernel
    if (ntiles > j0) stage(j0);
    AFK_ATTN_BARRIER();
    int j = j0;
    constexpr std::false_type NO_DMA{};
    constexpr std::integral_constant<bool, SCH == 1> IN_TILE{};   // SCH 1: the fast tiles issue the next tile's DMA themselves
    for (; j < min(j_int0, ntiles); ++j) {
        stage(min(j + 1, ntiles - 1));
        if (q0 < p.S && j * 64 < kv_end) tile(j, std::true_type{}, DynPar{j & 1}, NO_DMA);
        AFK_ATTN_BARRIER();
    }
    const int n_fast = min(n_int, n_full - 1);
    if (j < n_fast && (j & 1)) {                // odd tile in front of the unrolled pairs
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, DynPar{1}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
        ++j;
    }
    for (; j + 1 < n_fast; j += 2) {            // interior pairs: buffer parity is a compile-time constant
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, StaticPar<0>{}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
        if constexpr (SCH == 0) stage_fast(j + 2);
        tile(j + 1, std::false_type{}, StaticPar<1>{}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
    }
    for (; j < n_fast; ++j) {
        if constexpr (SCH == 0) stage_fast(j + 1);
        tile(j, std::false_type{}, DynPar{j & 1}, IN_TILE);
        AFK_ATTN_BARRIER_P(p);
    }
    for (; j < n_int; ++j) {
        stage(min(j + 1, ntiles - 1));
        tile(j, std::false_type{}, DynPar{j & 1}, NO_DMA);
        AFK_ATTN_BARRIER();
    }
    for (; j < ntiles; ++j) {
        stage(min(j + 1, ntiles - 1));
        if (q0 < p.S && j * 64 < kv_end) tile(j, std::true_type{}, DynPar{j & 1}, NO_DMA);
        AFK_ATTN_BARRIER();
    }
    if (q < p.S) {
        bf16* dQp = p.dQ + b * p.dq_bs + h * p.dq_hs + (int64_t)q * p.dq_rs;
        float mul = p.scale;   // (softmax scale folded out of dS)
        if (p.rope_cos) {
            if (p.rope_cos_lanes && !p.rope_pos) {   // positions = row index: the wave's rows q0 .. q0 + 31 are ONE 32-row block of the lane-major tables
                const int64_t rb = (int64_t)(q0 >> 5) * 32 * D;
                rope_bwd_rows<DT, true>(dqacc, mul, p.rope_cos_lanes + rb, p.rope_sin_lanes + rb, hi, l31);
            } else {
                const int64_t pr = p.rope_pos ? p.rope_pos[(int64_t)b * p.S + q] : q;
                rope_bwd_rows<DT>(dqacc, mul, p.rope_cos + pr * D, p.rope_sin + pr * D, hi);
            }
            mul = 1.f;
        }
        if (p.wide) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) store_block32<true>(dQp + dt * 32, dqacc[dt], mul, hi);
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) store_block32<false>(dQp + dt * 32, dqacc[dt], mul, hi);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// block = 128 keys of one kv head (wave = 32 keys); streams 64-query tiles of Q and dO of every head of the GQA group.
// Per tile and wave: S = Q.K^T and dP = dO.V^T (4 x KS MFMAs, K/V of the wave's 32 keys live in registers), P / dS as in the dQ
// kernel (per-query lse / delta: two float4 loads per 8 queries, issued BEFORE the tile's LDS-DMA prefetch so that waiting for them
// never waits for the prefetch), dV^T += dO^T.P and dK^T += Q^T.dS (8 x DT MFMAs, transposed fragments by opaque tr-reads in two
// halves).  Interior tiles = every (query, key) pair of the tile visible; boundary = causal diagonal, ragged S, padded keys.
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dkdv_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][Q image | dO image]
    uint64_t rt_entry = 0, rt1 = 0;
    if (AFK_DBG(p) & 4) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt_entry)::"memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const int b = blockIdx.y, group = p.Hq / p.Hkv;  // grid = (heads, batch, key blocks), see the forward kernel; causal: key block 0 is the longest
    const int hy = blockIdx.x;
    // split_heads = P > 0: P blocks per kv head, block part = hy % P sweeps query heads [part * group / P, (part + 1) * group / P) of the group and writes
    // ONE partial dK / dV (round 6: P < group - fewer, longer blocks: a block's prologue + store tail were ~8 us against 2.5 us per tile - and P instead of
    // `group` partials through HBM); P = group is the one-block-per-query-head form of rounds 2-5
    const int P = p.split_heads;
    const int hk = P ? hy / P : hy;
    const int g_begin = P ? ((hy % P) * group) / P : 0;
    const int g_count = P ? (((hy % P) + 1) * group) / P - g_begin : group;
    const int kb0 = blockIdx.z * 128;
    const int key0 = kb0 + wave * 32;
    const int key = key0 + l31;
    const int keyc = min(key, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const int kv_lo = p.kv_lo ? min(max(p.kv_lo[b], 0), kv_len) : 0;  // keys before it are padding (left-padded sample)
    const bool wave_live = key0 < p.S;
    const bool key_dead = key >= kv_len || key < kv_lo;
    const float c2 = p.scale * LOG2E;

    const bf16* Kp = p.K + b * p.k_bs + hk * p.k_hs + (int64_t)keyc * p.k_rs + hi * 8;
    const bf16* Vp = p.V + b * p.v_bs + hk * p.v_hs + (int64_t)keyc * p.v_rs + hi * 8;
    bf16x8 kf[KS], vf[KS];  // this wave's 32 keys, resident for the whole sweep (head_dim 128 runs one wave per SIMD to afford it)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *(const bf16x8*)(Kp + ks * 16);
        vf[ks] = *(const bf16x8*)(Vp + ks * 16);
    }

    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dkacc[dt] = zero16();
        dvacc[dt] = zero16();
    }
    const int qt_begin = p.causal ? (kb0 >> 6) : 0;  // block-uniform first 64-query tile
    const int qt_end = (p.S + 63) >> 6;
    // a key block made of padding only (right of kv_len or left of kv_lo) sweeps nothing and writes zeros
    const bool block_dead = kb0 >= kv_len || kb0 + 128 <= kv_lo;
    const int ntiles = block_dead ? 0 : (qt_end - qt_begin) * g_count;
    // interior query tiles of one head: [qt_int0, qt_int1): all 64 queries exist, causal: every query >= every key of the block;
    // a block holding padded keys (kb0 + 128 > kv_len) has none
    const int qt_int0 = (kb0 + 128 > kv_len || kb0 < kv_lo) ? qt_end : (p.causal ? min((kb0 + 128 + 63) >> 6, qt_end) : 0);
    const int qt_int1 = max(p.S >> 6, qt_int0);
    const uint32_t lds0 = afk_lds_addr(smem);
    uint32_t qtr[DT][2];  // Q^T fragment addresses in buffer 0 (dO^T: + T::BYTES)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) qtr[dt][pc] = lds0 + offs.tr[dt][pc];

    constexpr int STATS = 2 * T::BYTES;      // offset of the strip inside a buffer
    constexpr int BUF = 2 * T::BYTES + 1024;  // buffer pitch
    // Staging tile (h, qt) into buffer `slot`: per-lane part of every source address is a tile-independent 32-bit byte offset (row r of
    // the tile, swizzled chunk), the tile part (batch, head, first row) is wave-uniform - scalar base + VGPR offset addressing, no
    // per-tile address VALU.  A ragged last tile (rows >= S) takes the row-clamping general form.
    constexpr int NP = T::UNITS / 4;
    uint32_t qoff[NP], dooff[NP];
#pragma unroll
    for (int u0 = 0; u0 < NP; ++u0) {
        const int u = wave + 4 * u0, r = u * T::RPU + lane / T::CPR, chunk = (lane % T::CPR) ^ swz<D>(r);
        qoff[u0] = (uint32_t)(r * (int)p.q_rs + chunk * 8) * 2u;
        dooff[u0] = (uint32_t)(r * (int)p.do_rs + chunk * 8) * 2u;
    }
    const float* stat_lane = ((lane & 16) ? p.delta : p.LSE) + (int64_t)b * p.Hq * p.Spad + (lane & 15) * 4;
    const int n_full = p.S >> 6;  // tiles whose 64 query rows all exist
    auto stage = [&](int slot, int h, int qt) {
        char* buf = smem + slot * BUF;
        const bf16* qb = p.Q + b * p.q_bs + h * p.q_hs;
        const bf16* dob = p.dO + b * p.do_bs + h * p.do_hs;
        if (qt < n_full) {
            const bf16* q0 = qb + (int64_t)qt * 64 * p.q_rs;
            const bf16* d0 = dob + (int64_t)qt * 64 * p.do_rs;
            const uint32_t dst = lds0 + slot * BUF + wave * 1024;
#pragma unroll
            for (int u0 = 0; u0 < NP; ++u0) {   // scalar base + per-lane 32-bit offset, issued as asm: no address VALU (common.h afk_dma16_saddr)
                afk_dma16_saddr(q0, qoff[u0], dst + 4 * u0 * 1024);
                afk_dma16_saddr(d0, dooff[u0], dst + T::BYTES + 4 * u0 * 1024);
            }
        } else {
            T::stage(buf, qb, p.q_rs, qt * 64, p.S - 1, wave, lane);
            T::stage(buf + T::BYTES, dob, p.do_rs, qt * 64, p.S - 1, wave, lane);
        }
        // lse / delta of the tile's 64 queries travel with the tile: one extra 1-KiB LDS-DMA piece (wave 0; lanes 0-15 fetch lse[64],
        // lanes 16-31 delta[64], the upper half duplicates them) into a stats strip behind the two images
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((gbl_void*)(stat_lane + (int64_t)h * p.Spad + qt * 64), (lds_void*)(buf + STATS), 16, 0, 0);
    };

    // Per tile and wave (one wave per SIMD for head_dim 128: everything below is ONE in-order instruction stream, so the order written
    // here is the overlap that happens):
    //   0. the accumulators of S and dP START at -lse (score units) and -delta, read from the stats strip: S' = Q.K^T - lse and
    //      dP' = dO.V^T - delta come out of the matrix pipe, P = 2^(c2 S') and dS = P o dP' are one multiply each;
    //   1. S' (2 KS MFMAs) then dP' (2 KS MFMAs), row fragments through a ring of opaque ds_read_b128 groups (4 fragments per group,
    //      RD groups deep: 8 reads in flight behind the group being consumed - two reads per lgkmcnt(0) reached a fifth of the LDS
    //      rate); the exponentials of S' are written between the dP' MFMAs, which they do not depend on;
    //   2. dV^T += dO^T.P (4 DT MFMAs, tr-read ring) with dS = P o dP' written between them, then dK^T += Q^T.dS (4 DT MFMAs).
    // ONE body for interior and boundary tiles (the mask is a wave-uniform branch around VALU-only code): with two inlined copies of
    // the MFMA chains the register allocator kept the 128 dK/dV accumulators of each copy in different AGPRs and moved all of them
    // back at the end of every tile (128 v_accvgpr_mov + the MFMA-drain s_nops in front of them, ~15 % of the tile).
    uint64_t c_start = 0, ts0 = 0, ts1 = 0, ts2 = 0, tsm = 0, sum_body = 0, sum_ph1 = 0, sum_bar = 0, rt0 = 0;
    const bool probe = (AFK_DBG(p) & 4) != 0;
    uint32_t qrow[KS];  // Q row-fragment addresses in buffer 0 (kt2 = 1: + 32 rows, dO: + T::BYTES)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qrow[ks] = lds0 + offs.row[ks];
    constexpr int NG = KS;                     // row-fragment groups: [0, NG/2) = Q (S'), [NG/2, NG) = dO (dP'); group = 2 k-steps
    constexpr int RD = (D == 128) ? 3 : 2;     // ring depth (head_dim 64 runs two waves per SIMD on half the registers)
    constexpr int PCH = 8 / NG;                // 8-value chunks of P per dO group (4 chunks per tile)
    constexpr int SCH = 4 / DT;                // 8-value chunks of dS per dV group
    auto body = [&](int t, int qt, bool masked) {
        const uint32_t boff = (t & 1) * BUF;
        const float* stats = (const float*)(smem + boff + STATS) + 4 * hi;
        f32x16 st[2], dp[2];
#pragma unroll
        for (int qt2 = 0; qt2 < 2; ++qt2)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 a = *(const f32x4*)(stats + qt2 * 32 + 8 * qd);
                const f32x4 d = *(const f32x4*)(stats + 64 + qt2 * 32 + 8 * qd);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[qt2][4 * qd + e] = a[e];
                    dp[qt2][4 * qd + e] = d[e];
                }
            }
        bf16x8 r0[4], r1[4], r2[4];
        auto rbuf = [&](auto i_) -> bf16x8(&)[4] {
            constexpr int i = decltype(i_)::value;
            if constexpr (i == 0) return r0;
            else if constexpr (i == 1) return r1;
            else return r2;
        };
        auto issue_rows = [&](auto g_) {
            constexpr int G = decltype(g_)::value;
            constexpr int IMG = (G >= NG / 2) ? T::BYTES : 0, ks0 = (2 * G) % KS;
            bf16x8(&dst)[4] = rbuf(std::integral_constant<int, G % RD>{});
            const uint32_t a0 = qrow[ks0] + boff, a1 = qrow[ks0 + 1] + boff;
            dst[0] = afk_lds_b128<IMG>(a0);
            dst[1] = afk_lds_b128<IMG + 32 * T::RS>(a0);
            dst[2] = afk_lds_b128<IMG>(a1);
            dst[3] = afk_lds_b128<IMG + 32 * T::RS>(a1);
        };
        // transposed fragments: groups [0, DT) = dO^T d-tile g (-> dV), [DT, 2 DT) = Q^T d-tile g - DT (-> dK)
        bf16x8 fa[4], fb[4];
        auto issue = [&](auto g_, bf16x8(&dst)[4]) {
            constexpr int g = decltype(g_)::value, dt = g % DT;
            constexpr int IMG = (g < DT) ? T::BYTES : 0;
            const uint32_t a0 = qtr[dt][0] + boff, a1 = qtr[dt][1] + boff;
            afk_static_for<4>([&](auto s_) { constexpr int s4 = decltype(s_)::value; dst[s4] = afk_lds_tr_frag<IMG + s4 * 16 * T::RS>(a0, a1); });
        };
        const f32x2 c2v = {c2, c2};
        bf16x8 pb[4], dsb[4];
        float pr[4][8];
        // first query this lane keeps: a key sees queries >= key (causal) and < S; a padded key sees none
        const int q_lo = key_dead ? p.S : (p.causal ? key : 0);
        auto p_chunk = [&](auto c_) {
            constexpr int c = decltype(c_)::value, qt2 = c >> 1, rb = 8 * (c & 1);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 s2 = {st[qt2][rb + e], st[qt2][rb + e + 1]};
                const f32x2 t2 = s2 * c2v;
                pr[c][e] = __builtin_amdgcn_exp2f(t2[0]);
                pr[c][e + 1] = __builtin_amdgcn_exp2f(t2[1]);
            }
            if (masked) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int qq = qt * 64 + qt2 * 32 + 8 * ((rb + e) >> 2) + 4 * hi + ((rb + e) & 3);
                    if (qq < q_lo || qq >= p.S) pr[c][e] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[c][e] = (bf16)pr[c][e];
        };
        auto ds_chunk = [&](auto c_) {
            constexpr int c = decltype(c_)::value, qt2 = c >> 1, rb = 8 * (c & 1);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 p2 = {pr[c][e], pr[c][e + 1]};
                const f32x2 d2 = {dp[qt2][rb + e], dp[qt2][rb + e + 1]};
                const f32x2 ds2 = p2 * d2;
                dsb[c][e] = (bf16)ds2[0];
                dsb[c][e + 1] = (bf16)ds2[1];
            }
        };

        afk_static_for<RD - 1>([&](auto g_) { issue_rows(g_); });
        afk_static_for<NG>([&](auto g_) {
            constexpr int G = decltype(g_)::value;
            bf16x8(&cur)[4] = rbuf(std::integral_constant<int, G % RD>{});
            if constexpr (G + RD - 1 < NG) {
                issue_rows(std::integral_constant<int, G + RD - 1>{});
                afk_lgkmcnt<4 * (RD - 1)>();
            } else if constexpr (G + 1 < NG) {
                afk_lgkmcnt<4 * (NG - 1 - G)>();
            } else {
                afk_lgkmcnt<8>();  // behind this group: the 8 tr-reads of dO^T group 0
            }
            afk_lds_tie(cur[0], cur[1], cur[2], cur[3]);
            constexpr int ks0 = (2 * G) % KS;
            if constexpr (G < NG / 2) {
                st[0] = MFMA(cur[0], kf[ks0], st[0]);
                st[1] = MFMA(cur[1], kf[ks0], st[1]);
                st[0] = MFMA(cur[2], kf[ks0 + 1], st[0]);
                st[1] = MFMA(cur[3], kf[ks0 + 1], st[1]);
            } else {
                dp[0] = MFMA(cur[0], vf[ks0], dp[0]);
                dp[1] = MFMA(cur[1], vf[ks0], dp[1]);
                dp[0] = MFMA(cur[2], vf[ks0 + 1], dp[0]);
                dp[1] = MFMA(cur[3], vf[ks0 + 1], dp[1]);
            }
            if constexpr (G + 2 == NG) issue(std::integral_constant<int, 0>{}, fa);
            if constexpr (G >= NG / 2)
                afk_static_for<PCH>([&](auto i_) { p_chunk(std::integral_constant<int, (G - NG / 2) * PCH + decltype(i_)::value>{}); });
        });
        if (AFK_DBG(p) & 8) {
            asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tsm)::"memory");
            sum_ph1 += tsm - ts0;
        }
        afk_frag_ring<2 * DT>(issue, [&](auto g_, bf16x8(&f)[4]) {
            constexpr int g = decltype(g_)::value, dt = g % DT;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (g < DT) dvacc[dt] = MFMA(f[s4], pb[s4], dvacc[dt]);
                else dkacc[dt] = MFMA(f[s4], dsb[s4], dkacc[dt]);
            }
            if constexpr (g < DT) afk_static_for<SCH>([&](auto i_) { ds_chunk(std::integral_constant<int, g * SCH + decltype(i_)::value>{}); });
        }, fa, fb);
    };

    // AFK_ATTN_DBG & 4 (tools/attn_waits.py): s_memtime stamps around the body and the barrier of every tile, summed per wave
    const int h0 = hk * group + g_begin;
    if (ntiles > 0) stage(0, h0, qt_begin);
    AFK_ATTN_BARRIER();
    if (probe) asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c_start), "=s"(rt0)::"memory");
    int t = 0;
    for (int gi = 0; gi < (block_dead ? 0 : g_count); ++gi) {
        const int h = h0 + gi;
        for (int qt = qt_begin; qt < qt_end; ++qt, ++t) {
            {
                // next tile (all scalar): the following query tile of this head, else the first tile of the next head; after the last tile a
                // redundant re-stage of it lands in the other buffer and is never read
                const bool wrap = qt + 1 == qt_end;
                const bool last = wrap && gi + 1 == g_count;
                const int nh = (wrap && !last) ? h + 1 : h, nqt = last ? qt : (wrap ? qt_begin : qt + 1);
                if (!(AFK_DBG(p) & 1) || t < 1) stage((t + 1) & 1, nh, nqt);
            }
            const bool interior = qt >= qt_int0 && qt < qt_int1;
            if (probe) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ts0)::"memory");
            // wave-uniform: causal tiles entirely before this wave's keys contribute nothing
            if (!(AFK_DBG(p) & 2) && (interior || (wave_live && !(p.causal && qt * 64 + 63 < key0)))) body(t, qt, !interior);
            if (probe) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory");
            AFK_ATTN_BARRIER();
            if (probe) {
                asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ts2)::"memory");
                sum_body += ts1 - ts0;
                sum_bar += ts2 - ts1;
            }
        }
    }
    uint64_t c_end = 0;
    if (probe) {
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c_end), "=s"(rt1)::"memory");
        if (lane == 0) {
            float* o = (float*)p.dQ + ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            o[0] = (float)(c_end - c_start); o[1] = (float)sum_body; o[2] = (float)sum_bar; o[3] = (float)ntiles;
            o[4] = (float)(rt1 - rt0); o[5] = (float)sum_ph1; o[6] = (float)blockIdx.z;
        }
    }
    if (key < p.S) {
        bf16* dKp = p.dK + b * p.dk_bs + hy * p.dk_hs + (int64_t)key * p.dk_rs;
        bf16* dVp = p.dV + b * p.dv_bs + hy * p.dv_hs + (int64_t)key * p.dv_rs;
        float kmul = p.scale;   // (softmax scale folded out of dS)
        if (p.rope_cos && !p.split_heads) {   // this block writes the final dK (partials are rotated by the reduce)
            const int64_t pr = p.rope_pos ? p.rope_pos[(int64_t)b * p.S + key] : key;
            rope_bwd_rows<DT>(dkacc, kmul, p.rope_cos + pr * D, p.rope_sin + pr * D, hi);
            kmul = 1.f;
        }
        if (p.wide) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                store_block32<true>(dKp + dt * 32, dkacc[dt], kmul, hi);
                store_block32<true>(dVp + dt * 32, dvacc[dt], 1.f, hi);
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                store_block32<false>(dKp + dt * 32, dkacc[dt], kmul, hi);
                store_block32<false>(dVp + dt * 32, dvacc[dt], 1.f, hi);
            }
        }
    }
    if (probe) {
        // block timeline on the 100 MHz clock: entry -> loop start -> loop end -> stores retired, + the CU the block ran on
        uint64_t rt_end;
        asm volatile("s_waitcnt vmcnt(0)\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
        if (threadIdx.x == 0) {
            const uint32_t hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
            const int64_t nblk = (int64_t)gridDim.x * gridDim.y * gridDim.z;
            const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            double* o = (double*)((float*)p.dQ + nblk * 4 * 8) + blk * 6;
            o[0] = (double)rt_entry; o[1] = (double)rt0; o[2] = (double)rt1; o[3] = (double)rt_end; o[4] = (double)hwid; o[5] = (double)xcc;
        }
    }
}

// delta[b,h,s] = -sum_d dO*O  written with row pitch Spad
template <int D>
__global__ __launch_bounds__(256) void attn2_delta_kernel(const bf16* __restrict__ O, int64_t o_bs, int64_t o_hs, int64_t o_rs,
                                                          const bf16* __restrict__ dO, int64_t do_bs, int64_t do_hs, int64_t do_rs,
                                                          float* __restrict__ delta, int B, int H, int S, int Spad) {
    constexpr int LPR = D / 8, IPB = 256 / LPR;
    const int64_t total = (int64_t)B * H * S;
    const int sub = threadIdx.x % LPR;
    for (int64_t base = (int64_t)blockIdx.x * IPB; base < total; base += (int64_t)gridDim.x * IPB) {
        const int64_t i = base + threadIdx.x / LPR;
        float acc = 0.f;
        const bool ok = i < total;
        int s = 0;
        int64_t t = 0;
        if (ok) {
            s = (int)(i % S);
            t = i / S;
            const int h = (int)(t % H), b = (int)(t / H);
            const bf16x8 o = *(const bf16x8*)(O + b * o_bs + h * o_hs + (int64_t)s * o_rs + sub * 8);
            const bf16x8 d = *(const bf16x8*)(dO + b * do_bs + h * do_hs + (int64_t)s * do_rs + sub * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)o[e] * (float)d[e];
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (ok && sub == 0) delta[t * Spad + s] = -acc;  // negated: the backward kernels start the dP accumulators from it
    }
}

// GQA reduce: out[row][hk][d] = sum_g part[row][hk*group+g][d]   (rows = B*S; part rows are Hq*D wide, out rows ld_out wide)
__global__ __launch_bounds__(256) void gqa_reduce_kernel(const bf16* __restrict__ part, bf16* __restrict__ out, int64_t rows, int Hkv,
                                                         int group, int D, int64_t ld_out) {
    const int vpr = Hkv * D / 8;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const int hk = (v * 8) / D, d = (v * 8) % D;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int g = 0; g < group; ++g) {
            const bf16x8 t = *(const bf16x8*)(part + r * (int64_t)(Hkv * group * D) + (hk * group + g) * D + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)t[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
        *(bf16x8*)(out + r * ld_out + hk * D + d) = o;
    }
}

// Both GQA reduces in ONE launch, the rotary backward of dK folded in (round 6).  A dK item = one (row, kv head, 8-column vector of the FIRST half head) plus its
// rotate-half partner vector d + D/2: both are summed over the P partial images in part order (fp32, as gqa_reduce_kernel), rounded to bf16 - the value the
// separate reduce stored - then rotated with rope_kernel's rounding points (sign = -1).  A dV item = one 8-column vector, summed and rounded.  Items
// [0, n_k) are dK, [n_k, n_k + n_v) dV.
__global__ __launch_bounds__(256) void gqa_reduce_rope_kernel(const bf16* __restrict__ pk, const bf16* __restrict__ pv, bf16* __restrict__ dk, bf16* __restrict__ dv,
                                                              int64_t rows, int Hkv, int P, int D, int64_t ld_k, int64_t ld_v, const bf16* __restrict__ cos_t,
                                                              const bf16* __restrict__ sin_t, const int* __restrict__ pos, int S) {
    const int half = D >> 1, vk = Hkv * half / 8, vv = Hkv * D / 8;
    const int64_t n_k = rows * vk, total = n_k + rows * vv, pw = (int64_t)Hkv * P * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < n_k) {
            const int v = (int)(i % vk);
            const int64_t r = i / vk;
            const int hk = (v * 8) / half, d = (v * 8) % half;
            float a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
            for (int g = 0; g < P; ++g) {
                const bf16* src = pk + r * pw + (hk * P + g) * D + d;
                const bf16x8 t1 = *(const bf16x8*)src, t2 = *(const bf16x8*)(src + half);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a[e] += (float)t1[e];
                    b[e] += (float)t2[e];
                }
            }
            const int64_t pr = pos ? pos[r] : r % S;
            const bf16x8 c1 = *(const bf16x8*)(cos_t + pr * D + d), s1 = *(const bf16x8*)(sin_t + pr * D + d);
            const bf16x8 c2 = *(const bf16x8*)(cos_t + pr * D + half + d), s2 = *(const bf16x8*)(sin_t + pr * D + half + d);
            bf16x8 o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = rbf_strict(a[e]), y = rbf_strict(b[e]);
                o1[e] = (bf16)(rbf_strict(x * (float)c1[e]) + rbf_strict(y * (float)s1[e]));
                o2[e] = (bf16)(rbf_strict(y * (float)c2[e]) + rbf_strict(-x * (float)s2[e]));
            }
            *(bf16x8*)(dk + r * ld_k + hk * D + d) = o1;
            *(bf16x8*)(dk + r * ld_k + hk * D + half + d) = o2;
        } else {
            const int64_t j = i - n_k;
            const int v = (int)(j % vv);
            const int64_t r = j / vv;
            const int hk = (v * 8) / D, d = (v * 8) % D;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int g = 0; g < P; ++g) {
                const bf16x8 t = *(const bf16x8*)(pv + r * pw + (hk * P + g) * D + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)t[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
            *(bf16x8*)(dv + r * ld_v + hk * D + d) = o;
        }
    }
}

// lane-major copy of a [rows, D] rotary table (AttnArgs2::rope_cos_lanes): out has ceil(rows / 32) * 32 * D elements, rows beyond `rows` read zero.
// form 1 = the qkv GEMM epilogue's layout (GemmArgs::rope_cos_lanes): element ((rb * (D / 8) + c) * 32 + l31) * 8 + e = table[32 rb + l31][8 c + e]
__global__ __launch_bounds__(256) void rope_lanes_kernel(const bf16* __restrict__ tab, bf16* __restrict__ out, int rows, int D, int form) {
    const int64_t total = (int64_t)((rows + 31) / 32) * 32 * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (form == 1) {
            const int e = (int)(i & 7), l31 = (int)((i >> 3) & 31);
            const int64_t t = i >> 8;   // rb * (D / 8) + c
            const int c = (int)(t % (D / 8));
            const int64_t r = (t / (D / 8)) * 32 + l31;
            out[i] = r < rows ? tab[r * D + 8 * c + e] : (bf16)0.f;
            continue;
        }
        const int e = (int)(i & 3), l31 = (int)((i >> 2) & 31), hi = (int)((i >> 7) & 1);
        const int PH = D / 16;      // 8-column pieces per half head
        const int64_t t = i >> 8;   // (rb * 2 + h2) * PH + j
        const int j = (int)(t % PH), h2 = (int)((t / PH) & 1);
        const int64_t rb = t / PH / 2;
        const int64_t r = rb * 32 + l31;
        out[i] = r < rows ? tab[r * D + h2 * (D / 2) + 8 * j + 4 * hi + e] : (bf16)0.f;
    }
}

// AFK_ATTN_WIDE=0: the 8-byte epilogue stores of rounds 1-3 (A/B)
bool attn_wide_stores() {
    static const bool on = [] { const char* e = getenv("AFK_ATTN_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}

// AFK_ATTN_SCHED=0: the compiler-scheduled fragment reads of rounds 1-5 (A/B); default 1 = explicit read rings + interleaved LDS-DMA (round 6)
int g_persist_paired = 0; // afk_attn_set_persist_paired: the persistent forward without its queue, two complementary items per block
int g_xcd_map = -1;      // -1: not chosen yet (AFK_ATTN_XCD, default 1)
int attn_xcd_map() {
    if (g_xcd_map < 0) {
        const char* e = getenv("AFK_ATTN_XCD");
        g_xcd_map = e ? (atoi(e) != 0) : 1;
    }
    return g_xcd_map;
}
int g_dkdv_parts = 0;    // afk_attn_set_dkdv_parts: > 0 overrides AFK_ATTN_DKDV_PARTS / the default (A/B inside one process)
int g_attn_sched = -1;   // -1: not chosen yet (environment, then the default)
int attn_sched() {
    if (g_attn_sched < 0) {
        const char* e = getenv("AFK_ATTN_SCHED");
        g_attn_sched = e ? (atoi(e) != 0) : 1;
    }
    return g_attn_sched;
}

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 1;
}

}  // namespace

// LSE / delta rows are Spad long (multiple of 64) so that the kernels can use aligned float4 reads; the tail [S, Spad) reads zero: afk_attn2_fwd
// and afk_attn2_bwd_fused write it themselves (round 4), afk_attn2_delta does not - its caller clears the delta tail.
extern "C" int afk_attn2_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, void* O, int64_t o_bs,
                             int64_t o_hs, int64_t o_rs, float* LSE, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S, int Spad,
                             int D, float scale, int causal, void* stream) {
    AFK_REQUIRE(Q && K && V && O && LSE, "afk_attn2_fwd: null pointer");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && S > 0 && Spad >= S && Spad % 64 == 0, "afk_attn2_fwd: bad shape");
    AFK_REQUIRE(D == 64 || D == 128, "afk_attn2_fwd: head_dim %d unsupported by the LDS kernels (64/128)", D);
    AFK_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 && v_hs % 8 == 0 && o_rs % 4 == 0,
                "afk_attn2_fwd: strides must keep 16-byte alignment");
    AttnArgs2 p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.V = (const bf16*)V; p.v_bs = v_bs; p.v_hs = v_hs; p.v_rs = v_rs;
    p.O = (bf16*)O; p.o_bs = o_bs; p.o_hs = o_hs; p.o_rs = o_rs;
    AFK_REQUIRE(!kv_lo || causal, "afk_attn2_fwd: kv_lo (left padding) is defined for causal attention only");
    p.LSE = LSE; p.kv_len = kv_len; p.kv_lo = kv_lo;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.scale = scale; p.causal = causal;
    p.wide = attn_wide_stores() && (uintptr_t)O % 16 == 0 && o_bs % 8 == 0 && o_hs % 8 == 0 && o_rs % 8 == 0;
    p.nz = (int)afk_cdiv(S, 128);
    p.xcd_map = attn_xcd_map();
#ifdef AFK_PROBES
    {
        static const char* dbg = getenv("AFK_ATTN_DBG");
        if (dbg) p.dbg = atoi(dbg);
    }
#endif
    dim3 grid((unsigned)Hq, (unsigned)B, (unsigned)p.nz);
    if (p.xcd_map) grid = dim3((unsigned)(Hq * B * p.nz));
    hipStream_t st = (hipStream_t)stream;
    afk_count(D == 128 ? AFK_CNT_ATTN2_FWD_D128 : AFK_CNT_ATTN2_FWD_D64);
    // row sum on the matrix pipe (kernel template LM): default at head_dim 128 - measured (profiles/r04_kernel_ab.md): -4 % on the 5-minute decoder
    // shape, neutral at S = 1024; at head_dim 64 it is neutral to +3 % (three waves per SIMD already hide the adds), so it stays off there.
    // AFK_ATTN_LSUM=0 / 1 forces it off / on for both head sizes (A/B)
    static const int lsum_env = [] { const char* e = getenv("AFK_ATTN_LSUM"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool lm = lsum_env < 0 ? D == 128 : lsum_env == 1;
#define AFK_FWD(DD, LM_, SCH_)                                                                      \
    do {                                                                                            \
        constexpr int L = 4 * Tile<DD>::BYTES;                                                      \
        static int once = set_lds(attn_fwd_lds_kernel<DD, LM_, SCH_>, L);                           \
        (void)once;                                                                                 \
        hipLaunchKernelGGL((attn_fwd_lds_kernel<DD, LM_, SCH_>), grid, dim3(256), L, st, p);        \
    } while (0)
    const int sch = attn_sched();
    if (D == 128) {
        if (sch) { if (lm) AFK_FWD(128, true, 1); else AFK_FWD(128, false, 1); }
        else { if (lm) AFK_FWD(128, true, 0); else AFK_FWD(128, false, 0); }
    } else {
        if (sch) { if (lm) AFK_FWD(64, true, 1); else AFK_FWD(64, false, 1); }
        else { if (lm) AFK_FWD(64, true, 0); else AFK_FWD(64, false, 0); }
    }
#undef AFK_FWD
    AFK_LAUNCH_CHECK("afk_attn2_fwd");
    return AFK_OK;
}

// A/B switch of the round-6 fragment schedule (0 = compiler-scheduled reads of rounds 1-5, 1 = explicit read rings + interleaved LDS-DMA; both bit-identical)
extern "C" int afk_attn_set_sched(int sched) {
    AFK_REQUIRE(sched == 0 || sched == 1, "afk_attn_set_sched: 0 or 1");
    g_attn_sched = sched;
    return AFK_OK;
}

extern "C" int afk_attn_set_persist_paired(int on) {
    AFK_REQUIRE(on == 0 || on == 1, "afk_attn_set_persist_paired: 0 or 1");
    g_persist_paired = on;
    return AFK_OK;
}

extern "C" int afk_attn_set_xcd_map(int on) {
    AFK_REQUIRE(on == 0 || on == 1, "afk_attn_set_xcd_map: 0 or 1");
    g_xcd_map = on;
    return AFK_OK;
}

extern "C" int afk_attn_set_dkdv_parts(int parts) {
    AFK_REQUIRE(parts >= 0 && parts <= 64, "afk_attn_set_dkdv_parts: 0 (environment / default) or the number of dK/dV partials per kv head");
    g_dkdv_parts = parts;
    return AFK_OK;
}

// persistent form of afk_attn2_fwd (attn_fwd_persist_kernel): `queue` = two device ints, zero before the first call, left at zero by every call.
// Falls back to the plain launch when the form's restrictions do not hold (kv_len / kv_lo given, S % 128 != 0).
extern "C" int afk_attn2_fwd_persistent(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                                        int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, void* O, int64_t o_bs,
                                        int64_t o_hs, int64_t o_rs, float* LSE, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S, int Spad,
                                        int D, float scale, int causal, int* queue, void* stream) {
    if (kv_len || kv_lo || S % 128 != 0 || !queue)
        return afk_attn2_fwd(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, V, v_bs, v_hs, v_rs, O, o_bs, o_hs, o_rs, LSE, kv_len, kv_lo, B, Hq, Hkv, S, Spad, D, scale, causal, stream);
    AFK_REQUIRE(Q && K && V && O && LSE, "afk_attn2_fwd_persistent: null pointer");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Spad >= S && Spad % 64 == 0 && (D == 64 || D == 128), "afk_attn2_fwd_persistent: bad shape");
    AFK_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 && v_hs % 8 == 0 && o_rs % 4 == 0,
                "afk_attn2_fwd_persistent: strides must keep 16-byte alignment");
    AttnArgs2 p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.V = (const bf16*)V; p.v_bs = v_bs; p.v_hs = v_hs; p.v_rs = v_rs;
    p.O = (bf16*)O; p.o_bs = o_bs; p.o_hs = o_hs; p.o_rs = o_rs;
    p.LSE = LSE;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.scale = scale; p.causal = causal;
    p.wide = attn_wide_stores() && (uintptr_t)O % 16 == 0 && o_bs % 8 == 0 && o_hs % 8 == 0 && o_rs % 8 == 0;
    const int total = Hq * B * (S / 128);
    hipStream_t st = (hipStream_t)stream;
    afk_count(D == 128 ? AFK_CNT_ATTN2_FWD_D128 : AFK_CNT_ATTN2_FWD_D64);
    static const int lsum_env = [] { const char* e = getenv("AFK_ATTN_LSUM"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool lm = lsum_env < 0 ? D == 128 : lsum_env == 1;
    static const int slots_env = [] { const char* e = getenv("AFK_ATTN_PERSIST_BLOCKS"); return e ? atoi(e) : 0; }();
    p.paired = g_persist_paired;
#define AFK_FWDP(DD, LM_)                                                                                                \
    do {                                                                                                                 \
        constexpr int L = 4 * Tile<DD>::BYTES;                                                                           \
        static int once = set_lds(attn_fwd_persist_kernel<DD, LM_>, L);                                                  \
        (void)once;                                                                                                      \
        const int slots = slots_env > 0 ? slots_env : 512;   /* two resident blocks per CU (237 / 175-189 VGPRs: two waves per SIMD) */          \
        const int nblk = p.paired ? (total + 1) / 2 : std::min(total, slots);                                            \
        hipLaunchKernelGGL((attn_fwd_persist_kernel<DD, LM_>), dim3((unsigned)nblk), dim3(256), L, st, p, queue, total); \
    } while (0)
    if (D == 128) {
        if (lm) AFK_FWDP(128, true); else AFK_FWDP(128, false);
    } else {
        if (lm) AFK_FWDP(64, true); else AFK_FWDP(64, false);
    }
#undef AFK_FWDP
    AFK_LAUNCH_CHECK("afk_attn2_fwd_persistent");
    return AFK_OK;
}

extern "C" int afk_attn2_delta(const void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs,
                               int64_t do_rs, float* delta, int B, int H, int S, int Spad, int D, void* stream) {
    AFK_REQUIRE(O && dO && delta && (D == 64 || D == 128) && Spad >= S, "afk_attn2_delta: bad args");
    const int64_t total = (int64_t)B * H * S;
    int grid = (int)afk_cdiv(total, 256 / (D / 8));
    if (grid > 8192) grid = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (D == 128)
        hipLaunchKernelGGL(attn2_delta_kernel<128>, dim3(grid), dim3(256), 0, st, (const bf16*)O, o_bs, o_hs, o_rs, (const bf16*)dO, do_bs, do_hs, do_rs, delta, B, H, S, Spad);
    else
        hipLaunchKernelGGL(attn2_delta_kernel<64>, dim3(grid), dim3(256), 0, st, (const bf16*)O, o_bs, o_hs, o_rs, (const bf16*)dO, do_bs, do_hs, do_rs, delta, B, H, S, Spad);
    AFK_LAUNCH_CHECK("afk_attn2_delta");
    return AFK_OK;
}

static int attn2_bwd_impl(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                          int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* O, int64_t o_bs, int64_t o_hs,
                          int64_t o_rs, const void* dO, int64_t do_bs,
                          int64_t do_hs, int64_t do_rs, const float* LSE, const float* delta, void* dQ, int64_t dq_bs,
                          int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV,
                          int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S,
                          int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream, const void* rope_cos = nullptr,
                          const void* rope_sin = nullptr, const int* rope_pos = nullptr, const void* rope_cos_lanes = nullptr, const void* rope_sin_lanes = nullptr) {
    AFK_REQUIRE(Q && K && V && dO && LSE && delta && dQ && dK && dV, "afk_attn2_bwd: null pointer");
    AFK_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), "afk_attn2_bwd_fused_rope: cos and sin tables come together");
    AFK_REQUIRE(!rope_cos || ((uintptr_t)rope_cos % 16 == 0 && (uintptr_t)rope_sin % 16 == 0), "afk_attn2_bwd_fused_rope: the tables must be 16-byte aligned");
    AFK_REQUIRE(!O || (o_rs % 8 == 0 && o_hs % 8 == 0), "afk_attn2_bwd_fused: O strides must keep 16-byte alignment");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && S > 0 && Spad >= S && Spad % 64 == 0, "afk_attn2_bwd: bad shape");
    AFK_REQUIRE(D == 64 || D == 128, "afk_attn2_bwd: head_dim %d unsupported by the LDS kernels (64/128)", D);
    AFK_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && do_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 && v_hs % 8 == 0 &&
                    do_hs % 8 == 0,
                "afk_attn2_bwd: strides must keep 16-byte alignment");
    AttnArgs2 p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.V = (const bf16*)V; p.v_bs = v_bs; p.v_hs = v_hs; p.v_rs = v_rs;
    p.dO = (const bf16*)dO; p.do_bs = do_bs; p.do_hs = do_hs; p.do_rs = do_rs;
    p.O = (bf16*)O; p.o_bs = o_bs; p.o_hs = o_hs; p.o_rs = o_rs;   // non-null: the dQ kernel computes delta itself and runs FIRST
    p.dQ = (bf16*)dQ; p.dq_bs = dq_bs; p.dq_hs = dq_hs; p.dq_rs = dq_rs;
    p.dK = (bf16*)dK; p.dk_bs = dk_bs; p.dk_hs = dk_hs; p.dk_rs = dk_rs;
    p.dV = (bf16*)dV; p.dv_bs = dv_bs; p.dv_hs = dv_hs; p.dv_rs = dv_rs;
    AFK_REQUIRE(!kv_lo || causal, "afk_attn2_bwd: kv_lo (left padding) is defined for causal attention only");
    p.LSE = (float*)LSE; p.delta = delta; p.kv_len = kv_len; p.kv_lo = kv_lo;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.scale = scale; p.causal = causal;
    p.rope_cos = (const bf16*)rope_cos; p.rope_sin = (const bf16*)rope_sin; p.rope_pos = rope_pos;
    AFK_REQUIRE((rope_cos_lanes == nullptr) == (rope_sin_lanes == nullptr) && (!rope_cos_lanes || (rope_cos && (uintptr_t)rope_cos_lanes % 8 == 0 && (uintptr_t)rope_sin_lanes % 8 == 0)),
                "afk_attn2_bwd_fused_rope: the lane-major tables come as a pair, beside the plain ones");
    p.rope_cos_lanes = (const bf16*)rope_cos_lanes; p.rope_sin_lanes = (const bf16*)rope_sin_lanes;
    hipStream_t st = (hipStream_t)stream;
    // GQA: with few kv heads the dK/dV sweep has too few blocks to fill 256 CUs (decoder: 8x4x8 = 256 long blocks).
    // Given a scratch of 2 * B*S*Hq*D bf16 the sweep runs one block per QUERY head and a reduce folds the group.
    const int group = Hq / Hkv;
    const bool split = gqa_scratch != nullptr && group > 1;
#ifdef AFK_PROBES
    {
        static const char* dbg = getenv("AFK_ATTN_DBG");
        if (dbg) p.dbg = atoi(dbg);
    }
#endif
    auto al16 = [](const void* ptr, int64_t bs, int64_t hs, int64_t rs) { return (uintptr_t)ptr % 16 == 0 && bs % 8 == 0 && hs % 8 == 0 && rs % 8 == 0; };
    p.wide = attn_wide_stores() && al16(dQ, dq_bs, dq_hs, dq_rs) && al16(dK, dk_bs, dk_hs, dk_rs) && al16(dV, dv_bs, dv_hs, dv_rs) &&
             (!split || ((uintptr_t)gqa_scratch % 16 == 0 && D % 8 == 0));
    AttnArgs2 pk = p;
    // parts per kv head: AFK_ATTN_DKDV_PARTS (1 .. group; default below).  The scratch is sized for `group` partials, any P <= group fits.
    static const int parts_env = [] { const char* e = getenv("AFK_ATTN_DKDV_PARTS"); return e ? atoi(e) : 0; }();
    // default (round 6, profiles/r06_attn_ab.md): the SMALLEST part count that still gives the chip two blocks per CU - fewer, longer blocks amortise a block's
    // prologue + store tail (~8 us against 2.5 us per tile) and fewer partials cross HBM: AF3 decoder (B = 8, S = 1024) P = 2: 345 -> 307 us for the whole
    // backward, S = 2048 (B = 4) 535 -> 509; B = 1 at S = 7774 P = 3 (level with 7; P = 2 starves the grid: +3 %)
    int P = 0;
    if (split) {
        const int64_t per_part = (int64_t)B * Hkv * afk_cdiv(S, 128);
        int want = g_dkdv_parts > 0 ? g_dkdv_parts : parts_env;
        if (want <= 0) want = (int)std::min<int64_t>(group, std::max<int64_t>(1, afk_cdiv(512, per_part)));
        P = std::max(1, std::min(group, want));
    }
    const bool direct = split && P == 1;   // one block per kv head sweeps the whole group: no partials, no reduce
    if (split && !direct) {
        pk.split_heads = P;
        pk.dK = (bf16*)gqa_scratch;
        pk.dV = (bf16*)gqa_scratch + (int64_t)B * S * Hq * D;
        pk.dk_bs = pk.dv_bs = (int64_t)S * Hkv * P * D;
        pk.dk_hs = pk.dv_hs = D;
        pk.dk_rs = pk.dv_rs = (int64_t)Hkv * P * D;
    }
    afk_count(D == 128 ? AFK_CNT_ATTN2_BWD_D128 : AFK_CNT_ATTN2_BWD_D64);
    if (split && !direct) afk_count(AFK_CNT_GQA_REDUCE);
    dim3 gkv((unsigned)(split && !direct ? Hkv * P : Hkv), (unsigned)B, (unsigned)afk_cdiv(S, 128));
    p.nz = pk.nz = (int)afk_cdiv(S, 128);
    p.xcd_map = attn_xcd_map();
    dim3 gq((unsigned)Hq, (unsigned)B, (unsigned)p.nz);
    if (p.xcd_map) gq = dim3((unsigned)(Hq * B * p.nz));
    const int sch = attn_sched();
    auto launch_dq = [&](auto d_) {
        constexpr int DD = decltype(d_)::value;
        constexpr int L = 4 * Tile<DD>::BYTES;
        if (sch) hipLaunchKernelGGL((attn_bwd_dq_lds_kernel<DD, 1>), gq, dim3(256), L, st, p);
        else hipLaunchKernelGGL((attn_bwd_dq_lds_kernel<DD, 0>), gq, dim3(256), L, st, p);
    };
    if (D == 128) {
        constexpr int L = 4 * Tile<128>::BYTES, LKV = L + 2048;  // dK/dV sweep: + one lse/delta strip per buffer
        static int once = set_lds(attn_bwd_dkdv_lds_kernel<128>, LKV) + set_lds(attn_bwd_dq_lds_kernel<128, 0>, L) + set_lds(attn_bwd_dq_lds_kernel<128, 1>, L);
        (void)once;
        if (O) launch_dq(std::integral_constant<int, 128>{});   // publishes delta for the sweep behind it
        hipLaunchKernelGGL(attn_bwd_dkdv_lds_kernel<128>, gkv, dim3(256), LKV, st, pk);
        if (!O && !(AFK_DBG(p) & 4)) launch_dq(std::integral_constant<int, 128>{});
    } else {
        constexpr int L = 4 * Tile<64>::BYTES, LKV = L + 2048;
        static int once = set_lds(attn_bwd_dkdv_lds_kernel<64>, LKV) + set_lds(attn_bwd_dq_lds_kernel<64, 0>, L) + set_lds(attn_bwd_dq_lds_kernel<64, 1>, L);
        (void)once;
        if (O) launch_dq(std::integral_constant<int, 64>{});
        hipLaunchKernelGGL(attn_bwd_dkdv_lds_kernel<64>, gkv, dim3(256), LKV, st, pk);
        if (!O && !(AFK_DBG(p) & 4)) launch_dq(std::integral_constant<int, 64>{});
    }
    if (split && !direct && !(AFK_DBG(p) & 4)) {
        AFK_REQUIRE(dk_hs == D && dv_hs == D && dk_bs == (int64_t)S * dk_rs && dv_bs == (int64_t)S * dv_rs && dk_rs == dv_rs,
                    "afk_attn2_bwd: GQA split path expects dK/dV inside one [B*S, ld] buffer with contiguous heads");
        const int64_t rows = (int64_t)B * S;
        int g = (int)afk_cdiv(rows * (Hkv * D / 8), 256);
        if (g > 4096) g = 4096;
        if (rope_cos) {
            const int64_t items = rows * (Hkv * D / 16) + rows * (Hkv * D / 8);
            int g2 = (int)std::min<int64_t>(afk_cdiv(items, 256), 4096);
            hipLaunchKernelGGL(gqa_reduce_rope_kernel, dim3(g2), dim3(256), 0, st, pk.dK, pk.dV, (bf16*)dK, (bf16*)dV, rows, Hkv, P, D, dk_rs, dv_rs,
                               (const bf16*)rope_cos, (const bf16*)rope_sin, rope_pos, S);
        } else {
            hipLaunchKernelGGL(gqa_reduce_kernel, dim3(g), dim3(256), 0, st, pk.dK, (bf16*)dK, rows, Hkv, P, D, dk_rs);
            hipLaunchKernelGGL(gqa_reduce_kernel, dim3(g), dim3(256), 0, st, pk.dV, (bf16*)dV, rows, Hkv, P, D, dv_rs);
        }
    }
    AFK_LAUNCH_CHECK("afk_attn2_bwd");
    return AFK_OK;
}

extern "C" int afk_attn2_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO, int64_t do_bs,
                             int64_t do_hs, int64_t do_rs, const float* LSE, const float* delta, void* dQ, int64_t dq_bs,
                             int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV,
                             int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq, int Hkv, int S,
                             int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream) {
    return attn2_bwd_impl(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, V, v_bs, v_hs, v_rs, nullptr, 0, 0, 0, dO, do_bs, do_hs, do_rs, LSE, delta, dQ, dq_bs,
                          dq_hs, dq_rs, dK, dk_bs, dk_hs, dk_rs, dV, dv_bs, dv_hs, dv_rs, kv_len, kv_lo, B, Hq, Hkv, S, Spad, D, scale, causal, gqa_scratch, stream);
}

// the same with delta = rowsum(dO o O) computed inside the dQ kernel (no afk_attn2_delta call): delta_ws [B, Hq, Spad] fp32 is a WORKSPACE written
// by this call (its padding tail [S, Spad) must read zero)
extern "C" int afk_attn2_bwd_fused(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                                   int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* O, int64_t o_bs, int64_t o_hs,
                                   int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs, int64_t do_rs, const float* LSE, float* delta_ws,
                                   void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs,
                                   void* dV, int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq,
                                   int Hkv, int S, int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream) {
    AFK_REQUIRE(O != nullptr, "afk_attn2_bwd_fused: O is required (use afk_attn2_bwd with a precomputed delta otherwise)");
    return attn2_bwd_impl(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, V, v_bs, v_hs, v_rs, O, o_bs, o_hs, o_rs, dO, do_bs, do_hs, do_rs, LSE, delta_ws, dQ, dq_bs,
                          dq_hs, dq_rs, dK, dk_bs, dk_hs, dk_rs, dV, dv_bs, dv_hs, dv_rs, kv_len, kv_lo, B, Hq, Hkv, S, Spad, D, scale, causal, gqa_scratch, stream);
}

// afk_attn2_bwd_fused + the backward of the rotary embedding on dQ and dK (the reference rotates q and k before the attention, modeling_qwen2.py:213; its
// autograd multiplies the incoming gradients by the transposed rotation): cos_t / sin_t [positions, D] bf16 (16-byte aligned), pos [B * S] int32 or null
// (= row % S).  dQ and dK come out ROTATED - exactly the bits afk_attn2_bwd_fused followed by afk_rope_inplace(backward = 1) on the q and k columns gives;
// dV is untouched.  Saves the separate pass over dq | dk (29 us per decoder layer at the AF3 shape) and one of the two GQA reduce launches.
extern "C" int afk_attn2_bwd_fused_rope(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                                        int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* O, int64_t o_bs, int64_t o_hs,
                                        int64_t o_rs, const void* dO, int64_t do_bs, int64_t do_hs, int64_t do_rs, const float* LSE, float* delta_ws,
                                        void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs,
                                        void* dV, int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, const int* kv_lo, int B, int Hq,
                                        int Hkv, int S, int Spad, int D, float scale, int causal, void* gqa_scratch, const void* cos_t, const void* sin_t,
                                        const int* pos, const void* cos_lanes, const void* sin_lanes, void* stream) {
    AFK_REQUIRE(O != nullptr && cos_t && sin_t, "afk_attn2_bwd_fused_rope: O and the cos / sin tables are required");
    return attn2_bwd_impl(Q, q_bs, q_hs, q_rs, K, k_bs, k_hs, k_rs, V, v_bs, v_hs, v_rs, O, o_bs, o_hs, o_rs, dO, do_bs, do_hs, do_rs, LSE, delta_ws, dQ, dq_bs,
                          dq_hs, dq_rs, dK, dk_bs, dk_hs, dk_rs, dV, dv_bs, dv_hs, dv_rs, kv_len, kv_lo, B, Hq, Hkv, S, Spad, D, scale, causal, gqa_scratch, stream,
                          cos_t, sin_t, pos, cos_lanes, sin_lanes);
}

// lane-major copy of a rotary table for afk_attn2_bwd_fused_rope (cos_lanes / sin_lanes): table [rows, D] bf16 -> out [ceil(rows / 32) * 32 * D] bf16.
// Built once per table (the host caches it beside the table); rows = at least the S of the calls that use it.
extern "C" int afk_rope_lanes_table(const void* table, void* out, int rows, int D, int form, void* stream) {
    AFK_REQUIRE(table && out && rows > 0 && (D == 64 || D == 128) && (form == 0 || form == 1), "afk_rope_lanes_table: bad args (D = 64 / 128, form 0 / 1)");
    const int64_t total = (int64_t)((rows + 31) / 32) * 32 * D;
    hipLaunchKernelGGL(rope_lanes_kernel, dim3((unsigned)std::min<int64_t>(afk_cdiv(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream, (const bf16*)table, (bf16*)out,
                       rows, D, form);
    AFK_LAUNCH_CHECK("afk_rope_lanes_table");
    return AFK_OK;
}
