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
    int B, Hq, Hkv, S, Spad;
    float scale;
    int causal;
    int split_heads;  // dK/dV sweep: one block per QUERY head, partial dK/dV per query head (GQA), reduced afterwards
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

// ------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][K image | V image]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (p.Hq / p.Hkv);
    // causal: late query blocks sweep the most keys - dispatch them first so the grid drains evenly
    const int qb0 = (p.causal ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 128;
    const int q0 = qb0 + wave * 32;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const bool wave_live = q0 < p.S;

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qp + ks * 16);

    const bf16* Kbase = p.K + b * p.k_bs + hk * p.k_hs;
    const bf16* Vbase = p.V + b * p.v_bs + hk * p.v_hs;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
    float m = NEG_INF, l = 0.f;  // running max in the log2 domain
    const float c2 = p.scale * LOG2E;

    const int kv_end_blk = p.causal ? min(kv_len, min(qb0 + 128, p.S)) : kv_len;
    const int ntiles = (kv_end_blk + 63) >> 6;
    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;  // this wave's horizon

    auto stage = [&](int j) {
        char* buf = smem + (j & 1) * 2 * T::BYTES;
        T::stage(buf, Kbase, p.k_rs, j * 64, p.S - 1, wave, lane);
        T::stage(buf + T::BYTES, Vbase, p.v_rs, j * 64, p.S - 1, wave, lane);
    };
    if (ntiles > 0) stage(0);
    __syncthreads();
    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) stage(j + 1);
        const char* kimg = smem + (j & 1) * 2 * T::BYTES;
        const char* vimg = kimg + T::BYTES;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            const int key0 = j * 64 + kt2 * 32;
            if (wave_live && key0 < kv_end) {  // wave-uniform
                f32x16 st = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) st = MFMA(T::row_frag(kimg, offs, kt2, ks), qf[ks], st);
                // softmax in the log2 domain: t = s*scale*log2(e); p = 2^(t - m).  The kernels are VALU-bound, so the mask
                // arithmetic runs only on boundary tiles (wave-uniform test), the O rescale only when some row max moved.
                const bool need_mask = (key0 + 32 > kv_len) || (p.causal && key0 + 31 > q0);
                float t[16];
                float mx = NEG_INF;
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + ROW_OF(r, hi);
                        const bool dead = (key >= kv_len) || (p.causal && key > q);
                        t[r] = dead ? NEG_INF : st[r] * c2;
                        mx = fmaxf(mx, t[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        t[r] = st[r] * c2;
                        mx = fmaxf(mx, t[r]);
                    }
                }
                mx = fmaxf(mx, other_half(mx));
                const float m_new = fmaxf(m, mx);
                const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float rs = 0.f;
                bf16x8 pb[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(t[r] - m_use);
                    rs += pv;
                    pb[r >> 3][r & 7] = (bf16)pv;
                }
                rs += other_half(rs);
                if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {  // some row's running max moved: rescale (rare after the first tiles)
                    const float alpha = __builtin_amdgcn_exp2f(m - m_use);
                    l *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                }
                l += rs;
                m = m_new;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) oacc[dt] = MFMA(T::tr_frag(vimg, offs, dt, 2 * kt2 + s2), pb[s2], oacc[dt]);
            }
        }
        __syncthreads();
    }
    if (q < p.S) {
        const float inv = (l > 0.f) ? 1.f / l : 0.f;
        bf16* Op = p.O + b * p.o_bs + h * p.o_hs + (int64_t)q * p.o_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(oacc[dt][4 * qd + e] * inv);
                *(bf16x4*)(Op + dt * 32 + 8 * qd + 4 * hi) = o;
            }
        if (hi == 0 && p.LSE) p.LSE[((int64_t)b * p.Hq + h) * p.Spad + q] = (l > 0.f) ? m + __log2f(l) : NEG_INF;  // log2 domain (internal to the v2 kernels)
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (p.Hq / p.Hkv);
    // causal: late query blocks sweep the most keys - dispatch them first so the grid drains evenly
    const int qb0 = (p.causal ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 128;
    const int q0 = qb0 + wave * 32;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const bool wave_live = q0 < p.S;

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    const bf16* dOp = p.dO + b * p.do_bs + h * p.do_hs + (int64_t)qc * p.do_rs + hi * 8;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const bf16x8*)(Qp + ks * 16);
        dof[ks] = *(const bf16x8*)(dOp + ks * 16);
    }
    const float c2 = p.scale * LOG2E;
    const float lse2 = p.LSE[((int64_t)b * p.Hq + h) * p.Spad + qc];  // already in the log2 domain
    const float dlt = p.delta[((int64_t)b * p.Hq + h) * p.Spad + qc];
    const bf16* Kbase = p.K + b * p.k_bs + hk * p.k_hs;
    const bf16* Vbase = p.V + b * p.v_bs + hk * p.v_hs;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = zero16();

    const int kv_end_blk = p.causal ? min(kv_len, min(qb0 + 128, p.S)) : kv_len;
    const int ntiles = (kv_end_blk + 63) >> 6;
    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;

    auto stage = [&](int j) {
        char* buf = smem + (j & 1) * 2 * T::BYTES;
        T::stage(buf, Kbase, p.k_rs, j * 64, p.S - 1, wave, lane);
        T::stage(buf + T::BYTES, Vbase, p.v_rs, j * 64, p.S - 1, wave, lane);
    };
    if (ntiles > 0) stage(0);
    __syncthreads();
    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) stage(j + 1);
        const char* kimg = smem + (j & 1) * 2 * T::BYTES;
        const char* vimg = kimg + T::BYTES;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            const int key0 = j * 64 + kt2 * 32;
            if (wave_live && key0 < kv_end) {
                f32x16 st = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    st = MFMA(T::row_frag(kimg, offs, kt2, ks), qf[ks], st);
                    dp = MFMA(T::row_frag(vimg, offs, kt2, ks), dof[ks], dp);
                }
                bf16x8 dsb[2];
                const bool need_mask = (key0 + 32 > kv_len) || (p.causal && key0 + 31 > q0);
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + ROW_OF(r, hi);
                        const bool dead = (key >= kv_len) || (p.causal && key > q);
                        const float pv = __builtin_amdgcn_exp2f(st[r] * c2 - lse2);
                        dsb[r >> 3][r & 7] = (bf16)(dead ? 0.f : pv * (dp[r] - dlt));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(st[r] * c2 - lse2);
                        dsb[r >> 3][r & 7] = (bf16)(pv * (dp[r] - dlt));
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) dqacc[dt] = MFMA(T::tr_frag(kimg, offs, dt, 2 * kt2 + s2), dsb[s2], dqacc[dt]);
            }
        }
        __syncthreads();
    }
    if (q < p.S) {
        bf16* dQp = p.dQ + b * p.dq_bs + h * p.dq_hs + (int64_t)q * p.dq_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(dqacc[dt][4 * qd + e] * p.scale);  // softmax scale folded out of dS
                *(bf16x4*)(dQp + dt * 32 + 8 * qd + 4 * hi) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// block = 128 keys of one kv head (wave = 32 keys); streams 64-query tiles of Q and dO of every head of the GQA group.
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dkdv_lds_kernel(AttnArgs2 p) {
    using T = Tile<D>;
    constexpr int KS = T::KS, DT = T::DT;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][Q image | dO image]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const typename T::Offs offs = T::make_offs(lane);
    const int b = blockIdx.z, group = p.Hq / p.Hkv;
    const int hy = blockIdx.y;
    const int hk = p.split_heads ? hy / group : hy;
    const int g_begin = p.split_heads ? hy % group : 0;
    const int g_count = p.split_heads ? 1 : group;
    const int kb0 = blockIdx.x * 128;
    const int key0 = kb0 + wave * 32;
    const int key = key0 + l31;
    const int keyc = min(key, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.S) : p.S;
    const bool wave_live = key0 < p.S;
    const bool key_dead = key >= kv_len;
    const float c2 = p.scale * LOG2E;

    const bf16* Kp = p.K + b * p.k_bs + hk * p.k_hs + (int64_t)keyc * p.k_rs + hi * 8;
    const bf16* Vp = p.V + b * p.v_bs + hk * p.v_hs + (int64_t)keyc * p.v_rs + hi * 8;
    // head_dim 64: this wave's 32 keys stay in registers for the whole sweep; head_dim 128 would spill, it re-reads
    // its 8 KiB of K/V rows from L1/L2 per tile instead
    constexpr bool HOIST = true;  // head_dim 128 runs one wave per SIMD (launch bound) so that 64 operand VGPRs fit beside 128 accumulators
    bf16x8 kf[HOIST ? KS : 1], vf[HOIST ? KS : 1];
    if (HOIST) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *(const bf16x8*)(Kp + ks * 16);
            vf[ks] = *(const bf16x8*)(Vp + ks * 16);
        }
    }

    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dkacc[dt] = zero16();
        dvacc[dt] = zero16();
    }
    const int qt_begin = p.causal ? (kb0 >> 6) : 0;  // block-uniform first 64-query tile
    const int qt_end = (p.S + 63) >> 6;
    const int per_head = qt_end - qt_begin;
    const int ntiles = per_head * g_count;

    auto stage = [&](int t) {
        const int gi = t / per_head, qt = qt_begin + (t - gi * per_head);
        const int h = hk * group + g_begin + gi;
        char* buf = smem + (t & 1) * 2 * T::BYTES;
        T::stage(buf, p.Q + b * p.q_bs + h * p.q_hs, p.q_rs, qt * 64, p.S - 1, wave, lane);
        T::stage(buf + T::BYTES, p.dO + b * p.do_bs + h * p.do_hs, p.do_rs, qt * 64, p.S - 1, wave, lane);
    };
    if (ntiles > 0) stage(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage(t + 1);
        const int gi = t / per_head, qt = qt_begin + (t - gi * per_head);
        const int h = hk * group + g_begin + gi;
        const char* qimg = smem + (t & 1) * 2 * T::BYTES;
        const char* doimg = qimg + T::BYTES;
        const float* lse = p.LSE + ((int64_t)b * p.Hq + h) * p.Spad;
        const float* dlt = p.delta + ((int64_t)b * p.Hq + h) * p.Spad;
#pragma unroll
        for (int qt2 = 0; qt2 < 2; ++qt2) {
            const int qt0 = qt * 64 + qt2 * 32;
            // causal: a 32-query half entirely before this wave's keys contributes nothing
            if (wave_live && qt0 < p.S && !(p.causal && qt0 + 31 < key0)) {
                // softmax statistics first: their global-load latency hides under the QK^T / dO.V^T MFMAs
                f32x4 l4v[4], d4v[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    l4v[qd] = *(const f32x4*)(lse + qt0 + 8 * qd + 4 * hi);  // 4 consecutive queries, inside [0, Spad)
                    d4v[qd] = *(const f32x4*)(dlt + qt0 + 8 * qd + 4 * hi);
                }
                f32x16 st = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    st = MFMA(T::row_frag(qimg, offs, qt2, ks), HOIST ? kf[HOIST ? ks : 0] : *(const bf16x8*)(Kp + ks * 16), st);
                    dp = MFMA(T::row_frag(doimg, offs, qt2, ks), HOIST ? vf[HOIST ? ks : 0] : *(const bf16x8*)(Vp + ks * 16), dp);
                }
                bf16x8 pb[2], dsb[2];
                const bool need_mask = (key0 + 32 > kv_len) || (qt0 + 32 > p.S) || (p.causal && key0 + 31 > qt0);
                if (need_mask) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int qq0 = qt0 + 8 * qd + 4 * hi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * qd + e;
                            const int qq = qq0 + e;
                            const bool dead = key_dead || (qq >= p.S) || (p.causal && key > qq);
                            const float pv = dead ? 0.f : __builtin_amdgcn_exp2f(st[r] * c2 - l4v[qd][e]);
                            pb[r >> 3][r & 7] = (bf16)pv;
                            dsb[r >> 3][r & 7] = (bf16)(dead ? 0.f : pv * (dp[r] - d4v[qd][e]));
                        }
                    }
                } else {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * qd + e;
                            const float pv = __builtin_amdgcn_exp2f(st[r] * c2 - l4v[qd][e]);
                            pb[r >> 3][r & 7] = (bf16)pv;
                            dsb[r >> 3][r & 7] = (bf16)(pv * (dp[r] - d4v[qd][e]));
                        }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        dvacc[dt] = MFMA(T::tr_frag(doimg, offs, dt, 2 * qt2 + s2), pb[s2], dvacc[dt]);
                        dkacc[dt] = MFMA(T::tr_frag(qimg, offs, dt, 2 * qt2 + s2), dsb[s2], dkacc[dt]);
                    }
            }
        }
        __syncthreads();
    }
    if (key < p.S) {
        bf16* dKp = p.dK + b * p.dk_bs + hy * p.dk_hs + (int64_t)key * p.dk_rs;
        bf16* dVp = p.dV + b * p.dv_bs + hy * p.dv_hs + (int64_t)key * p.dv_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 ok, ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ok[e] = (bf16)(dkacc[dt][4 * qd + e] * p.scale);  // softmax scale folded out of dS
                    ov[e] = (bf16)dvacc[dt][4 * qd + e];
                }
                *(bf16x4*)(dKp + dt * 32 + 8 * qd + 4 * hi) = ok;
                *(bf16x4*)(dVp + dt * 32 + 8 * qd + 4 * hi) = ov;
            }
    }
}

// delta[b,h,s] = sum_d dO*O  written with row pitch Spad
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
        if (ok && sub == 0) delta[t * Spad + s] = acc;
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

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 1;
}

}  // namespace

// LSE / delta rows are Spad long (multiple of 64, zero-initialised by the host) so that the kernels can use aligned float4 reads.
extern "C" int afk_attn2_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, void* O, int64_t o_bs,
                             int64_t o_hs, int64_t o_rs, float* LSE, const int* kv_len, int B, int Hq, int Hkv, int S, int Spad,
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
    p.LSE = LSE; p.kv_len = kv_len;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.scale = scale; p.causal = causal;
    dim3 grid((unsigned)afk_cdiv(S, 128), (unsigned)Hq, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (D == 128) {
        constexpr int L = 4 * Tile<128>::BYTES;
        static int once = set_lds(attn_fwd_lds_kernel<128>, L);
        (void)once;
        hipLaunchKernelGGL(attn_fwd_lds_kernel<128>, grid, dim3(256), L, st, p);
    } else {
        constexpr int L = 4 * Tile<64>::BYTES;
        static int once = set_lds(attn_fwd_lds_kernel<64>, L);
        (void)once;
        hipLaunchKernelGGL(attn_fwd_lds_kernel<64>, grid, dim3(256), L, st, p);
    }
    AFK_LAUNCH_CHECK("afk_attn2_fwd");
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

extern "C" int afk_attn2_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO, int64_t do_bs,
                             int64_t do_hs, int64_t do_rs, const float* LSE, const float* delta, void* dQ, int64_t dq_bs,
                             int64_t dq_hs, int64_t dq_rs, void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV,
                             int64_t dv_bs, int64_t dv_hs, int64_t dv_rs, const int* kv_len, int B, int Hq, int Hkv, int S,
                             int Spad, int D, float scale, int causal, void* gqa_scratch, void* stream) {
    AFK_REQUIRE(Q && K && V && dO && LSE && delta && dQ && dK && dV, "afk_attn2_bwd: null pointer");
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
    p.dQ = (bf16*)dQ; p.dq_bs = dq_bs; p.dq_hs = dq_hs; p.dq_rs = dq_rs;
    p.dK = (bf16*)dK; p.dk_bs = dk_bs; p.dk_hs = dk_hs; p.dk_rs = dk_rs;
    p.dV = (bf16*)dV; p.dv_bs = dv_bs; p.dv_hs = dv_hs; p.dv_rs = dv_rs;
    p.LSE = (float*)LSE; p.delta = delta; p.kv_len = kv_len;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.scale = scale; p.causal = causal;
    hipStream_t st = (hipStream_t)stream;
    // GQA: with few kv heads the dK/dV sweep has too few blocks to fill 256 CUs (decoder: 8x4x8 = 256 long blocks).
    // Given a scratch of 2 * B*S*Hq*D bf16 the sweep runs one block per QUERY head and a reduce folds the group.
    const int group = Hq / Hkv;
    const bool split = gqa_scratch != nullptr && group > 1;
    AttnArgs2 pk = p;
    if (split) {
        pk.split_heads = 1;
        pk.dK = (bf16*)gqa_scratch;
        pk.dV = (bf16*)gqa_scratch + (int64_t)B * S * Hq * D;
        pk.dk_bs = pk.dv_bs = (int64_t)S * Hq * D;
        pk.dk_hs = pk.dv_hs = D;
        pk.dk_rs = pk.dv_rs = (int64_t)Hq * D;
    }
    dim3 gkv((unsigned)afk_cdiv(S, 128), (unsigned)(split ? Hq : Hkv), (unsigned)B);
    dim3 gq((unsigned)afk_cdiv(S, 128), (unsigned)Hq, (unsigned)B);
    if (D == 128) {
        constexpr int L = 4 * Tile<128>::BYTES;
        static int once = set_lds(attn_bwd_dkdv_lds_kernel<128>, L) + set_lds(attn_bwd_dq_lds_kernel<128>, L);
        (void)once;
        hipLaunchKernelGGL(attn_bwd_dkdv_lds_kernel<128>, gkv, dim3(256), L, st, pk);
        hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<128>, gq, dim3(256), L, st, p);
    } else {
        constexpr int L = 4 * Tile<64>::BYTES;
        static int once = set_lds(attn_bwd_dkdv_lds_kernel<64>, L) + set_lds(attn_bwd_dq_lds_kernel<64>, L);
        (void)once;
        hipLaunchKernelGGL(attn_bwd_dkdv_lds_kernel<64>, gkv, dim3(256), L, st, pk);
        hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<64>, gq, dim3(256), L, st, p);
    }
    if (split) {
        AFK_REQUIRE(dk_hs == D && dv_hs == D && dk_bs == (int64_t)S * dk_rs && dv_bs == (int64_t)S * dv_rs && dk_rs == dv_rs,
                    "afk_attn2_bwd: GQA split path expects dK/dV inside one [B*S, ld] buffer with contiguous heads");
        const int64_t rows = (int64_t)B * S;
        int g = (int)afk_cdiv(rows * (Hkv * D / 8), 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(gqa_reduce_kernel, dim3(g), dim3(256), 0, st, pk.dK, (bf16*)dK, rows, Hkv, group, D, dk_rs);
        hipLaunchKernelGGL(gqa_reduce_kernel, dim3(g), dim3(256), 0, st, pk.dV, (bf16*)dV, rows, Hkv, group, D, dv_rs);
    }
    AFK_LAUNCH_CHECK("afk_attn2_bwd");
    return AFK_OK;
}
