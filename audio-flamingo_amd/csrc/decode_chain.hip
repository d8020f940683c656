// Decode step, single sequence: one launch per Linear, five per decoder layer (round 4).
//
// Rounds 2-3 ran every Linear of a decode token as a weight-streaming GEMV with split-K partials in HBM followed by a glue kernel (decode_glue.hip): ten
// launches per layer, and at ~4 us per dependent launch inside the replayed HIP graph the 28 layers spend 1.1 ms of a 3.6 ms token between kernels
// (the weights stream in 2.5 ms).  Here a group of S waves of ONE block owns eight complete output rows (the waves split K, their partial sums meet in
// LDS), so nothing partial reaches HBM and everything up to the next Linear's input happens where the dot products end:
//     qkv      prologue: RMSNorm of the residual stream (every wave re-derives the row statistic from the L2-resident 7 KiB row)
//              epilogue: + bias, bf16, rotate the four (d, d + D/2) pairs the group owns (modeling_qwen2.py:112-135), q to its buffer, k / v into the KV cache
//     o_proj   epilogue: bf16, + residual (Qwen2DecoderLayer :284)                                           -> x2
//     gate|up  prologue: RMSNorm(x2) (:294, Qwen2RMSNorm :247-252); the group owns gate rows c .. c + 3 AND up rows I + c ..:  silu(g) * u (Qwen2MLP :46-48) -> a
//     down     epilogue: bf16, + residual (:297)                                                             -> x  (the next layer's qkv launch normalises it)
// with the bf16 rounding points of the stand-alone kernels.  Attention is ONE launch as well (attention_decode.hip: the last chunk block of a head merges).
// Eight rows per wave share every load and conversion of x (the shape of the lm_head GEMV, gemm.hip, which streams at 6.9 TB/s); the next chunk's eight
// 16-byte weight loads are issued two chunks ahead of the dot products; weights are loaded non-temporal (streamed once).
#include "common.h"
#include "../../include/afk.h"

namespace {

#ifdef AFK_PROBES
// timing probe (PROBES=1 builds only, tools/probes/probe_decode_chain.py): lane 0 of every wave stamps the 100 MHz wall clock at the phase boundaries
__device__ long long* g_chain_stamps = nullptr;
#define AFK_STAMP(k)                                                                                                              \
    do {                                                                                                                          \
        if (g_chain_stamps && (threadIdx.x & 63) == 0) g_chain_stamps[((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define AFK_STAMP(k)
#endif

#define AFK_CHAIN_BATCH_MAX 8
enum { PRO_PLAIN = 0, PRO_RMS = 1, PRO_PLAIN_LDS = 2 };   // PRO_PLAIN_LDS (matrix-pipe form): input rows as they are, but loaded cooperatively and handed to the MFMA through the LDS strip like PRO_RMS
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_LOGITS = 3, EPI_RESID_NORM = 4 };

struct ChainArgs {
    const bf16* x;        // input row [K]
    const bf16* normw;    // PRO_RMS: RMSNorm weight [K]
    float eps;
    const bf16* W;        // [N, K] row-major (nn.Linear layout)
    int64_t ldw;
    int N, K;
    const bf16* bias;     // EPI_QKV
    const bf16* cos_t;    // [pos][D]
    const bf16* sin_t;
    const int* pos;       // position of the new token (device)
    const int* start;     // cache slot of the new token (device)
    bf16* q_out;          // [Hq * D]
    bf16* Kc;             // [Smax][Hkv * D]
    bf16* Vt;             // [Hkv * D][spad]
    int spad, Hq, Hkv, D;
    const bf16* residual; // EPI_RESID [N]
    bf16* out;            // EPI_RESID [N], EPI_SWIGLU [N / 2]
    float* out_f32;       // EPI_LOGITS [N] (or null)
    // batched form (gemv_chain_batched_kernel): M input rows, row strides in elements
    int M, pos_stride;
    int64_t ldx, ld_res, ld_out, ldq, k_bs, vt_bs;
    // EPI_RESID_NORM (gemv_chain_mfma_kernel): RMSNorm of the finished output rows by the last block to arrive
    const bf16* n2_w;     // [N] weight of the norm that FOLLOWS this Linear (post_attention_layernorm / the next layer's input_layernorm / the final norm)
    bf16* n2_out;         // [M][ld_n2]
    int64_t ld_n2;
    float n2_eps;
    int* n2_counter;      // one zero-initialised int32 (self-resetting)
    float* ss_out;        // EPI_RESID (matrix-pipe form), or null: ss_out[8][blocks] = this block's share of sum(out[m][:]^2) per sequence - the consumer's RMSNorm statistic
    const float* ss_in;   // PRO_RMS, or null: the producer's partial sums [8][ss_nparts] of the rows in x (null: every block takes the statistic from x itself)
    int ss_nparts, ss_first;
    int kil;              // gemv_chain_mfma_kernel: stages of the K loop dealt round-robin to the waves of a block (1) or one contiguous slice per wave (0)
    float* part_val;      // EPI_LOGITS: per row group, the largest logit ...
    int* part_idx;        //             ... and its row (lowest row among equals); or null
};

__device__ __forceinline__ float dot8(const bf16x8 a, const bf16x8 b, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16x2 u = {a[2 * e], a[2 * e + 1]}, v = {b[2 * e], b[2 * e + 1]};
        acc = __builtin_amdgcn_fdot2_f32_bf16(u, v, acc, false);
    }
    return acc;
}

// A row group = R output rows (two runs of R / 2: the rotate-half partners / the gate and up rows of the same columns / simply consecutive rows).
// S waves share one group and split its K chunks (chunk = 512 elements = one 16-byte load per lane and row) round-robin; a 256-thread block holds
// 4 / S groups (S = 8: 512 threads, one group).  S and R trade waves in flight against per-wave work: the narrow Linears of a 7B decoder (qkv: 4 608
// rows, o_proj / down: 3 584) need S = 4 / 8 to put >= 2 300 waves on the chip, gate|up (37 888 rows) and the lm_head (152 064) run S = 1.
#ifdef AFK_CHAIN_WPE
#define AFK_CHAIN_WPE_ATTR __attribute__((amdgpu_waves_per_eu(AFK_CHAIN_WPE)))
#else
#define AFK_CHAIN_WPE_ATTR
#endif
template <int PRO, int EPI, int S, int R>
__global__ __launch_bounds__(64 * (S > 4 ? S : 4)) AFK_CHAIN_WPE_ATTR void gemv_chain_kernel(ChainArgs p, int ngroups) {
    constexpr int G = S >= 4 ? 1 : 4 / S;   // groups per block
    constexpr int HR = R / 2;
    constexpr int LPR = 64 / R;             // after the reduce-scatter lane LPR * j holds row j
    __shared__ float red[G][S][R];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // the wave index as a scalar: rows, row pointers and chunk indices stay in SGPRs
    AFK_STAMP(0);
    const int gi = w / S, ks = w % S;
    const int grp = blockIdx.x * G + gi;
    const bool live = grp < ngroups;
    const int g = live ? grp : 0;
    // ---- the rows of this group: rA .. rA + HR - 1 and rB .. rB + HR - 1
    int rA, rB;
    const int half = p.D >> 1, nq = p.Hq * p.D, nk = p.Hkv * p.D;
    const int rot_groups = (p.Hq + p.Hkv) * half / HR;
    if (EPI == EPI_QKV) {
        if (g < rot_groups) {   // HR rotate-half pairs (d, d + D/2) of one head
            const int per_head = half / HR;
            rA = (g / per_head) * p.D + (g % per_head) * HR;
            rB = rA + half;
        } else {
            rA = nq + nk + R * (g - rot_groups);
            rB = rA + HR;
        }
    } else if (EPI == EPI_SWIGLU) {   // gate rows c .. c + HR - 1 and the up rows of the same columns
        rA = HR * g;
        rB = (p.N >> 1) + rA;
    } else {
        rA = R * g;
        rB = rA + HR;
    }
    // ---- what the epilogue needs besides the sums, requested before anything else so that it is there when the dot products end (round 4: read after
    // them, bias -> position -> cos / sin were three dependent round trips at the tail of a 10 us kernel)
    const int er = lane & (R - 1);
    const int erow = er < HR ? rA + er : rB + er - HR;
    float e_bias = 0.f, e_cos = 0.f, e_sin = 0.f, e_res = 0.f;
    int e_start = 0;
    if (EPI == EPI_QKV) {
        e_bias = (float)p.bias[erow];
        e_start = *p.start;
        if (g < rot_groups) {
            const int64_t ps = (int64_t)(*p.pos) * p.D + erow % p.D;
            e_cos = (float)p.cos_t[ps];
            e_sin = (float)p.sin_t[ps];
        }
    } else if (EPI == EPI_RESID) {
        e_res = (float)p.residual[erow];
    }
    const bf16* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = p.W + (int64_t)(r < HR ? rA + r : rB + r - HR) * p.ldw;
    const int nch = (p.K + 511) >> 9;
    // PRO_RMS: the whole input row FIRST (L2 / HBM latency of a row another XCD has just written) - loads return in order, so requested behind the weight
    // chunks the row would arrive behind them and the statistic, the normalisation and every dot product would start when the last weight byte is in
    constexpr int XCH = 8;   // row chunks held in registers for the statistic (K <= 4096); longer rows: a second pass below
    bf16x8 xs[XCH];
    if (PRO == PRO_RMS) {
#pragma unroll
        for (int cc = 0; cc < XCH; ++cc) {
            const int k = (cc << 9) + lane * 8;
            const bool ok = k < p.K;
            xs[cc] = *(const bf16x8*)(p.x + (ok ? k : 0));
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xs[cc][e] = (bf16)0.f;
            }
        }
    }
    // a chunk = its slice of x (and of the norm weight) and R weight vectors; x is requested ahead of the weights for the same reason as above
    struct Chunk {
        bf16x8 x, nw, w[R];
    };
    auto load_chunk = [&](Chunk& dst, int c) {
        const int k = (c << 9) + lane * 8;
        const bool ok = k < p.K;
        const int kk = ok ? k : 0;   // masked lanes re-read the row's first vector (valid memory); their x is zeroed
        dst.x = *(const bf16x8*)(p.x + kk);
        if (PRO == PRO_RMS) dst.nw = *(const bf16x8*)(p.normw + kk);
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst.x[e] = (bf16)0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) dst.w[r] = __builtin_nontemporal_load((const bf16x8*)(wrow[r] + kk));   // streamed once
    };
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    Chunk q0, q1;
    const int c = ks;
    if (live && c < nch) load_chunk(q0, c);   // two chunks deep in flight while the row statistic below is computed
    if (live && c + S < nch) load_chunk(q1, c + S);
    float rstd = 0.f;
    if (PRO == PRO_RMS) {
        // h = w_norm * bf16(x * rsqrt(mean(x^2) + eps))  (cast BEFORE the weight multiply, Qwen2RMSNorm :247-252); every wave re-derives the statistic from
        // the row: 2 K bytes (L2) against the 16 K bytes of weights per chunk it streams
        float ss = 0.f;
#pragma unroll
        for (int cc = 0; cc < XCH; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)xs[cc][e] * (float)xs[cc][e];
        for (int cc = XCH; cc < nch; ++cc) {
            const int k = (cc << 9) + lane * 8;
            if (k < p.K) {
                const bf16x8 xv = *(const bf16x8*)(p.x + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += (float)xv[e] * (float)xv[e];
            }
        }
        ss = wave_sum(ss);
        rstd = rsqrtf(ss * (1.f / (float)p.K) + p.eps);
    }
    AFK_STAMP(1);
    auto fma_chunk = [&](const Chunk& q) {
        bf16x8 xv = q.x;
        if (PRO == PRO_RMS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = (bf16)((float)q.nw[e] * rbf((float)xv[e] * rstd));   // x = 0 on masked lanes -> h = 0
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot8(q.w[r], xv, acc[r]);
    };
    if (live) {
        for (int cc = c; cc < nch; cc += 2 * S) {
            fma_chunk(q0);
            if (cc == c) AFK_STAMP(2);
            if (cc + 2 * S < nch) load_chunk(q0, cc + 2 * S);
            if (cc + S < nch) {
                fma_chunk(q1);
                if (cc + 3 * S < nch) load_chunk(q1, cc + 3 * S);
            }
        }
    }
    // ---- cross-lane reduce-scatter of the R per-lane sums (R - 1 exchanges, then plain halvings): lane LPR * j ends up with row j
    {
        int n = R;
#pragma unroll
        for (int sft = 5; sft >= 0; --sft) {
            const int st = 1 << sft;
            const bool up = (lane >> sft) & 1;
            if (n > 1) {
                const int h = n >> 1;
#pragma unroll
                for (int i = 0; i < R / 2; ++i)
                    if (i < h) {
                        const float lo = acc[i], hi_ = acc[i + h];
                        acc[i] = (up ? hi_ : lo) + __shfl_xor(up ? lo : hi_, st, 64);
                    }
                n = h;
            } else {
                acc[0] += __shfl_xor(acc[0], st, 64);
            }
        }
    }
    AFK_STAMP(3);
    if ((lane & (LPR - 1)) == 0) red[gi][ks][lane / LPR] = acc[0];
    __syncthreads();
    AFK_STAMP(4);
    if (ks != 0 || !live) return;
    // wave 0 of the group: lane r < R finishes row r (K slices summed in slice order)
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < S; ++q) tot += red[gi][q][er];
    if (EPI == EPI_QKV) {
        const float mine = rbf(tot + e_bias);
        const float other = __shfl_xor(mine, HR, 64);
        if (lane >= R) return;
        if (g < rot_groups) {
            // rotate_half (modeling_qwen2.py:112-135): first half  a cos - b sin,  second half  b cos + a sin  - three bf16 roundings each
            const float rot = er < HR ? -other : other;
            const float o = rbf(rbf_strict(mine * e_cos) + rbf_strict(rot * e_sin));   // rbf_strict: contraction-proof (common.h)
            if (erow < nq) p.q_out[erow] = (bf16)o;
            else p.Kc[(int64_t)e_start * nk + (erow - nq)] = (bf16)o;
        } else {
            p.Vt[(int64_t)(erow - nq - nk) * p.spad + e_start] = (bf16)mine;
        }
    } else if (EPI == EPI_RESID) {
        if (lane >= R) return;
        p.out[erow] = (bf16)(rbf(tot) + e_res);
    } else if (EPI == EPI_SWIGLU) {
        const float mine = rbf(tot);
        const float u = __shfl_xor(mine, HR, 64);
        if (lane >= HR) return;
        p.out[erow] = (bf16)(rbf(mine * sigmoid_f(mine)) * u);
    } else {
        const float mine = rbf(tot);   // the bf16 logit nn.Linear returns, widened (what .float() of it gives)
        if (p.out_f32 && lane < R) p.out_f32[erow] = mine;
        if (p.part_val) {              // greedy decoding: the group's (max, argmax), ties to the lowest row (torch.argmax's order) - lanes 0 .. R-1 hold rows in order
            float bv = mine;
            int bi = erow;
#pragma unroll
            for (int o = R / 2; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { p.part_val[g] = bv; p.part_idx[g] = bi; }
        }
    }
}

// greedy token selection + everything else between two decode steps of ONE sequence, in one launch (round 4: argmax, the copy into the next-token buffer, the
// position update and the embedding lookup of the next step were four launches of ~5-10 us each): reduce the lm_head launch's per-group (max, argmax) pairs,
// write the token (next_token, tokens_out[cur + tok_off]), advance [key-range end, cache slot, position] and fetch the token's embedding row for the next step
__global__ __launch_bounds__(1024) void decode_select_greedy_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int nparts,
                                                                    long long* __restrict__ next_token, long long* __restrict__ tokens_out, int tok_off,
                                                                    int* __restrict__ state, const bf16* __restrict__ emb, int64_t ld_emb, int H,
                                                                    bf16* __restrict__ x_out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int tok;
    const int t = threadIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = t; i < nparts; i += 8 * 1024) {   // eight pairs per thread and round trip (19 008 pairs for the AF3 vocabulary: three rounds)
        float v[8];
        int ix[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = min(i + 1024 * u, nparts - 1);   // clamped: a repeated pair never changes the result
            v[u] = part_val[j];
            ix[u] = part_idx[j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > bv || (v[u] == bv && ix[u] < bi)) { bv = v[u]; bi = ix[u]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((t & 63) == 0) { sv[t >> 6] = bv; si[t >> 6] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;   // every logit NaN / -inf: torch.argmax answers 0 for an all -inf row
        tok = bi;
        next_token[0] = bi;
        if (tokens_out) tokens_out[state[2] + tok_off] = bi;
        state[1] += 1;   // key-range end
        state[2] += 1;   // cache slot of the next token
        state[3] += 1;   // its position
    }
    __syncthreads();
    const bf16* row = emb + (int64_t)tok * ld_emb;
    for (int k = t * 4; k < H; k += 4096) *(bf16x4*)(x_out + k) = *(const bf16x4*)(row + k);
}

// 2 .. 8 sequences per step: the same row groups and K split, MB input rows against every weight vector (the weights are still read once per step).
// The inputs arrive normalised (the RMSNorm of MB rows in every wave's prologue would cost more VALU than the dot products): PRO_PLAIN only, eight rows per
// group; R x MB sums per lane, reduce-scattered so that lane (r * MB + m) finishes row r of sequence m.
template <int EPI, int S, int MB>
__global__ __launch_bounds__(64 * (S > 4 ? S : 4)) void gemv_chain_batched_kernel(ChainArgs p, int ngroups) {
    constexpr int R = 8, HR = 4, V = R * MB;
    constexpr int G = S >= 4 ? 1 : 4 / S;
    constexpr int LPV = 64 / V;
    static_assert(V <= 64 && (MB & (MB - 1)) == 0, "R x MB sums are reduce-scattered over one wave");
    __shared__ float red[G][S][V];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gi = w / S, ks = w % S;
    const int grp = blockIdx.x * G + gi;
    const bool live = grp < ngroups;
    const int g = live ? grp : 0;
    int rA, rB;
    const int half = p.D >> 1, nq = p.Hq * p.D, nk = p.Hkv * p.D;
    const int rot_groups = (p.Hq + p.Hkv) * half / HR;
    if (EPI == EPI_QKV) {
        if (g < rot_groups) {
            const int per_head = half / HR;
            rA = (g / per_head) * p.D + (g % per_head) * HR;
            rB = rA + half;
        } else {
            rA = nq + nk + R * (g - rot_groups);
            rB = rA + HR;
        }
    } else if (EPI == EPI_SWIGLU) {
        rA = HR * g;
        rB = (p.N >> 1) + rA;
    } else {
        rA = R * g;
        rB = rA + HR;
    }
    const bf16* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = p.W + (int64_t)(r < HR ? rA + r : rB + r - HR) * p.ldw;
    const int nch = (p.K + 511) >> 9;
    auto load_w = [&](bf16x8(&dst)[R], int c) {
        const int k = (c << 9) + lane * 8;
        const int kk = k < p.K ? k : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r] = __builtin_nontemporal_load((const bf16x8*)(wrow[r] + kk));
    };
    float acc[R * MB];
#pragma unroll
    for (int i = 0; i < R * MB; ++i) acc[i] = 0.f;
    bf16x8 w0[R], w1[R];
    const int c = ks;
    if (live && c < nch) load_w(w0, c);
    if (live && c + S < nch) load_w(w1, c + S);
    auto fma_chunk = [&](const bf16x8(&wv)[R], int cc) {
        const int k = (cc << 9) + lane * 8;
        const bool ok = k < p.K;
        const int kk = ok ? k : 0;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            bf16x8 xv = *(const bf16x8*)(p.x + (int64_t)min(m, p.M - 1) * p.ldx + kk);   // rows >= M repeat the last row; their sums are dropped
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = (bf16)0.f;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r * MB + m] = dot8(wv[r], xv, acc[r * MB + m]);
        }
    };
    if (live) {
        for (int cc = c; cc < nch; cc += 2 * S) {
            fma_chunk(w0, cc);
            if (cc + 2 * S < nch) load_w(w0, cc + 2 * S);
            if (cc + S < nch) {
                fma_chunk(w1, cc + S);
                if (cc + 3 * S < nch) load_w(w1, cc + 3 * S);
            }
        }
    }
    {   // reduce-scatter of the V per-lane sums: lane LPV * j ends up with sum j = r * MB + m
        int n = V;
#pragma unroll
        for (int sft = 5; sft >= 0; --sft) {
            const int st = 1 << sft;
            const bool up = (lane >> sft) & 1;
            if (n > 1) {
                const int h = n >> 1;
#pragma unroll
                for (int i = 0; i < V / 2; ++i)
                    if (i < h) {
                        const float lo = acc[i], hi_ = acc[i + h];
                        acc[i] = (up ? hi_ : lo) + __shfl_xor(up ? lo : hi_, st, 64);
                    }
                n = h;
            } else {
                acc[0] += __shfl_xor(acc[0], st, 64);
            }
        }
    }
    if ((lane & (LPV - 1)) == 0) red[gi][ks][lane / LPV] = acc[0];
    __syncthreads();
    if (ks != 0 || !live) return;
    const int j = lane & (V - 1);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < S; ++q) tot += red[gi][q][j];
    const int r = j / MB, m = j % MB;
    const int row = r < HR ? rA + r : rB + r - HR;
    const bool mine_ok = lane < V && m < p.M;
    if (EPI == EPI_QKV) {
        const float mine = rbf(tot + (float)p.bias[row]);
        const float other = __shfl_xor(mine, HR * MB, 64);
        if (!mine_ok) return;
        const int start = *p.start;
        if (g < rot_groups) {
            const int64_t ps = (int64_t)p.pos[m * p.pos_stride] * p.D + row % p.D;
            const float rot = r < HR ? -other : other;
            const float o = rbf(rbf_strict(mine * (float)p.cos_t[ps]) + rbf_strict(rot * (float)p.sin_t[ps]));   // rbf_strict: contraction-proof (common.h)
            if (row < nq) p.q_out[m * p.ldq + row] = (bf16)o;
            else p.Kc[m * p.k_bs + (int64_t)start * nk + (row - nq)] = (bf16)o;
        } else {
            p.Vt[m * p.vt_bs + (int64_t)(row - nq - nk) * p.spad + start] = (bf16)mine;
        }
    } else if (EPI == EPI_RESID) {
        if (!mine_ok) return;
        p.out[m * p.ld_out + row] = (bf16)(rbf(tot) + (float)p.residual[m * p.ld_res + row]);
    } else if (EPI == EPI_SWIGLU) {
        const float mine = rbf(tot);
        const float u = __shfl_xor(mine, HR * MB, 64);
        if (!mine_ok || r >= HR) return;
        p.out[m * p.ld_out + row] = (bf16)(rbf(mine * sigmoid_f(mine)) * u);
    } else {
        if (!mine_ok) return;
        p.out_f32[m * p.ld_out + row] = rbf(tot);
    }
}

// The batched form on the matrix pipe.  With 8 input rows the dot-product form above issues 256 v_dot2 per 8 KiB of weights and wave - it is VALU-bound at
// ~5 TB/s (gate|up 53.6 us against 42.5 us for one row).  v_mfma_f32_32x32x16_bf16 does the same contraction for 32 weight rows x up to 32 input rows in 32
// cycles per KiB of weights and wave.  A group = 32 output rows (two runs of 16); S waves of a block split K into contiguous slices of 64-element blocks.
// The weights cannot feed the MFMA straight from global memory at speed: its A operand wants lane -> (row l & 31, 16-byte piece l >> 5), i.e. 32 rows x 32
// bytes per load instruction - measured 2.4 TB/s (quarter lines).  So a wave loads row-contiguous (8 rows x 128 bytes per instruction, non-temporal), parks two
// 64-element blocks in its PRIVATE 8 KiB of LDS (16-byte pieces xor-swizzled by (row >> 1) & 7: conflict-free writes and fragment reads, as gemm.hip) and
// reads the fragments back - LDS operations of one wave execute in order, so no barrier is involved until the slices' 32 x M sums meet at the end.  The next
// two blocks' loads are issued before the current ones are multiplied.  The B operand (lane -> sequence l & 31, same piece) comes from the L2-resident input
// rows; lanes beyond M repeat row M - 1 and feed result columns nobody reads.
// RG = 16 (the narrow Linears: twice the groups, so twice the waves and bytes in flight): rows 16 .. 31 of the A operand repeat rows 0 .. 15 and their results are dropped.
// PRO = PRO_RMS (round 6): the input rows arrive RAW and the RMSNorm in front of this Linear (Qwen2RMSNorm :247-252) happens here, as in the single-sequence kernel -
// no norm launch, no hand-over.  Prologue: the S waves of a block sum the squares of the M rows together (each thread a strided share, per-row partials folded in
// wave order through LDS) - the K split covers every column, so every block can do this on its own.  Per stage a wave then loads its NB blocks of the M rows
// COOPERATIVELY (lane -> row lane >> 3, 16-byte piece lane & 7: every lane carries real data), normalises them (cast before the weight multiply), parks the bf16
// rows in a wave-private LDS strip and reads the MFMA's B fragments back from there (LDS operations of one wave execute in order): 2 NB loads and ~40 VALU
// instructions per block instead of 4 masked fragment loads per block and a kernel launch per norm.
template <int EPI, int S, int RG, int PRO = PRO_PLAIN>
__global__ __launch_bounds__(64 * S) void gemv_chain_mfma_kernel(ChainArgs p) {
    constexpr int R = RG, HR = RG / 2, MBX = AFK_CHAIN_BATCH_MAX;
    constexpr int JR = RG / 8;          // load instructions per 64-element block (8 rows x 128 bytes each)
    constexpr int NB = 8 / JR;          // blocks per stage: eight loads in flight per register set either way
    constexpr int BLK = RG * 128;       // bytes of a staged block
    static_assert(S >= 4 && S <= 16 && (RG == 32 || RG == 16), "256 finishing threads; S x 8 KiB of (dynamic) LDS");
    extern __shared__ __attribute__((aligned(16))) char stage_dyn[];
    char(*stage)[8192] = (char(*)[8192])stage_dyn;
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x;
    int rA, rB;
    const int half = p.D >> 1, nq = p.Hq * p.D, nk = p.Hkv * p.D;
    const int rot_groups = (p.Hq + p.Hkv) * half / HR;
    if (EPI == EPI_QKV) {
        if (g < rot_groups) {
            const int per_head = half / HR;
            rA = (g / per_head) * p.D + (g % per_head) * HR;
            rB = rA + half;
        } else {
            rA = nq + nk + R * (g - rot_groups);
            rB = rA + HR;
        }
    } else if (EPI == EPI_SWIGLU) {
        rA = HR * g;
        rB = (p.N >> 1) + rA;
    } else {
        rA = R * g;
        rB = rA + HR;
    }
    // this wave's K slice in blocks of 64 elements (K % 64 == 0)
    const int nkb = p.K >> 6;
    // K split over the S waves: contiguous slices (p.kil == 0: wave w owns blocks [w per, (w + 1) per)), or stages dealt round-robin (p.kil == 1: wave w owns the
    // NB-block stages w, w + S, ... - at any moment the waves of a block stream ADJACENT 128 NB-byte pieces of the same rows, as the single-sequence kernel's chunks do)
    const int per = (nkb + S - 1) / S;
    const int b0 = p.kil ? w * NB : w * per, b1 = p.kil ? nkb : min(b0 + per, nkb);
    const int STEP = p.kil ? S * NB : NB;   // blocks from one stage of this wave to its next
    // global side: load j of a block covers group rows 8 j + (lane >> 3), 16-byte piece lane & 7
    const int rowl = lane >> 3, piece = lane & 7;
    const bf16* wp[JR];
    uint32_t wr_off[JR];
    char* my = &stage[w][0];
#pragma unroll
    for (int j = 0; j < JR; ++j) {
        const int r = 8 * j + rowl;
        wp[j] = p.W + (int64_t)(r < HR ? rA + r : rB + r - HR) * p.ldw + piece * 8;
        wr_off[j] = r * 128 + ((piece ^ ((r >> 1) & 7)) << 4);
    }
    const int fr = l31 & (RG - 1);
    const uint32_t rd_row = fr * 128, rd_swz = (fr >> 1) & 7;
    const bf16* xp = p.x + (int64_t)min(l31, p.M - 1) * p.ldx + hi * 8;
    auto gload = [&](bf16x8(&dst)[8], int kb) {   // NB blocks; a block past the slice repeats the slice's last one (valid memory) and is skipped below
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int kk = min(kb + b, b1 - 1) << 6;
#pragma unroll
            for (int j = 0; j < JR; ++j) dst[b * JR + j] = __builtin_nontemporal_load((const bf16x8*)(wp[j] + kk));
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 ga[8], gb[8];
    // ---- PRO_RMS: strip of normalised rows [NB blocks][8 rows][128 B] behind the S weight stages, and the row statistic
    constexpr int HXB = NB * 1024;
    char* hx = stage_dyn + S * 8192 + w * HXB;
    const int xm = lane >> 3, xpc = lane & 7;                       // cooperative row / 16-byte piece of this lane
    const uint32_t hx_wr = xm * 128 + ((xpc ^ xm) << 4);           // piece index xor row: conflict-free 16-byte writes and fragment reads
    const uint32_t hx_rd_row = (l31 & 7) * 128, hx_rd_swz = l31 & 7;
    const bf16* xrow = p.x + (int64_t)min(xm, p.M - 1) * p.ldx + xpc * 8;
    float my_rstd = 0.f;
    if constexpr (PRO == PRO_RMS) {
        // the partial sums are requested AHEAD of the first weights (loads return in order: behind them the statistic would only be known once 8 KiB of weights had
        // landed): qkv 15.5 -> 14.2 us, lm_head 175 -> 171.5 at M = 8, gate|up level (AFK_CHAIN_SS_FIRST=0: A/B)
        const bool ss_first = p.ss_in != nullptr && p.ss_first;
        if (b0 < b1 && !ss_first) gload(ga, b0);   // the first weights travel while the statistic is taken
#ifdef AFK_PROBES
        if (p.eps < 0.f) my_rstd = 1.f;   // timing probe (wrong results): no statistic pass
        else
#endif
        if (p.ss_in != nullptr) {
            // the Linear that wrote these rows left its per-block sums of squares (EPI_RESID, ss_out): every WAVE folds the ss_nparts x 8 floats itself - lane -> row
            // lane >> 3, parts (lane & 7) + 8 i - and the eight lanes of a row meet through three shuffles: no block barrier, one L2 round trip beside the first weights
            constexpr int NP4 = 8;   // up to 256 parts (N <= 4096 at 16-row groups), laid out [row][part]: a lane takes parts 4 (lane & 7) + 32 i .. + 3 of its row as
                                     // ONE 16-byte load (a quarter of the load instructions of a [part][row] layout); every load is issued before the first sum -
                                     // a loop of load -> add is a round trip per part
            const float* ssr = p.ss_in + (int64_t)xm * p.ss_nparts;
            f32x4 v[NP4];
#pragma unroll
            for (int i = 0; i < NP4; ++i) {
                const int gp = 4 * (xpc + 8 * i);
                v[i] = gp < p.ss_nparts ? *(const f32x4*)(ssr + gp) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (b0 < b1 && ss_first) gload(ga, b0);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < NP4; ++i) a += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            my_rstd = rsqrtf(a * (1.f / (float)p.K) + p.eps);
        } else {
        constexpr int T = 64 * S;
        const int nch = p.K >> 3;     // 16-byte chunks per row
        float ss[MBX];
        constexpr int CPT = (512 + T - 1) / T;   // chunks per thread and row for K <= 4096, all requested before the first is used (longer rows: the loop below)
        bf16x8 xv[MBX][CPT];
#pragma unroll
        for (int m = 0; m < MBX; ++m)
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = threadIdx.x + T * i;
                const bool ok = m < p.M && c < nch;
                xv[m][i] = *(const bf16x8*)(p.x + (int64_t)(ok ? m : 0) * p.ldx + (ok ? c : 0) * 8);
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[m][i][e] = (bf16)0.f;
                }
            }
#pragma unroll
        for (int m = 0; m < MBX; ++m) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) a += (float)xv[m][i][e] * (float)xv[m][i][e];
            if (m < p.M) {
                for (int c = threadIdx.x + T * CPT; c < nch; c += T) {
                    const bf16x8 v = *(const bf16x8*)(p.x + (int64_t)m * p.ldx + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += (float)v[e] * (float)v[e];
                }
            }
            ss[m] = wave_sum(a);
        }
        float* ssred = (float*)(stage_dyn + S * 8192);   // [S][MBX], at the head of the (still unused) strips: S x (8 + NB) KiB keeps gate|up at four blocks per CU
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MBX; ++m) ssred[w * MBX + m] = ss[m];
        }
        __syncthreads();
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < S; ++q) a += ssred[q * MBX + xm];      // wave order: bit-reproducible
        my_rstd = rsqrtf(a * (1.f / (float)p.K) + p.eps);
        __syncthreads();   // wave 0's strip is about to be written
        }
    }
    // the B fragments (L2-resident input rows), two blocks at a time.  Column j of the MFMA result depends on column j of B alone and only columns < M are read
    // afterwards, so only the lanes of real sequences load (a quarter of the lanes at M = 8 - the other lanes' registers keep their zeros; no effect on the time measured, round 6).
    // HOIST (S <= 8: the registers are there): ALL of a four-block stage's fragments are requested ahead of the next stage's weight loads - loads return in order,
    // so the second pair, requested behind them (round 4), could only be used once the whole next stage had landed.
    constexpr bool HOIST = NB == 4 && S <= 8;
    const bool xlane = l31 < p.M;
    auto xload2 = [&](bf16x8(&dst)[8], int kb) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int kk = min(kb + b, b1 - 1) << 6;
            if (xlane) {
#pragma unroll
                for (int st = 0; st < 4; ++st) dst[b * 4 + st] = *(const bf16x8*)(xp + kk + st * 16);
            }
        }
    };
    auto mma2 = [&](const bf16x8(&xv)[8], int kb, int bb) {   // blocks bb, bb + 1 of the stage
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (kb + bb + b < b1) {   // wave-uniform
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const bf16x8 wf = *(const bf16x8*)(my + (bb + b) * BLK + rd_row + ((((2 * st + hi)) ^ rd_swz) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xv[b * 4 + st], acc, 0, 0, 0);
                }
            }
        }
    };
    bf16x8 xa[8], xb[HOIST ? 8 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xa[i][e] = (bf16)0.f;
            if (HOIST) xb[i][e] = (bf16)0.f;
        }
    auto consume = [&](bf16x8(&cur)[8], bf16x8(&nxt)[8], int kb) {
        xload2(xa, kb);                              // ahead of the next stage's weight loads: loads return in order
        if constexpr (HOIST) xload2(xb, kb + 2);
        if (kb + STEP < b1) gload(nxt, kb + STEP);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < JR; ++j) *(bf16x8*)(my + b * BLK + wr_off[j]) = cur[b * JR + j];
        mma2(xa, kb, 0);
        if constexpr (HOIST) {
            mma2(xb, kb, 2);
        } else if (NB == 4) {
            xload2(xa, kb + 2);
            mma2(xa, kb, 2);
        }
    };
    if constexpr (PRO == PRO_RMS || PRO == PRO_PLAIN_LDS) {
        if constexpr (PRO == PRO_PLAIN_LDS) {
            if (b0 < b1) gload(ga, b0);
        }
        auto consumeN = [&](bf16x8(&cur)[8], bf16x8(&nxt)[8], int kb) {
            bf16x8 xr[NB], gw[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {   // ahead of the next stage's weight loads: loads return in order
                const int kk = min(kb + b, b1 - 1) << 6;
                xr[b] = *(const bf16x8*)(xrow + kk);
                if constexpr (PRO == PRO_RMS) gw[b] = *(const bf16x8*)(p.normw + kk + xpc * 8);
            }
            if (kb + STEP < b1) gload(nxt, kb + STEP);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < JR; ++j) *(bf16x8*)(my + b * BLK + wr_off[j]) = cur[b * JR + j];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                bf16x8 h = xr[b];
                if constexpr (PRO == PRO_RMS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (bf16)((float)gw[b][e] * rbf((float)xr[b][e] * my_rstd));   // cast BEFORE the weight multiply (:250-252)
                }
                *(bf16x8*)(hx + b * 1024 + hx_wr) = h;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (kb + b < b1) {   // wave-uniform
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const bf16x8 wf = *(const bf16x8*)(my + b * BLK + rd_row + ((((2 * st + hi)) ^ rd_swz) << 4));
                        const bf16x8 xf = *(const bf16x8*)(hx + b * 1024 + hx_rd_row + ((((2 * st + hi)) ^ hx_rd_swz) << 4));   // lanes >= 8 repeat rows 0 .. 7: columns nobody reads
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
                    }
                }
            }
        };
        if (b0 < b1) {   // ga was requested in the prologue
            for (int kb = b0; kb < b1; kb += 2 * STEP) {
                consumeN(ga, gb, kb);
                if (kb + STEP < b1) consumeN(gb, ga, kb + STEP);
            }
        }
    } else if (b0 < b1) {
        gload(ga, b0);
        for (int kb = b0; kb < b1; kb += 2 * STEP) {
            consume(ga, gb, kb);
            if (kb + STEP < b1) consume(gb, ga, kb + STEP);
        }
    }
    // D[i][j]: lane holds column j = l31 (the sequence), rows i = 8 q + 4 hi + e in acc[4 q + e]; the wave's sums go to the head of its own staging area
    float* red = (float*)my;   // [R][MBX]
    if (l31 < MBX) {
#pragma unroll
        for (int q = 0; q < RG / 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(8 * q + 4 * hi + e) * MBX + l31] = acc[4 * q + e];
    }
    __syncthreads();
    if constexpr (EPI == EPI_RESID_NORM) {
        // Linear + residual as EPI_RESID, then the RMSNorm that follows it (Qwen2DecoderLayer :284 -> :294, :297 -> the next layer's :271 / Qwen2Model.norm; Qwen2RMSNorm
        // :247-252) WITHOUT a launch of its own: the statistic needs every column, i.e. every block of this launch, so the rows go out as agent-scope write-through
        // stores (two bf16 per 32-bit word), each block bumps a counter behind its drained stores, and the block that arrives last re-reads the M x N outputs with
        // agent-scope loads, normalises them and writes h.  The hand-over is afk_attn_decode_fused's (attention_decode.hip); nobody waits for anybody.
        __shared__ int last_flag;
        float* fin = (float*)&stage[0][4096];
        const int t = threadIdx.x;
        if (t < R * MBX) {
            const int r = t / MBX, m = t % MBX;
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < S; ++q) tot += ((const float*)&stage[q][0])[r * MBX + m];
            const int row = rA + r;   // rB = rA + HR: the R rows of a group are consecutive
            fin[r * MBX + m] = m < p.M ? (float)(bf16)(rbf(tot) + (float)p.residual[m * p.ld_res + row]) : 0.f;
        }
        __syncthreads();
        if (t < HR * MBX) {   // thread (pair j, m): rows rA + 2 j, rA + 2 j + 1 of sequence m as one word
            const int j = t / MBX, m = t % MBX;
            if (m < p.M) {
                const bf16x2 pr = {(bf16)fin[(2 * j) * MBX + m], (bf16)fin[(2 * j + 1) * MBX + m]};
                __hip_atomic_store((uint32_t*)(p.out + m * p.ld_out + rA + 2 * j), __builtin_bit_cast(uint32_t, pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing waves drain their write-through stores themselves (attention_decode.hip: no fence does it here)
        __syncthreads();
        if (t == 0) {
            const int prev = __hip_atomic_fetch_add(p.n2_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = prev == (int)gridDim.x - 1;
            if (last_flag) __hip_atomic_store(p.n2_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-resetting (graph replays)
        }
        __syncthreads();
        if (!last_flag) return;
        constexpr int T = 64 * S, WPT = 4;   // the launcher guarantees N / 2 <= WPT * T words per row
        const int nw2 = p.N >> 1;
        uint32_t xw[MBX][WPT];
#pragma unroll
        for (int m = 0; m < MBX; ++m)
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int w2 = t + T * i;
                xw[m][i] = (m < p.M && w2 < nw2) ? __hip_atomic_load((const uint32_t*)(p.out + m * p.ld_out) + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
        uint32_t nwv[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) nwv[i] = (t + T * i) < nw2 ? ((const uint32_t*)p.n2_w)[t + T * i] : 0u;
        float* ssred = (float*)&stage[0][0];   // [S][MBX]: the K-slice sums there have been consumed
        float ss[MBX];
#pragma unroll
        for (int m = 0; m < MBX; ++m) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const float lo = __uint_as_float(xw[m][i] << 16), hi_ = __uint_as_float(xw[m][i] & 0xffff0000u);
                a += lo * lo + hi_ * hi_;
            }
            ss[m] = wave_sum(a);
        }
        __syncthreads();   // every thread has read its slice sums / fin before the area is reused
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MBX; ++m) ssred[w * MBX + m] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MBX; ++m) {
            if (m >= p.M) break;
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < S; ++q) a += ssred[q * MBX + m];
            const float rstd = rsqrtf(a * (1.f / (float)p.N) + p.n2_eps);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int w2 = t + T * i;
                if (w2 < nw2) {
                    const float x0 = __uint_as_float(xw[m][i] << 16), x1 = __uint_as_float(xw[m][i] & 0xffff0000u);
                    const float g0 = __uint_as_float(nwv[i] << 16), g1 = __uint_as_float(nwv[i] & 0xffff0000u);
                    const bf16x2 o = {(bf16)(g0 * rbf(x0 * rstd)), (bf16)(g1 * rbf(x1 * rstd))};   // cast BEFORE the weight multiply (:250-252)
                    ((uint32_t*)(p.n2_out + m * p.ld_n2))[w2] = __builtin_bit_cast(uint32_t, o);
                }
            }
        }
        return;
    }
    if (threadIdx.x >= R * MBX) return;
    const int r = threadIdx.x / MBX, m = threadIdx.x % MBX;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < S; ++q) tot += ((const float*)&stage[q][0])[r * MBX + m];   // K slices summed in slice order
    float* fin = (float*)&stage[0][4096];   // [R][MBX], nobody's sums live there
    const int row = r < HR ? rA + r : rB + r - HR;
    const bool valid = m < p.M;
    if (EPI == EPI_QKV) {
        const float mine = rbf(tot + (float)p.bias[row]);
        fin[r * MBX + m] = mine;
        __syncthreads();
        if (!valid) return;
        const float other = fin[(r ^ HR) * MBX + m];
        const int start = *p.start;
        if (g < rot_groups) {
            const int64_t ps = (int64_t)p.pos[m * p.pos_stride] * p.D + row % p.D;
            const float rot = r < HR ? -other : other;
            const float o = rbf(rbf_strict(mine * (float)p.cos_t[ps]) + rbf_strict(rot * (float)p.sin_t[ps]));   // rbf_strict: contraction-proof (common.h)
            if (row < nq) p.q_out[m * p.ldq + row] = (bf16)o;
            else p.Kc[m * p.k_bs + (int64_t)start * nk + (row - nq)] = (bf16)o;
        } else {
            p.Vt[m * p.vt_bs + (int64_t)(row - nq - nk) * p.spad + start] = (bf16)mine;
        }
    } else if (EPI == EPI_RESID) {
        const bf16 ov = (bf16)(rbf(tot) + (float)p.residual[min(m, p.M - 1) * p.ld_res + row]);
        if (valid) p.out[m * p.ld_out + row] = ov;
        if (p.ss_out != nullptr) {   // this block's share of the NEXT norm's statistic: sum over its R columns of out^2 per sequence, rows in order
            fin[r * MBX + m] = valid ? (float)ov * (float)ov : 0.f;
            __syncthreads();
            if (threadIdx.x < MBX) {
                float a = 0.f;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) a += fin[rr * MBX + threadIdx.x];
                p.ss_out[(int64_t)threadIdx.x * gridDim.x + g] = a;   // [row][part]
            }
        }
    } else if (EPI == EPI_SWIGLU) {
        const float mine = rbf(tot);
        fin[r * MBX + m] = mine;
        __syncthreads();
        if (!valid || r >= HR) return;
        const float u = fin[(r + HR) * MBX + m];
        p.out[m * p.ld_out + row] = (bf16)(rbf(mine * sigmoid_f(mine)) * u);
    } else {
        if (!valid) return;
        p.out_f32[m * p.ld_out + row] = rbf(tot);
    }
}

// 9 .. 32 sequences per step (round 6): the norm-in-prologue / LDS-strip launches above for NG = 2 or 4 GROUPS of eight sequences in one pass over the weights.
// Rounds 3-5 sent batches above eight down the split-K tile path (B = 9: 5.7 ms per step against 3.6 at B = 8 - fewer tokens per second than the smaller batch).  A
// group is an independent instance of the eight-sequence arithmetic (its own rows, statistic and epilogue - index for index the code of gemv_chain_mfma_kernel) and
// its sequences are eight more COLUMNS of the same MFMA: the strip holds 8 NG rows, one fragment read and one v_mfma_f32_32x32x16_bf16 per 16 reduction elements
// serve all groups - the matrix pipe and the LDS do the work of an eight-sequence step (a first form ran the MFMAs group by group: B = 17 6.75 ms, slower than
// the tile path).  The sums of a column are those of the eight-sequence launch bit for bit.  Every block barrier is executed by every thread (no early exits: the
// epilogue runs once per group).  ss_in / ss_out are [NG][8][parts].
template <int EPI, int S, int RG, int PRO, int NG>
__global__ __launch_bounds__(64 * S) void gemv_chain_mfma_ng_kernel(ChainArgs p) {
    constexpr int R = RG, HR = RG / 2, MBX = AFK_CHAIN_BATCH_MAX;
    constexpr int JR = RG / 8, NB = 8 / JR, BLK = RG * 128;
    static_assert(S >= 4 && S <= 8 && (RG == 32 || RG == 16) && (PRO == PRO_RMS || PRO == PRO_PLAIN_LDS) && NG >= 2 && NG <= 4, "see gemv_chain_mfma_kernel");
    static_assert(EPI == EPI_QKV || EPI == EPI_RESID || EPI == EPI_SWIGLU || EPI == EPI_LOGITS, "no hand-over epilogue here");
    extern __shared__ __attribute__((aligned(16))) char stage_dyn[];
    char(*stage)[8192] = (char(*)[8192])stage_dyn;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = blockIdx.x;
    int rA, rB;
    const int half = p.D >> 1, nq = p.Hq * p.D, nk = p.Hkv * p.D;
    const int rot_groups = (p.Hq + p.Hkv) * half / HR;
    if (EPI == EPI_QKV) {
        if (g < rot_groups) {
            const int per_head = half / HR;
            rA = (g / per_head) * p.D + (g % per_head) * HR;
            rB = rA + half;
        } else {
            rA = nq + nk + R * (g - rot_groups);
            rB = rA + HR;
        }
    } else if (EPI == EPI_SWIGLU) {
        rA = HR * g;
        rB = (p.N >> 1) + rA;
    } else {
        rA = R * g;
        rB = rA + HR;
    }
    const int nkb = p.K >> 6;
    const int b0 = w * NB, b1 = nkb, STEP = S * NB;   // stages dealt round-robin to the waves
    const int rowl = lane >> 3, piece = lane & 7;
    const bf16* wp[JR];
    uint32_t wr_off[JR];
    char* my = &stage[w][0];
#pragma unroll
    for (int j = 0; j < JR; ++j) {
        const int r = 8 * j + rowl;
        wp[j] = p.W + (int64_t)(r < HR ? rA + r : rB + r - HR) * p.ldw + piece * 8;
        wr_off[j] = r * 128 + ((piece ^ ((r >> 1) & 7)) << 4);
    }
    const int fr = l31 & (RG - 1);
    const uint32_t rd_row = fr * 128, rd_swz = (fr >> 1) & 7;
    auto gload = [&](bf16x8(&dst)[8], int kb) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int kk = min(kb + b, b1 - 1) << 6;
#pragma unroll
            for (int j = 0; j < JR; ++j) dst[b * JR + j] = __builtin_nontemporal_load((const bf16x8*)(wp[j] + kk));
        }
    };
    f32x16 acc;   // ONE accumulator block: the 8 NG sequences are the columns of the same MFMA (column 8 sg + m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 ga[8], gb[8];
    constexpr int MC = MBX * NG;                   // sequences = MFMA columns in use
    constexpr int HB = 2;                          // blocks per strip fill (a four-block stage of the 16-row groups goes through the strip in two halves)
    constexpr int HXB = HB * MC * 128;             // strip: [HB blocks][MC rows][128 B]
    char* hx = stage_dyn + S * 8192 + w * HXB;
    const int xm = lane >> 3, xpc = lane & 7;
    const uint32_t hx_wr = xm * 128 + ((xpc ^ xm) << 4);   // + sg * 1024: row 8 sg + xm (the swizzle uses row & 7 = xm)
    const uint32_t hx_rd_row = (l31 & (MC - 1)) * 128, hx_rd_swz = l31 & 7;
    auto Mg = [&](int sg) { return max(0, min(MBX, p.M - MBX * sg)); };   // sequences of group sg (0: an empty group computes on row 0 and stores nothing)
    const bf16* xrow[NG];
#pragma unroll
    for (int sg = 0; sg < NG; ++sg) xrow[sg] = p.x + (int64_t)min(MBX * sg + min(xm, max(Mg(sg), 1) - 1), p.M - 1) * p.ldx + xpc * 8;
    float my_rstd[NG];
#pragma unroll
    for (int sg = 0; sg < NG; ++sg) my_rstd[sg] = 0.f;
    if constexpr (PRO == PRO_RMS) {
        if (p.ss_in != nullptr) {   // per group: the producer's partial sums, all loads of a group in flight together, before the first weights
#pragma unroll
            for (int sg = 0; sg < NG; ++sg) {
                constexpr int NP4 = 8;
                const float* ssr = p.ss_in + ((int64_t)sg * MBX + xm) * p.ss_nparts;   // [group][row][part]
                f32x4 v[NP4];
#pragma unroll
                for (int i = 0; i < NP4; ++i) {
                    const int gp = 4 * (xpc + 8 * i);
                    v[i] = gp < p.ss_nparts ? *(const f32x4*)(ssr + gp) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (sg == 0 && b0 < b1) gload(ga, b0);   // the first weights right behind the first group's partial sums
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < NP4; ++i) a += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                a += __shfl_xor(a, 1, 64);
                a += __shfl_xor(a, 2, 64);
                a += __shfl_xor(a, 4, 64);
                my_rstd[sg] = rsqrtf(a * (1.f / (float)p.K) + p.eps);
            }
        } else {                    // the first layer's rows: the block takes the statistic itself, group after group (two barriers each, executed by every thread)
            if (b0 < b1) gload(ga, b0);
            constexpr int T = 64 * S;
            const int nch = p.K >> 3;
            float* ssred = (float*)(stage_dyn + S * 8192);
#pragma unroll 1
            for (int sg = 0; sg < NG; ++sg) {
                float ss[MBX];
#pragma unroll
                for (int m = 0; m < MBX; ++m) {
                    float a = 0.f;
                    if (m < Mg(sg)) {
                        for (int c = t; c < nch; c += T) {
                            const bf16x8 v = *(const bf16x8*)(p.x + (int64_t)(MBX * sg + m) * p.ldx + c * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) a += (float)v[e] * (float)v[e];
                        }
                    }
                    ss[m] = wave_sum(a);
                }
                if (lane == 0) {
#pragma unroll
                    for (int m = 0; m < MBX; ++m) ssred[w * MBX + m] = ss[m];
                }
                __syncthreads();
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < S; ++q) a += ssred[q * MBX + xm];
                const float rs = rsqrtf(a * (1.f / (float)p.K) + p.eps);
#pragma unroll
                for (int s2 = 0; s2 < NG; ++s2)
                    if (s2 == sg) my_rstd[s2] = rs;
                __syncthreads();
            }
        }
    } else {
        if (b0 < b1) gload(ga, b0);
    }
    auto consumeG = [&](bf16x8(&cur)[8], bf16x8(&nxt)[8], int kb) {
        bf16x8 xr[NG][NB], gw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {   // ahead of the next stage's weight loads: loads return in order
            const int kk = min(kb + b, b1 - 1) << 6;
#pragma unroll
            for (int sg = 0; sg < NG; ++sg) xr[sg][b] = *(const bf16x8*)(xrow[sg] + kk);
            if constexpr (PRO == PRO_RMS) gw[b] = *(const bf16x8*)(p.normw + kk + xpc * 8);
        }
        if (kb + STEP < b1) gload(nxt, kb + STEP);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < JR; ++j) *(bf16x8*)(my + b * BLK + wr_off[j]) = cur[b * JR + j];
#pragma unroll
        for (int bh = 0; bh < NB; bh += HB) {   // HB blocks at a time through the strip (LDS operations of a wave execute in order)
#pragma unroll
            for (int sg = 0; sg < NG; ++sg)
#pragma unroll
                for (int bb = 0; bb < HB; ++bb) {
                    const int b = bh + bb;
                    bf16x8 h = xr[sg][b];
                    if constexpr (PRO == PRO_RMS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (bf16)((float)gw[b][e] * rbf((float)xr[sg][b][e] * my_rstd[sg]));   // cast BEFORE the weight multiply (:250-252)
                    }
                    *(bf16x8*)(hx + bb * (MC * 128) + sg * 1024 + hx_wr) = h;
                }
#pragma unroll
            for (int bb = 0; bb < HB; ++bb) {
                const int b = bh + bb;
                if (kb + b < b1) {   // wave-uniform
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const bf16x8 wf = *(const bf16x8*)(my + b * BLK + rd_row + ((((2 * st + hi)) ^ rd_swz) << 4));
                        const bf16x8 xf = *(const bf16x8*)(hx + bb * (MC * 128) + hx_rd_row + ((((2 * st + hi)) ^ hx_rd_swz) << 4));   // lanes beyond MC repeat rows: columns nobody reads
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
                    }
                }
            }
        }
    };
    if (b0 < b1) {
        for (int kb = b0; kb < b1; kb += 2 * STEP) {
            consumeG(ga, gb, kb);
            if (kb + STEP < b1) consumeG(gb, ga, kb + STEP);
        }
    }
    // ---- epilogue, once per group: the eight-sequence epilogue of gemv_chain_mfma_kernel with every barrier executed by every thread
    float* red = (float*)my;                      // [R][MC], head of this wave's own staging area (R x MC x 4 B <= 4 KiB)
    float* fin = (float*)&stage[0][4096];         // [R][MBX]
    const bool fin_t = t < R * MBX;
    const int r = fin_t ? t / MBX : 0, m = t % MBX;
    const int row = r < HR ? rA + r : rB + r - HR;
    if (l31 < MC) {
#pragma unroll
        for (int q = 0; q < RG / 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(8 * q + 4 * hi + e) * MC + l31] = acc[4 * q + e];
    }
    __syncthreads();
#pragma unroll 1
    for (int sg = 0; sg < NG; ++sg) {
        const int mg = MBX * sg + m;               // the sequence of this thread
        const bool valid = fin_t && m < Mg(sg);
        float tot = 0.f;
        if (fin_t) {
#pragma unroll
            for (int q = 0; q < S; ++q) tot += ((const float*)&stage[q][0])[r * MC + MBX * sg + m];   // K slices summed in slice order
        }
        float mine = 0.f;
        if (EPI == EPI_QKV) {
            mine = rbf(tot + (float)p.bias[row]);
            if (fin_t) fin[r * MBX + m] = mine;
        } else if (EPI == EPI_SWIGLU) {
            mine = rbf(tot);
            if (fin_t) fin[r * MBX + m] = mine;
        } else if (EPI == EPI_RESID) {
            const bf16 ov = (bf16)(rbf(tot) + (float)p.residual[(int64_t)min(mg, p.M - 1) * p.ld_res + row]);
            if (valid) p.out[(int64_t)mg * p.ld_out + row] = ov;
            if (fin_t && p.ss_out != nullptr) fin[r * MBX + m] = valid ? (float)ov * (float)ov : 0.f;
        } else {
            if (valid) p.out_f32[(int64_t)mg * p.ld_out + row] = rbf(tot);
        }
        __syncthreads();
        if (EPI == EPI_QKV) {
            if (valid) {
                const float other = fin[(r ^ HR) * MBX + m];
                const int start = *p.start;
                if (g < rot_groups) {
                    const int64_t ps = (int64_t)p.pos[mg * p.pos_stride] * p.D + row % p.D;
                    const float rot = r < HR ? -other : other;
                    const float o = rbf(rbf_strict(mine * (float)p.cos_t[ps]) + rbf_strict(rot * (float)p.sin_t[ps]));   // rbf_strict: contraction-proof (common.h)
                    if (row < nq) p.q_out[(int64_t)mg * p.ldq + row] = (bf16)o;
                    else p.Kc[(int64_t)mg * p.k_bs + (int64_t)start * nk + (row - nq)] = (bf16)o;
                } else {
                    p.Vt[(int64_t)mg * p.vt_bs + (int64_t)(row - nq - nk) * p.spad + start] = (bf16)mine;
                }
            }
        } else if (EPI == EPI_SWIGLU) {
            if (valid && r < HR) {
                const float u = fin[(r + HR) * MBX + m];
                p.out[(int64_t)mg * p.ld_out + row] = (bf16)(rbf(mine * sigmoid_f(mine)) * u);
            }
        } else if (EPI == EPI_RESID) {
            if (p.ss_out != nullptr && t < MBX) {
                float a = 0.f;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) a += fin[rr * MBX + t];
                p.ss_out[((int64_t)sg * MBX + t) * gridDim.x + g] = a;   // [group][row][part]
            }
        }
        __syncthreads();   // fin is rewritten by the next group
    }
}

#define AFK_CHAIN_SEQ_MAX (4 * AFK_CHAIN_BATCH_MAX)   // the norm-in-prologue / LDS-strip launches: up to four groups of eight sequences = the 32 columns of the MFMA
template <int EPI, int S, int RG, int PRO, int NG>
int launch_chain_ng(const ChainArgs& p, int rows, hipStream_t st) {
    constexpr int LDS = S * 8192 + S * 2 * NG * 1024;   // weight stages + one [2 blocks][8 NG rows][128 B] strip per wave
    static bool attr_set = false;
    if (LDS > 65536 && !attr_set) {
        hipFuncSetAttribute((const void*)gemv_chain_mfma_ng_kernel<EPI, S, RG, PRO, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemv_chain_mfma_ng_kernel<EPI, S, RG, PRO, NG>), dim3((unsigned)(rows / RG)), dim3(64 * S), LDS, st, p);
    return AFK_OK;
}

// AFK_CHAIN_S / AFK_CHAIN_R = "qkv,linear(K<=4096),linear(K>4096),gate_up,lm_head" (measurement knobs; 0 = default)
int chain_knob(int which, int dflt, bool is_r) {   // read per launch (a getenv; nothing inside a replayed graph): the tests switch forms in-process
    int v[5] = {0, 0, 0, 0, 0};
    if (const char* e = getenv(is_r ? "AFK_CHAIN_R" : "AFK_CHAIN_S")) sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    const int s = v[which];
    if (is_r) return (s == 4 || s == 8) ? s : dflt;
    return (s == 1 || s == 2 || s == 4 || s == 8) ? s : dflt;
}

template <int PRO, int EPI>
int launch_chain(const ChainArgs& p, int rows, int which, int S_dflt, int R_dflt, bool r4_ok, hipStream_t st) {
    const int S = chain_knob(which, S_dflt, false);
    int R_ = chain_knob(which, R_dflt, true);
    if (!r4_ok) R_ = 8;
    const int ngroups = rows / R_;
#define AFK_CHAIN(S_, R__)                                                                                                                          \
    {                                                                                                                                               \
        constexpr int G_ = S_ >= 4 ? 1 : 4 / S_;                                                                                                    \
        hipLaunchKernelGGL((gemv_chain_kernel<PRO, EPI, S_, R__>), dim3((unsigned)afk_cdiv(ngroups, G_)), dim3(64 * (S_ > 4 ? S_ : 4)), 0, st, p,   \
                           ngroups);                                                                                                                \
    }
#define AFK_CHAIN_R(S_)                 \
    case S_:                            \
        if (R_ == 8) AFK_CHAIN(S_, 8)   \
        else AFK_CHAIN(S_, 4)           \
        break;
    switch (S) {
        AFK_CHAIN_R(1)
        AFK_CHAIN_R(2)
        AFK_CHAIN_R(4)
        AFK_CHAIN_R(8)
    }
#undef AFK_CHAIN_R
#undef AFK_CHAIN
    return AFK_OK;
}

#define KIL_DEFAULT 1, 1, 1, 1, 1   // round 6: round-robin stages measured 3-9 % faster on the narrow Linears, level on the wide ones
template <int EPI>
int launch_chain_batched(const ChainArgs& p, int rows, int which, int S_dflt, hipStream_t st) {
    // matrix-pipe form when the shape allows (32- / 16-row groups, 64-element blocks); AFK_CHAIN_MFMA=0 / 1 forces the dot-product / the matrix-pipe form (A/B, tests)
    const char* e = getenv("AFK_CHAIN_MFMA");
    const bool shape_ok = rows % 32 == 0 && p.K % 64 == 0 && (EPI != EPI_QKV || ((p.D / 2) % 16 == 0 && (p.Hkv * p.D) % 32 == 0)) && (EPI != EPI_SWIGLU || (rows / 2) % 16 == 0);
    // measured on the AF3-7B decode step (ms per step, B = 2 / 4 / 8): dot-product form 3.32 / 3.73 / 4.75, matrix-pipe form 3.66 / 3.89 / 4.43 -> MFMA from five rows on (round 4; round 6 below: from four)
    const bool use_mfma = e ? e[0] == '1' : p.M >= 4;   // round 6 (per-Linear group shapes, fused norm): B = 4 step 3.73 ms on the dot-product form against 3.65 at B = 5 on the matrix pipe
    if (shape_ok && use_mfma) {
        ChainArgs q = p;
        {   // AFK_CHAIN_KIL = "a/b/c/d/e" per launch kind (qkv / o_proj / down / gate|up / lm_head), or one digit for all (measurement knob)
            int v[5] = {KIL_DEFAULT};
            if (const char* c = getenv("AFK_CHAIN_KIL")) {
                const int n = sscanf(c, "%d/%d/%d/%d/%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
                if (n == 1) v[1] = v[2] = v[3] = v[4] = v[0];
            }
            q.kil = v[which] != 0;
        }
        // 1 184 / 4 752 groups of 32 rows (gate|up, lm_head): 4 K slices.  The narrow Linears: qkv 32-row groups x 8 slices, o_proj and down 16-row groups x 8 slices
        // (224 blocks of 512 threads; round 5 ran down as 16 x 16 = one 1024-thread block per CU: 34 us against 30).
        // AFK_CHAIN_MFMA_NARROW = "rg,s" overrides the narrow form (measurement knob)
        // round 6, per launch at M = 8 (tools/bench_decode_chain_batched.py, us): qkv 32,8 10.6 / 16,8 16.1 · o_proj 32,8 9.6 / 16,8 8.4 / 16,16 8.2 · down 16,16 33.7 / 16,8 30.1 / 32,8 39.2
        int rg = which == 0 ? 32 : 16, sl = 8;
        if (const char* c = getenv("AFK_CHAIN_MFMA_NARROW")) {   // "rg,s" for every narrow Linear, or "rg,s/rg,s/rg,s" = qkv / o_proj (K <= 4096) / down
            int v[3][2] = {{0, 0}, {0, 0}, {0, 0}};
            const int n = sscanf(c, "%d,%d/%d,%d/%d,%d", &v[0][0], &v[0][1], &v[1][0], &v[1][1], &v[2][0], &v[2][1]);
            const int w3 = which <= 2 ? which : 0;
            if (n == 2) rg = v[0][0], sl = v[0][1];
            else if (n == 6 && v[w3][0] > 0) rg = v[w3][0], sl = v[w3][1];
        }
#define AFK_MFMA(S_, RG_)                                                                                                                              \
    do {                                                                                                                                                 \
        static bool attr_set = false;                                                                                                                    \
        if (S_ * 8192 > 65536 && !attr_set) {                                                                                                            \
            hipFuncSetAttribute((const void*)gemv_chain_mfma_kernel<EPI, S_, RG_>, hipFuncAttributeMaxDynamicSharedMemorySize, S_ * 8192);              \
            attr_set = true;                                                                                                                             \
        }                                                                                                                                                \
        hipLaunchKernelGGL((gemv_chain_mfma_kernel<EPI, S_, RG_>), dim3((unsigned)(rows / RG_)), dim3(64 * S_), S_ * 8192, st, q);                      \
    } while (0)
        if (rows / 32 >= 1024) AFK_MFMA(4, 32);
        else if (rg == 16 && sl == 16) AFK_MFMA(16, 16);
        else if (rg == 16) AFK_MFMA(8, 16);
        else AFK_MFMA(8, 32);
#undef AFK_MFMA
        return AFK_OK;
    }
    const int S = chain_knob(which, S_dflt, false);
    const int ngroups = rows / 8;
    const int MB = p.M <= 2 ? 2 : p.M <= 4 ? 4 : 8;
#define AFK_CHAINB(S_, MB_)                                                                                                                             \
    {                                                                                                                                                   \
        constexpr int G_ = S_ >= 4 ? 1 : 4 / S_;                                                                                                        \
        hipLaunchKernelGGL((gemv_chain_batched_kernel<EPI, S_, MB_>), dim3((unsigned)afk_cdiv(ngroups, G_)), dim3(64 * (S_ > 4 ? S_ : 4)), 0, st, p,    \
                           ngroups);                                                                                                                    \
    }
#define AFK_CHAINB_M(S_)                    \
    case S_:                                \
        if (MB == 2) AFK_CHAINB(S_, 2)      \
        else if (MB == 4) AFK_CHAINB(S_, 4) \
        else AFK_CHAINB(S_, 8)              \
        break;
    switch (S) {
        AFK_CHAINB_M(1)
        AFK_CHAINB_M(2)
        AFK_CHAINB_M(4)
        AFK_CHAINB_M(8)
    }
#undef AFK_CHAINB_M
#undef AFK_CHAINB
    return AFK_OK;
}

}  // namespace

#define ST ((hipStream_t)stream)

#ifdef AFK_PROBES
extern "C" int afk_probe_decode_chain_stamps(long long* device_buffer) {   // [waves][8] int64, or null to switch the stamps off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamps), &device_buffer, sizeof(device_buffer)) == hipSuccess ? AFK_OK : AFK_ERR_LAUNCH;
}
#endif

extern "C" int afk_decode_chain_qkv(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int K, const void* bias, const void* cos_t,
                                    const void* sin_t, const int* pos, void* q_out, void* kcache, void* vtcache, int spad, const int* start_dev, int Hq,
                                    int Hkv, int D, void* stream) {
    AFK_REQUIRE(x && norm_w && W && bias && cos_t && sin_t && pos && q_out && kcache && vtcache && start_dev, "afk_decode_chain_qkv: null pointer");
    AFK_REQUIRE(K > 0 && K % 8 == 0 && ldw % 8 == 0 && Hq > 0 && Hkv > 0 && D % 16 == 0 && spad > 0, "afk_decode_chain_qkv: unsupported shape (K %% 8 == 0, head_dim %% 16 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = (Hq + 2 * Hkv) * D; p.K = K;
    p.bias = (const bf16*)bias; p.cos_t = (const bf16*)cos_t; p.sin_t = (const bf16*)sin_t; p.pos = pos; p.start = start_dev; p.q_out = (bf16*)q_out;
    p.Kc = (bf16*)kcache; p.Vt = (bf16*)vtcache; p.spad = spad; p.Hq = Hq; p.Hkv = Hkv; p.D = D;
    launch_chain<PRO_RMS, EPI_QKV>(p, p.N, 0, 4, 8, true, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_qkv");
    return AFK_OK;
}

extern "C" int afk_decode_chain_linear_residual(const void* x, const void* W, int64_t ldw, int N, int K, const void* residual, void* out, void* stream) {
    AFK_REQUIRE(x && W && residual && out && N > 0 && N % 8 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "afk_decode_chain_linear_residual: bad arguments (N %% 8 == 0, K %% 8 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.residual = (const bf16*)residual; p.out = (bf16*)out; p.D = 2;
    launch_chain<PRO_PLAIN, EPI_RESID>(p, N, K > 4096 ? 2 : 1, K > 4096 ? 8 : 4, 8, true, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual");
    return AFK_OK;
}

extern "C" int afk_decode_chain_gate_up(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int I, int K, void* act_out, void* stream) {
    AFK_REQUIRE(x && norm_w && W && act_out && I > 0 && I % 4 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "afk_decode_chain_gate_up: unsupported shape (I %% 4 == 0, K %% 8 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = 2 * I; p.K = K; p.out = (bf16*)act_out; p.D = 2;
    launch_chain<PRO_RMS, EPI_SWIGLU>(p, 2 * I, 3, I / 4 >= 2048 ? 1 : 4, 8, true, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_gate_up");
    return AFK_OK;
}

// final RMSNorm (Qwen2Model.norm) + lm_head on one row: logits[N] fp32 = float(bf16(W h)) - the values `lm_head(norm(x)).float()` holds (logits may be null);
// part_val / part_idx [N / 8] (or null): largest logit and its row per group of eight rows, for afk_decode_select_greedy
extern "C" int afk_decode_chain_lm_head(const void* x, const void* norm_w, float eps, const void* W, int64_t ldw, int N, int K, float* logits, float* part_val,
                                        int* part_idx, void* stream) {
    AFK_REQUIRE(x && norm_w && W && N > 0 && N % 8 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "afk_decode_chain_lm_head: unsupported shape (N %% 8 == 0, K %% 8 == 0)");
    AFK_REQUIRE((logits || part_val) && (!part_val == !part_idx), "afk_decode_chain_lm_head: logits and / or both partial-argmax buffers");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.out_f32 = logits; p.D = 2;
    p.part_val = part_val; p.part_idx = part_idx;
    launch_chain<PRO_RMS, EPI_LOGITS>(p, N, 4, N / 8 >= 2048 ? 1 : 4, 8, false, ST);   // always eight rows per group: part_* are indexed by it
    AFK_LAUNCH_CHECK("afk_decode_chain_lm_head");
    return AFK_OK;
}

extern "C" int afk_decode_select_greedy(const float* part_val, const int* part_idx, int nparts, int64_t* next_token, int64_t* tokens_out, int tok_off, int* state,
                                        const void* emb, int64_t ld_emb, int H, void* x_out, void* stream) {
    AFK_REQUIRE(part_val && part_idx && nparts > 0 && next_token && state && emb && x_out && H > 0 && H % 4 == 0 && ld_emb % 4 == 0,
                "afk_decode_select_greedy: bad arguments (H %% 4 == 0)");
    hipLaunchKernelGGL(decode_select_greedy_kernel, dim3(1), dim3(1024), 0, ST, part_val, part_idx, nparts, (long long*)next_token, (long long*)tokens_out, tok_off,
                       state, (const bf16*)emb, ld_emb, H, (bf16*)x_out);
    AFK_LAUNCH_CHECK("afk_decode_select_greedy");
    return AFK_OK;
}

// ---------------------------------------------------------------- 2 .. 8 sequences per step (inputs already normalised where the Linear follows a norm)
extern "C" int afk_decode_chain_qkv_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int K, const void* bias, const void* cos_t,
                                            const void* sin_t, const int* pos, void* q_out, int64_t ldq, void* kcache, int64_t k_bs, void* vtcache, int64_t vt_bs,
                                            int spad, const int* start_dev, int Hq, int Hkv, int D, void* stream) {
    AFK_REQUIRE(h && W && bias && cos_t && sin_t && pos && q_out && kcache && vtcache && start_dev, "afk_decode_chain_qkv_batched: null pointer");
    AFK_REQUIRE(M >= 1 && M <= AFK_CHAIN_SEQ_MAX && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldh % 8 == 0 && Hq > 0 && Hkv > 0 && D % 16 == 0 && spad > 0,
                "afk_decode_chain_qkv_batched: unsupported shape (1 <= M <= 32, K %% 8 == 0, head_dim %% 16 == 0)");
    AFK_REQUIRE(M <= AFK_CHAIN_BATCH_MAX || (K % 64 == 0 && (D / 2) % 16 == 0 && (Hkv * D) % 32 == 0 && ((Hq + 2 * Hkv) * D) % 32 == 0),
                "afk_decode_chain_qkv_batched: more than 8 sequences need K %% 64 == 0 and head_dim %% 32 == 0 (groups of eight on the matrix pipe)");
    ChainArgs p = {};
    p.x = (const bf16*)h; p.ldx = ldh; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = (Hq + 2 * Hkv) * D; p.K = K;
    p.bias = (const bf16*)bias; p.cos_t = (const bf16*)cos_t; p.sin_t = (const bf16*)sin_t; p.pos = pos; p.pos_stride = 1; p.start = start_dev;
    p.q_out = (bf16*)q_out; p.ldq = ldq; p.Kc = (bf16*)kcache; p.k_bs = k_bs; p.Vt = (bf16*)vtcache; p.vt_bs = vt_bs; p.spad = spad; p.Hq = Hq; p.Hkv = Hkv; p.D = D;
    if (M > AFK_CHAIN_BATCH_MAX) {   // 9 .. 32 sequences: groups of eight as columns of one MFMA, input rows through the LDS strip (gemv_chain_mfma_ng_kernel)
        p.kil = 1;
        if (M > 2 * AFK_CHAIN_BATCH_MAX) launch_chain_ng<EPI_QKV, 8, 32, PRO_PLAIN_LDS, 4>(p, p.N, ST);
        else launch_chain_ng<EPI_QKV, 8, 32, PRO_PLAIN_LDS, 2>(p, p.N, ST);
        AFK_LAUNCH_CHECK("afk_decode_chain_qkv_batched");
        return AFK_OK;
    }
    launch_chain_batched<EPI_QKV>(p, p.N, 0, 4, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_qkv_batched");
    return AFK_OK;
}

extern "C" int afk_decode_chain_linear_residual_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual,
                                                        int64_t ld_res, void* out, int64_t ld_out, void* stream) {
    AFK_REQUIRE(x && W && residual && out && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && N > 0 && N % 8 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0,
                "afk_decode_chain_linear_residual_batched: bad arguments (1 <= M <= 32, N %% 8 == 0, K %% 8 == 0)");
    AFK_REQUIRE(M <= AFK_CHAIN_BATCH_MAX || (N % 32 == 0 && K % 64 == 0), "afk_decode_chain_linear_residual_batched: more than 8 sequences need N %% 32 == 0 and K %% 64 == 0");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.residual = (const bf16*)residual; p.ld_res = ld_res;
    p.out = (bf16*)out; p.ld_out = ld_out; p.D = 2;
    if (M > AFK_CHAIN_BATCH_MAX) {
        p.kil = 1;
        if (M > 2 * AFK_CHAIN_BATCH_MAX) launch_chain_ng<EPI_RESID, 8, 16, PRO_PLAIN_LDS, 4>(p, N, ST);
        else launch_chain_ng<EPI_RESID, 8, 16, PRO_PLAIN_LDS, 2>(p, N, ST);
        AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual_batched");
        return AFK_OK;
    }
    launch_chain_batched<EPI_RESID>(p, N, K > 4096 ? 2 : 1, K > 4096 ? 8 : 4, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual_batched");
    return AFK_OK;
}

// ---------------------------------------------------------------- 1 .. 8 sequences, RMSNorm in the Linear's own prologue (matrix-pipe form only, round 6)
namespace {
template <int EPI, int S>
int launch_chain_norm(ChainArgs p, int rows, hipStream_t st) {
    constexpr int NBX = 2;   // 32-row groups: two 64-element blocks per stage
    constexpr int LDS = S * 8192 + S * NBX * 1024;
    p.kil = 1;
    if (p.M > 2 * AFK_CHAIN_BATCH_MAX) return launch_chain_ng<EPI, S, 32, PRO_RMS, 4>(p, rows, st);   // 17 .. 32 sequences: four groups of eight
    if (p.M > AFK_CHAIN_BATCH_MAX) return launch_chain_ng<EPI, S, 32, PRO_RMS, 2>(p, rows, st);       // 9 .. 16: two
    static bool attr_set = false;
    if (LDS > 65536 && !attr_set) {
        hipFuncSetAttribute((const void*)gemv_chain_mfma_kernel<EPI, S, 32, PRO_RMS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    static const int ssf = [] { const char* e = getenv("AFK_CHAIN_SS_FIRST"); return e && e[0] == '0' ? 0 : 1; }();
    p.kil = 1;
    p.ss_first = ssf;
    hipLaunchKernelGGL((gemv_chain_mfma_kernel<EPI, S, 32, PRO_RMS>), dim3((unsigned)(rows / 32)), dim3(64 * S), LDS, st, p);
    return AFK_OK;
}
}  // namespace

extern "C" int afk_decode_chain_qkv_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int K, const void* bias,
                                                 const void* cos_t, const void* sin_t, const int* pos, void* q_out, int64_t ldq, void* kcache, int64_t k_bs,
                                                 void* vtcache, int64_t vt_bs, int spad, const int* start_dev, int Hq, int Hkv, int D, const float* ss_part,
                                                 int ss_nparts, void* stream) {
    AFK_REQUIRE(x && norm_w && W && bias && cos_t && sin_t && pos && q_out && kcache && vtcache && start_dev, "afk_decode_chain_qkv_norm_batched: null pointer");
    AFK_REQUIRE(!ss_part || (ss_nparts > 0 && ss_nparts <= 256 && ss_nparts % 4 == 0 && (uintptr_t)ss_part % 16 == 0),
                "afk_decode_chain_qkv_norm_batched: 4 .. 256 partial sums per row, a multiple of 4, 16-byte aligned");
    AFK_REQUIRE(M >= 1 && M <= AFK_CHAIN_SEQ_MAX && K > 0 && K % 64 == 0 && ldw % 8 == 0 && ldx % 8 == 0 && Hq > 0 && Hkv > 0 && (D / 2) % 16 == 0 && (Hkv * D) % 32 == 0 &&
                    ((Hq + 2 * Hkv) * D) % 32 == 0 && spad > 0,
                "afk_decode_chain_qkv_norm_batched: unsupported shape (1 <= M <= 32, K %% 64 == 0, head_dim %% 32 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = (Hq + 2 * Hkv) * D; p.K = K;
    p.bias = (const bf16*)bias; p.cos_t = (const bf16*)cos_t; p.sin_t = (const bf16*)sin_t; p.pos = pos; p.pos_stride = 1; p.start = start_dev;
    p.q_out = (bf16*)q_out; p.ldq = ldq; p.Kc = (bf16*)kcache; p.k_bs = k_bs; p.Vt = (bf16*)vtcache; p.vt_bs = vt_bs; p.spad = spad; p.Hq = Hq; p.Hkv = Hkv; p.D = D;
    p.ss_in = ss_part; p.ss_nparts = ss_part ? ss_nparts : 0;
    launch_chain_norm<EPI_QKV, 8>(p, p.N, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_qkv_norm_batched");
    return AFK_OK;
}

extern "C" int afk_decode_chain_gate_up_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int I, int K,
                                                     void* act_out, int64_t ld_act, const float* ss_part, int ss_nparts, void* stream) {
    AFK_REQUIRE(!ss_part || (ss_nparts > 0 && ss_nparts <= 256 && ss_nparts % 4 == 0 && (uintptr_t)ss_part % 16 == 0),
                "afk_decode_chain_gate_up_norm_batched: 4 .. 256 partial sums per row, a multiple of 4, 16-byte aligned");
    AFK_REQUIRE(x && norm_w && W && act_out && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && I > 0 && I % 16 == 0 && K > 0 && K % 64 == 0 && ldw % 8 == 0 && ldx % 8 == 0,
                "afk_decode_chain_gate_up_norm_batched: unsupported shape (1 <= M <= 32, I %% 16 == 0, K %% 64 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = 2 * I; p.K = K;
    p.out = (bf16*)act_out; p.ld_out = ld_act; p.D = 2; p.ss_in = ss_part; p.ss_nparts = ss_part ? ss_nparts : 0;
    launch_chain_norm<EPI_SWIGLU, 4>(p, 2 * I, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_gate_up_norm_batched");
    return AFK_OK;
}

extern "C" int afk_decode_chain_lm_head_norm_batched(const void* x, int64_t ldx, int M, const void* norm_w, float eps, const void* W, int64_t ldw, int N, int K,
                                                     float* logits, int64_t ld_logits, const float* ss_part, int ss_nparts, void* stream) {
    AFK_REQUIRE(!ss_part || (ss_nparts > 0 && ss_nparts <= 256 && ss_nparts % 4 == 0 && (uintptr_t)ss_part % 16 == 0),
                "afk_decode_chain_lm_head_norm_batched: 4 .. 256 partial sums per row, a multiple of 4, 16-byte aligned");
    AFK_REQUIRE(x && norm_w && W && logits && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && N > 0 && N % 32 == 0 && K > 0 && K % 64 == 0 && ldw % 8 == 0 && ldx % 8 == 0,
                "afk_decode_chain_lm_head_norm_batched: unsupported shape (1 <= M <= 32, N %% 32 == 0, K %% 64 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.normw = (const bf16*)norm_w; p.eps = eps; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K;
    p.out_f32 = logits; p.ld_out = ld_logits; p.D = 2; p.ss_in = ss_part; p.ss_nparts = ss_part ? ss_nparts : 0;
    launch_chain_norm<EPI_LOGITS, 4>(p, N, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_lm_head_norm_batched");
    return AFK_OK;
}

// Linear + residual, 1 .. 8 sequences, on the matrix-pipe form (16-row groups x 8 K slices), that also leaves ss_part[N / 16][8]: every block's share of sum_n out[m][n]^2 -
// the statistic of the RMSNorm that FOLLOWS, folded by the next Linear's prologue (afk_decode_chain_*_norm_batched with ss_part): no pass over the rows, no hand-over.
extern "C" int afk_decode_chain_linear_residual_ss_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual, int64_t ld_res,
                                                           void* out, int64_t ld_out, float* ss_part, void* stream) {
    AFK_REQUIRE(x && W && residual && out && ss_part && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && N > 0 && N % 64 == 0 && K > 0 && K % 64 == 0 && ldw % 8 == 0 && ldx % 8 == 0,
                "afk_decode_chain_linear_residual_ss_batched: bad arguments (1 <= M <= 32, N %% 64 == 0, K %% 64 == 0)");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.residual = (const bf16*)residual; p.ld_res = ld_res;
    p.out = (bf16*)out; p.ld_out = ld_out; p.D = 2; p.ss_out = ss_part; p.kil = 1;
    if (M > AFK_CHAIN_BATCH_MAX) {   // 9 .. 32 sequences: groups of eight in one pass over the weights; ss_part is [groups][N / 16][8]
        if (M > 2 * AFK_CHAIN_BATCH_MAX) launch_chain_ng<EPI_RESID, 8, 16, PRO_PLAIN_LDS, 4>(p, N, ST);
        else launch_chain_ng<EPI_RESID, 8, 16, PRO_PLAIN_LDS, 2>(p, N, ST);
        AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual_ss_batched");
        return AFK_OK;
    }
    // input rows through the LDS strip (cooperative 16-byte loads, fragments read back from LDS) instead of four lane-masked fragment loads per block from the L2:
    // down 30.9 -> 26.5 us at M = 8, o_proj level (AFK_CHAIN_XLDS=0: A/B)
    static const bool xlds = [] { const char* e = getenv("AFK_CHAIN_XLDS"); return !(e && e[0] == '0'); }();
    if (xlds) {
        constexpr int LDS = 8 * 8192 + 8 * 4 * 1024;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)gemv_chain_mfma_kernel<EPI_RESID, 8, 16, PRO_PLAIN_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            attr_set = true;
        }
        hipLaunchKernelGGL((gemv_chain_mfma_kernel<EPI_RESID, 8, 16, PRO_PLAIN_LDS>), dim3((unsigned)(N / 16)), dim3(512), LDS, ST, p);
    } else
    hipLaunchKernelGGL((gemv_chain_mfma_kernel<EPI_RESID, 8, 16>), dim3((unsigned)(N / 16)), dim3(512), 8 * 8192, ST, p);
    AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual_ss_batched");
    return AFK_OK;
}

// Linear + residual + the RMSNorm that follows, 1 .. 8 sequences, always on the matrix-pipe form (16-row groups x 8 K slices): out = bf16(W x) + residual,
// h_out = norm_w * bf16(out * rsqrt(mean(out^2) + eps)) written by the last block of the launch to arrive (EPI_RESID_NORM above).  counter: one zero-initialised int32.
extern "C" int afk_decode_chain_linear_residual_norm_batched(const void* x, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, const void* residual,
                                                             int64_t ld_res, void* out, int64_t ld_out, const void* norm_w, float eps, void* h_out, int64_t ld_h,
                                                             int* counter, void* stream) {
    AFK_REQUIRE(x && W && residual && out && norm_w && h_out && counter, "afk_decode_chain_linear_residual_norm_batched: null pointer");
    AFK_REQUIRE(M >= 1 && M <= AFK_CHAIN_BATCH_MAX && N > 0 && N % 32 == 0 && N / 2 <= 4 * 512 && K > 0 && K % 64 == 0 && ldw % 8 == 0 && ldx % 8 == 0,
                "afk_decode_chain_linear_residual_norm_batched: unsupported shape (1 <= M <= 8, N %% 32 == 0, N <= 4096, K %% 64 == 0)");
    AFK_REQUIRE(ld_out % 2 == 0 && ld_h % 2 == 0 && ((uintptr_t)out % 4 == 0) && ((uintptr_t)h_out % 4 == 0) && ((uintptr_t)norm_w % 4 == 0),
                "afk_decode_chain_linear_residual_norm_batched: out / h_out / norm_w must be 4-byte aligned with even leading dimensions");
    ChainArgs p = {};
    p.x = (const bf16*)x; p.ldx = ldx; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.residual = (const bf16*)residual; p.ld_res = ld_res;
    p.out = (bf16*)out; p.ld_out = ld_out; p.D = 2; p.n2_w = (const bf16*)norm_w; p.n2_out = (bf16*)h_out; p.ld_n2 = ld_h; p.n2_eps = eps; p.n2_counter = counter;
    p.kil = 1;
    hipLaunchKernelGGL((gemv_chain_mfma_kernel<EPI_RESID_NORM, 8, 16>), dim3((unsigned)(N / 16)), dim3(512), 8 * 8192, ST, p);
    AFK_LAUNCH_CHECK("afk_decode_chain_linear_residual_norm_batched");
    return AFK_OK;
}

extern "C" int afk_decode_chain_gate_up_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int I, int K, void* act_out, int64_t ld_act,
                                                void* stream) {
    AFK_REQUIRE(h && W && act_out && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && I > 0 && I % 4 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldh % 8 == 0,
                "afk_decode_chain_gate_up_batched: unsupported shape (1 <= M <= 32, I %% 4 == 0, K %% 8 == 0)");
    AFK_REQUIRE(M <= AFK_CHAIN_BATCH_MAX || (I % 16 == 0 && K % 64 == 0), "afk_decode_chain_gate_up_batched: more than 8 sequences need I %% 16 == 0 and K %% 64 == 0");
    ChainArgs p = {};
    p.x = (const bf16*)h; p.ldx = ldh; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = 2 * I; p.K = K; p.out = (bf16*)act_out; p.ld_out = ld_act; p.D = 2;
    if (M > AFK_CHAIN_BATCH_MAX) {
        p.kil = 1;
        if (M > 2 * AFK_CHAIN_BATCH_MAX) launch_chain_ng<EPI_SWIGLU, 4, 32, PRO_PLAIN_LDS, 4>(p, 2 * I, ST);
        else launch_chain_ng<EPI_SWIGLU, 4, 32, PRO_PLAIN_LDS, 2>(p, 2 * I, ST);
        AFK_LAUNCH_CHECK("afk_decode_chain_gate_up_batched");
        return AFK_OK;
    }
    launch_chain_batched<EPI_SWIGLU>(p, 2 * I, 3, I / 4 >= 2048 ? 1 : 4, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_gate_up_batched");
    return AFK_OK;
}

extern "C" int afk_decode_chain_lm_head_batched(const void* h, int64_t ldh, int M, const void* W, int64_t ldw, int N, int K, float* logits, int64_t ld_logits,
                                                void* stream) {
    AFK_REQUIRE(h && W && logits && M >= 1 && M <= AFK_CHAIN_SEQ_MAX && N > 0 && N % 8 == 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldh % 8 == 0,
                "afk_decode_chain_lm_head_batched: unsupported shape (1 <= M <= 32, N %% 8 == 0, K %% 8 == 0)");
    AFK_REQUIRE(M <= AFK_CHAIN_BATCH_MAX || (N % 32 == 0 && K % 64 == 0), "afk_decode_chain_lm_head_batched: more than 8 sequences need N %% 32 == 0 and K %% 64 == 0");
    ChainArgs p = {};
    p.x = (const bf16*)h; p.ldx = ldh; p.M = M; p.W = (const bf16*)W; p.ldw = ldw; p.N = N; p.K = K; p.out_f32 = logits; p.ld_out = ld_logits; p.D = 2;
    if (M > AFK_CHAIN_BATCH_MAX) {
        p.kil = 1;
        if (M > 2 * AFK_CHAIN_BATCH_MAX) launch_chain_ng<EPI_LOGITS, 4, 32, PRO_PLAIN_LDS, 4>(p, N, ST);
        else launch_chain_ng<EPI_LOGITS, 4, 32, PRO_PLAIN_LDS, 2>(p, N, ST);
        AFK_LAUNCH_CHECK("afk_decode_chain_lm_head_batched");
        return AFK_OK;
    }
    launch_chain_batched<EPI_LOGITS>(p, N, 4, N / 8 >= 2048 ? 1 : 4, ST);
    AFK_LAUNCH_CHECK("afk_decode_chain_lm_head_batched");
    return AFK_OK;
}
