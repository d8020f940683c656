// Flash attention forward + backward on MFMA for gfx950 (bf16 in/out, fp32 softmax and accumulation).
//
// Serves both attention shapes of the AF3 training path:
//   encoder  AudioFlamingo3Attention.forward, modeling_audioflamingo3.py:117-189 -> sdpa_attention_forward
//            (integrations/sdpa_attention.py:79-166): bidirectional, 20 heads x 64, optional key-padding length
//   decoder  Qwen2Attention.forward, modeling_qwen2.py:195-234: causal, GQA 28:4, head 128, softmax in fp32 (:167)
//
// v1 structure (correctness-first, no LDS): a 256-thread block is 4 independent waves; each wave owns 32 query
// rows (forward, dQ) or 32 key rows (dK/dV) and walks the other sequence in tiles of 32, feeding
// v_mfma_f32_32x32x16_bf16 straight from global/L2.  Score tiles are computed TRANSPOSED where that makes the
// softmax row lane-local (forward, dQ:  S^T = K.Q^T, lane = one query, 16 keys in registers) and un-transposed
// where the reduction runs over queries (dK/dV).  In both cases the probabilities come out of the MFMA already
// in the register layout of the NEXT MFMA's k-operand (the k index inside one MFMA may be permuted freely as
// long as both operands agree), so P never goes through LDS; the other operand of that second contraction must
// then be contiguous along the reduction index, which is why the host passes transposed copies
// (Vt, Kt, Qt, dOt = [B, H, D, Spad], zero padded) made by afk_transpose_bf16.
// The backward is the two-kernel FlashAttention-2 form (dK/dV sweep + dQ sweep, no atomics, deterministic).
#include "common.h"
#include "../../include/afk.h"

namespace {

struct AttnArgs {
    // row-major tensors addressed as base + b*bs + h*hs + s*rs + d   (element strides)
    const bf16* Q; int64_t q_bs, q_hs, q_rs;
    const bf16* K; int64_t k_bs, k_hs, k_rs;
    const bf16* V; int64_t v_bs, v_hs, v_rs;
    bf16* O; int64_t o_bs, o_hs, o_rs;         // forward out / backward: O (for nothing), see dO
    const bf16* dO; int64_t do_bs, do_hs, do_rs;
    bf16* dQ; int64_t dq_bs, dq_hs, dq_rs;
    bf16* dK; int64_t dk_bs, dk_hs, dk_rs;
    bf16* dV; int64_t dv_bs, dv_hs, dv_rs;
    // transposed copies [B, H, D, Spad]
    const bf16* Vt; const bf16* Kt; const bf16* Qt; const bf16* dOt;
    float* LSE;          // [B, Hq, S]
    const float* delta;  // [B, Hq, S]
    const int* kv_len;   // [B] or null
    const int* krange;   // [B, S, 2] = per-query [key_begin, key_end) or null (cross-attention segment masks)
    int B, Hq, Hkv, S, Spad;   // S / Spad: QUERY length (and pitch of Qt / dOt / LSE / delta)
    int Sk, Skpad;             // KEY length (and pitch of Kt / Vt); == S, Spad for self-attention
    float scale;
    int causal;
};

constexpr float NEG_INF = -INFINITY;

__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ bf16x8 ld4x2(const bf16* p0, const bf16* p1) {
    const bf16x4 a = *(const bf16x4*)p0, b = *(const bf16x4*)p1;
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// row index carried by accumulator register r in lane-half hi of a 32x32 MFMA result
#define ROW_OF(r, hi) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hi))

// ------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
    constexpr int KS = D / 16, DT = D / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // grid = (heads, batch, query blocks, last block first): workgroups go to the XCDs round-robin in linear order, so the fastest grid
    // dimension must not be the one the (causal) work per block depends on - see attention_lds.hip
    const int b = blockIdx.y, h = blockIdx.x, hk = h / (p.Hq / p.Hkv);
    const int q0 = ((int)gridDim.z - 1 - (int)blockIdx.z) * 128 + wave * 32;
    if (q0 >= p.S) return;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    int kb = 0, ke = p.Sk;
    if (p.krange) {
        kb = p.krange[((int64_t)b * p.S + qc) * 2];
        ke = p.krange[((int64_t)b * p.S + qc) * 2 + 1];
    }

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = ld8(Qp + ks * 16);

    const bf16* Kb = p.K + b * p.k_bs + hk * p.k_hs + hi * 8;
    const bf16* Vtb = p.Vt + ((int64_t)(b * p.Hkv + hk) * D + l31) * p.Skpad + 4 * hi;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
    float m = NEG_INF, l = 0.f;

    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;
    const int ntiles = (kv_end + 31) >> 5;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        const int krow = min(key0 + l31, p.Sk - 1);
        const bf16* Kp = Kb + (int64_t)krow * p.k_rs;
        f32x16 st = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st = MFMA(ld8(Kp + ks * 16), qf[ks], st);
        float s[16];
        float mx = NEG_INF;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + ROW_OF(r, hi);
            const bool dead = (key >= kv_len) || (p.causal && key > q) || (key < kb) || (key >= ke);
            s[r] = dead ? NEG_INF : st[r] * p.scale;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __expf(m - m_use);  // m=-inf -> 0
        float rs = 0.f;
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = __expf(s[r] - m_use);
            rs += pr[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        bf16x8 pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[s2][e] = (bf16)pr[8 * s2 + e];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16* vp = Vtb + (int64_t)dt * 32 * p.Skpad + key0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) oacc[dt] = MFMA(ld4x2(vp + 16 * s2, vp + 16 * s2 + 8), pb[s2], oacc[dt]);
        }
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
        if (hi == 0 && p.LSE) p.LSE[((int64_t)b * p.Hq + h) * p.S + q] = (l > 0.f) ? m + __logf(l) : NEG_INF;
    }
}

// ------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ O, int64_t o_bs, int64_t o_hs, int64_t o_rs,
                                                         const bf16* __restrict__ dO, int64_t do_bs, int64_t do_hs,
                                                         int64_t do_rs, float* __restrict__ delta, int B, int H, int S) {
    constexpr int LPR = D / 8;  // lanes per (b,h,s) row
    constexpr int IPB = 256 / LPR;
    const int64_t total = (int64_t)B * H * S;
    const int sub = threadIdx.x % LPR;
    for (int64_t base = (int64_t)blockIdx.x * IPB; base < total; base += (int64_t)gridDim.x * IPB) {
        const int64_t i = base + threadIdx.x / LPR;
        float acc = 0.f;
        const bool ok = i < total;
        if (ok) {
            const int s = (int)(i % S);
            const int64_t t = i / S;
            const int h = (int)(t % H), b = (int)(t / H);
            const bf16x8 o = ld8(O + b * o_bs + h * o_hs + (int64_t)s * o_rs + sub * 8);
            const bf16x8 d = ld8(dO + b * do_bs + h * do_hs + (int64_t)s * do_rs + sub * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)o[e] * (float)d[e];
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (ok && sub == 0) delta[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// wave = 32 keys of one kv head; sweeps every query head of the GQA group and every query tile.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnArgs p) {
    constexpr int KS = D / 16, DT = D / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.y, hk = blockIdx.x, group = p.Hq / p.Hkv;  // grid = (kv heads, batch, key blocks)
    const int key0 = blockIdx.z * 128 + wave * 32;
    if (key0 >= p.Sk) return;
    const int key = key0 + l31;
    const int keyc = min(key, p.Sk - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;

    const bf16* Kp = p.K + b * p.k_bs + hk * p.k_hs + (int64_t)keyc * p.k_rs + hi * 8;
    const bf16* Vp = p.V + b * p.v_bs + hk * p.v_hs + (int64_t)keyc * p.v_rs + hi * 8;

    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dkacc[dt] = zero16();
        dvacc[dt] = zero16();
    }
    const bool key_dead = key >= kv_len;
    const int qt_begin = p.causal ? (key0 >> 5) : 0;
    const int qt_end = (p.S + 31) >> 5;
    for (int g = 0; g < group; ++g) {
        const int h = hk * group + g;
        const bf16* Qb = p.Q + b * p.q_bs + h * p.q_hs + hi * 8;
        const bf16* dOb = p.dO + b * p.do_bs + h * p.do_hs + hi * 8;
        const bf16* Qtb = p.Qt + ((int64_t)(b * p.Hq + h) * D + l31) * p.Spad + 4 * hi;
        const bf16* dOtb = p.dOt + ((int64_t)(b * p.Hq + h) * D + l31) * p.Spad + 4 * hi;
        const float* lse = p.LSE + ((int64_t)b * p.Hq + h) * p.S;
        const float* dlt = p.delta + ((int64_t)b * p.Hq + h) * p.S;
        for (int qt = qt_begin; qt < qt_end; ++qt) {
            const int qt0 = qt * 32;
            const int qrow = min(qt0 + l31, p.S - 1);
            const bf16* Qp = Qb + (int64_t)qrow * p.q_rs;
            const bf16* dOp = dOb + (int64_t)qrow * p.do_rs;
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st = MFMA(ld8(Qp + ks * 16), ld8(Kp + ks * 16), st);
                dp = MFMA(ld8(dOp + ks * 16), ld8(Vp + ks * 16), dp);
            }
            bf16x8 pb[2], dsb[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qt0 + ROW_OF(r, hi);
                const int qi = min(qq, p.S - 1);
                bool dead = key_dead || (qq >= p.S) || (p.causal && key > qq);
                if (p.krange) dead = dead || key < p.krange[((int64_t)b * p.S + qi) * 2] || key >= p.krange[((int64_t)b * p.S + qi) * 2 + 1];
                const float pv = dead ? 0.f : __expf(st[r] * p.scale - lse[qi]);
                const float dsv = pv * (dp[r] - dlt[qi]) * p.scale;
                pb[r >> 3][r & 7] = (bf16)pv;
                dsb[r >> 3][r & 7] = (bf16)dsv;
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16* dop = dOtb + (int64_t)dt * 32 * p.Spad + qt0;
                const bf16* qp = Qtb + (int64_t)dt * 32 * p.Spad + qt0;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dvacc[dt] = MFMA(ld4x2(dop + 16 * s2, dop + 16 * s2 + 8), pb[s2], dvacc[dt]);
                    dkacc[dt] = MFMA(ld4x2(qp + 16 * s2, qp + 16 * s2 + 8), dsb[s2], dkacc[dt]);
                }
            }
        }
    }
    if (key < p.Sk) {
        bf16* dKp = p.dK + b * p.dk_bs + hk * p.dk_hs + (int64_t)key * p.dk_rs;
        bf16* dVp = p.dV + b * p.dv_bs + hk * p.dv_hs + (int64_t)key * p.dv_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 ok, ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ok[e] = (bf16)dkacc[dt][4 * qd + e];
                    ov[e] = (bf16)dvacc[dt][4 * qd + e];
                }
                *(bf16x4*)(dKp + dt * 32 + 8 * qd + 4 * hi) = ok;
                *(bf16x4*)(dVp + dt * 32 + 8 * qd + 4 * hi) = ov;
            }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
    constexpr int KS = D / 16, DT = D / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // grid = (heads, batch, query blocks, last block first): workgroups go to the XCDs round-robin in linear order, so the fastest grid
    // dimension must not be the one the (causal) work per block depends on - see attention_lds.hip
    const int b = blockIdx.y, h = blockIdx.x, hk = h / (p.Hq / p.Hkv);
    const int q0 = ((int)gridDim.z - 1 - (int)blockIdx.z) * 128 + wave * 32;
    if (q0 >= p.S) return;
    const int q = q0 + l31;
    const int qc = min(q, p.S - 1);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    int kb = 0, ke = p.Sk;
    if (p.krange) {
        kb = p.krange[((int64_t)b * p.S + qc) * 2];
        ke = p.krange[((int64_t)b * p.S + qc) * 2 + 1];
    }

    const bf16* Qp = p.Q + b * p.q_bs + h * p.q_hs + (int64_t)qc * p.q_rs + hi * 8;
    const bf16* dOp = p.dO + b * p.do_bs + h * p.do_hs + (int64_t)qc * p.do_rs + hi * 8;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = ld8(Qp + ks * 16);
        dof[ks] = ld8(dOp + ks * 16);
    }
    const float lse = p.LSE[((int64_t)b * p.Hq + h) * p.S + qc];
    const float dlt = p.delta[((int64_t)b * p.Hq + h) * p.S + qc];
    const bf16* Kb = p.K + b * p.k_bs + hk * p.k_hs + hi * 8;
    const bf16* Vb = p.V + b * p.v_bs + hk * p.v_hs + hi * 8;
    const bf16* Ktb = p.Kt + ((int64_t)(b * p.Hkv + hk) * D + l31) * p.Skpad + 4 * hi;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = zero16();

    const int kv_end = p.causal ? min(kv_len, q0 + 32) : kv_len;
    const int ntiles = (kv_end + 31) >> 5;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        const int krow = min(key0 + l31, p.Sk - 1);
        const bf16* Kp = Kb + (int64_t)krow * p.k_rs;
        const bf16* Vp = Vb + (int64_t)krow * p.v_rs;
        f32x16 st = zero16(), dp = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st = MFMA(ld8(Kp + ks * 16), qf[ks], st);
            dp = MFMA(ld8(Vp + ks * 16), dof[ks], dp);
        }
        bf16x8 dsb[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + ROW_OF(r, hi);
            const bool dead = (key >= kv_len) || (p.causal && key > q) || (key < kb) || (key >= ke);
            const float pv = dead ? 0.f : __expf(st[r] * p.scale - lse);
            dsb[r >> 3][r & 7] = (bf16)(pv * (dp[r] - dlt) * p.scale);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16* kp = Ktb + (int64_t)dt * 32 * p.Skpad + key0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) dqacc[dt] = MFMA(ld4x2(kp + 16 * s2, kp + 16 * s2 + 8), dsb[s2], dqacc[dt]);
        }
    }
    if (q < p.S) {
        bf16* dQp = p.dQ + b * p.dq_bs + h * p.dq_hs + (int64_t)q * p.dq_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)dqacc[dt][4 * qd + e];
                *(bf16x4*)(dQp + dt * 32 + 8 * qd + 4 * hi) = o;
            }
    }
}

int check_common(const char* name, int B, int Hq, int Hkv, int S, int Spad, int D) {
    if (!(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && S > 0)) return afk_set_error(AFK_ERR_ARG, "%s: bad shape", name);
    if (!(D == 64 || D == 128 || D == 32)) return afk_set_error(AFK_ERR_UNSUPPORTED, "%s: head_dim %d (32/64/128 supported)", name, D);
    if (!(Spad >= S && Spad % 32 == 0 && Spad >= ((S + 31) / 32) * 32)) return afk_set_error(AFK_ERR_ARG, "%s: Spad=%d must be a multiple of 32 >= S", name, Spad);
    return AFK_OK;
}

}  // namespace

extern "C" int afk_attn_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                            int64_t k_rs, const void* Vt, void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, float* LSE,
                            const int* kv_len, int B, int Hq, int Hkv, int S, int Spad, int D, float scale, int causal,
                            void* stream) {
    AFK_REQUIRE(Q && K && Vt && O, "afk_attn_fwd: null pointer");
    afk_count(AFK_CNT_ATTN1_FWD);
    if (int e = check_common("afk_attn_fwd", B, Hq, Hkv, S, Spad, D)) return e;
    AFK_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 && o_rs % 4 == 0 && o_hs % 4 == 0,
                "afk_attn_fwd: strides must keep 16-byte alignment");
    AttnArgs p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.Vt = (const bf16*)Vt;
    p.O = (bf16*)O; p.o_bs = o_bs; p.o_hs = o_hs; p.o_rs = o_rs;
    p.LSE = LSE; p.kv_len = kv_len;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.Sk = S; p.Skpad = Spad; p.scale = scale; p.causal = causal;
    dim3 grid((unsigned)Hq, (unsigned)B, (unsigned)afk_cdiv(S, 128));
    hipStream_t st = (hipStream_t)stream;
    if (D == 128) hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), 0, st, p);
    else if (D == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), 0, st, p);
    AFK_LAUNCH_CHECK("afk_attn_fwd");
    return AFK_OK;
}

extern "C" int afk_attn_delta(const void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, const void* dO, int64_t do_bs,
                              int64_t do_hs, int64_t do_rs, float* delta, int B, int H, int S, int D, void* stream) {
    AFK_REQUIRE(O && dO && delta, "afk_attn_delta: null pointer");
    AFK_REQUIRE(D == 32 || D == 64 || D == 128, "afk_attn_delta: head_dim %d unsupported", D);
    const int64_t total = (int64_t)B * H * S;
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)afk_cdiv(total, 256 / (D / 8));
    if (grid > 8192) grid = 8192;
    if (D == 128) hipLaunchKernelGGL(attn_delta_kernel<128>, dim3(grid), dim3(256), 0, st, (const bf16*)O, o_bs, o_hs, o_rs, (const bf16*)dO, do_bs, do_hs, do_rs, delta, B, H, S);
    else if (D == 64) hipLaunchKernelGGL(attn_delta_kernel<64>, dim3(grid), dim3(256), 0, st, (const bf16*)O, o_bs, o_hs, o_rs, (const bf16*)dO, do_bs, do_hs, do_rs, delta, B, H, S);
    else hipLaunchKernelGGL(attn_delta_kernel<32>, dim3(grid), dim3(256), 0, st, (const bf16*)O, o_bs, o_hs, o_rs, (const bf16*)dO, do_bs, do_hs, do_rs, delta, B, H, S);
    AFK_LAUNCH_CHECK("afk_attn_delta");
    return AFK_OK;
}

extern "C" int afk_attn_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                            int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO,
                            int64_t do_bs, int64_t do_hs, int64_t do_rs, const void* Qt, const void* Kt, const void* dOt,
                            const float* LSE, const float* delta, void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs,
                            void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV, int64_t dv_bs, int64_t dv_hs,
                            int64_t dv_rs, const int* kv_len, int B, int Hq, int Hkv, int S, int Spad, int D, float scale,
                            int causal, void* stream) {
    AFK_REQUIRE(Q && K && V && dO && Qt && Kt && dOt && LSE && delta && dQ && dK && dV, "afk_attn_bwd: null pointer");
    afk_count(AFK_CNT_ATTN1_BWD);
    if (int e = check_common("afk_attn_bwd", B, Hq, Hkv, S, Spad, D)) return e;
    AttnArgs p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.V = (const bf16*)V; p.v_bs = v_bs; p.v_hs = v_hs; p.v_rs = v_rs;
    p.dO = (const bf16*)dO; p.do_bs = do_bs; p.do_hs = do_hs; p.do_rs = do_rs;
    p.dQ = (bf16*)dQ; p.dq_bs = dq_bs; p.dq_hs = dq_hs; p.dq_rs = dq_rs;
    p.dK = (bf16*)dK; p.dk_bs = dk_bs; p.dk_hs = dk_hs; p.dk_rs = dk_rs;
    p.dV = (bf16*)dV; p.dv_bs = dv_bs; p.dv_hs = dv_hs; p.dv_rs = dv_rs;
    p.Qt = (const bf16*)Qt; p.Kt = (const bf16*)Kt; p.dOt = (const bf16*)dOt;
    p.LSE = (float*)LSE; p.delta = delta; p.kv_len = kv_len;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = S; p.Spad = Spad; p.Sk = S; p.Skpad = Spad; p.scale = scale; p.causal = causal;
    hipStream_t st = (hipStream_t)stream;
    dim3 gkv((unsigned)Hkv, (unsigned)B, (unsigned)afk_cdiv(S, 128));
    dim3 gq((unsigned)Hq, (unsigned)B, (unsigned)afk_cdiv(S, 128));
    if (D == 128) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<128>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), 0, st, p);
    } else if (D == 64) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<64>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<32>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, gq, dim3(256), 0, st, p);
    }
    AFK_LAUNCH_CHECK("afk_attn_bwd");
    return AFK_OK;
}

// ---- cross-attention form: Sq != Sk, optional per-query key range (block-diagonal "attend only to your own media segment"
// masks of the Flamingo gated cross-attention, BASELINE config 4).  Qt/dOt/LSE/delta use the query pitch, Kt/Vt the key pitch.
extern "C" int afk_xattn_fwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* Vt, void* O, int64_t o_bs, int64_t o_hs, int64_t o_rs, float* LSE,
                             const int* kv_len, const int* krange, int B, int Hq, int Hkv, int Sq, int Sk, int Sqpad, int Skpad,
                             int D, float scale, void* stream) {
    AFK_REQUIRE(Q && K && Vt && O && LSE, "afk_xattn_fwd: null pointer");
    afk_count(AFK_CNT_XATTN_FWD);
    if (int e = check_common("afk_xattn_fwd", B, Hq, Hkv, Sq, Sqpad, D)) return e;
    if (int e = check_common("afk_xattn_fwd", B, Hq, Hkv, Sk, Skpad, D)) return e;
    AttnArgs p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.Vt = (const bf16*)Vt;
    p.O = (bf16*)O; p.o_bs = o_bs; p.o_hs = o_hs; p.o_rs = o_rs;
    p.LSE = LSE; p.kv_len = kv_len; p.krange = krange;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = Sq; p.Spad = Sqpad; p.Sk = Sk; p.Skpad = Skpad; p.scale = scale; p.causal = 0;
    dim3 grid((unsigned)Hq, (unsigned)B, (unsigned)afk_cdiv(Sq, 128));
    hipStream_t st = (hipStream_t)stream;
    if (D == 128) hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), 0, st, p);
    else if (D == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), 0, st, p);
    AFK_LAUNCH_CHECK("afk_xattn_fwd");
    return AFK_OK;
}

extern "C" int afk_xattn_bwd(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* K, int64_t k_bs, int64_t k_hs,
                             int64_t k_rs, const void* V, int64_t v_bs, int64_t v_hs, int64_t v_rs, const void* dO,
                             int64_t do_bs, int64_t do_hs, int64_t do_rs, const void* Qt, const void* Kt, const void* dOt,
                             const float* LSE, const float* delta, void* dQ, int64_t dq_bs, int64_t dq_hs, int64_t dq_rs,
                             void* dK, int64_t dk_bs, int64_t dk_hs, int64_t dk_rs, void* dV, int64_t dv_bs, int64_t dv_hs,
                             int64_t dv_rs, const int* kv_len, const int* krange, int B, int Hq, int Hkv, int Sq, int Sk,
                             int Sqpad, int Skpad, int D, float scale, void* stream) {
    AFK_REQUIRE(Q && K && V && dO && Qt && Kt && dOt && LSE && delta && dQ && dK && dV, "afk_xattn_bwd: null pointer");
    afk_count(AFK_CNT_XATTN_BWD);
    if (int e = check_common("afk_xattn_bwd", B, Hq, Hkv, Sq, Sqpad, D)) return e;
    if (int e = check_common("afk_xattn_bwd", B, Hq, Hkv, Sk, Skpad, D)) return e;
    AttnArgs p = {};
    p.Q = (const bf16*)Q; p.q_bs = q_bs; p.q_hs = q_hs; p.q_rs = q_rs;
    p.K = (const bf16*)K; p.k_bs = k_bs; p.k_hs = k_hs; p.k_rs = k_rs;
    p.V = (const bf16*)V; p.v_bs = v_bs; p.v_hs = v_hs; p.v_rs = v_rs;
    p.dO = (const bf16*)dO; p.do_bs = do_bs; p.do_hs = do_hs; p.do_rs = do_rs;
    p.dQ = (bf16*)dQ; p.dq_bs = dq_bs; p.dq_hs = dq_hs; p.dq_rs = dq_rs;
    p.dK = (bf16*)dK; p.dk_bs = dk_bs; p.dk_hs = dk_hs; p.dk_rs = dk_rs;
    p.dV = (bf16*)dV; p.dv_bs = dv_bs; p.dv_hs = dv_hs; p.dv_rs = dv_rs;
    p.Qt = (const bf16*)Qt; p.Kt = (const bf16*)Kt; p.dOt = (const bf16*)dOt;
    p.LSE = (float*)LSE; p.delta = delta; p.kv_len = kv_len; p.krange = krange;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.S = Sq; p.Spad = Sqpad; p.Sk = Sk; p.Skpad = Skpad; p.scale = scale; p.causal = 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 gkv((unsigned)Hkv, (unsigned)B, (unsigned)afk_cdiv(Sk, 128));
    dim3 gq((unsigned)Hq, (unsigned)B, (unsigned)afk_cdiv(Sq, 128));
    if (D == 128) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<128>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), 0, st, p);
    } else if (D == 64) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<64>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<32>, gkv, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, gq, dim3(256), 0, st, p);
    }
    AFK_LAUNCH_CHECK("afk_xattn_bwd");
    return AFK_OK;
}
