// Decode-time attention (one query row per sample, keys/values in the KV cache): split-KV ("flash decoding").
//
// Oracle: Qwen2Attention.forward with a cache, modeling_qwen2.py:195-234 (softmax(q k^T * scale) v over the cached positions).
// A Q = 1 problem has no query dimension to parallelise over, so one block per (batch, head) walks the whole cache serially
// (measured 54 us per layer at 800 cached tokens on the interval MFMA kernel: 1.5 ms of a 5.1 ms token).  Here the key range of each
// sample is cut into `nsplit` chunks: grid (nsplit, Hq, B); every block scores its chunk with coalesced 16-byte key-row loads
// (D/8 lanes per key, xor-shuffle reduce), takes its own softmax (max m, sum l) and its partial output  sum_k p_k V[k]  from the
// TRANSPOSED value cache Vt[b][hk][d][key] (thread = one d, 8 keys per 16-byte load), and leaves (m, l, o[D]) in a workspace; a second
// tiny kernel merges the chunks:  o = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s.   HBM-bound: the cache is read exactly once.
// The visible key interval [lo, hi) of every sample comes from DEVICE memory (krange), so a captured HIP graph of the decode step can be
// replayed while the sequence grows.
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int MAXCHUNK = 4096;  // keys per split held in LDS

template <int D>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(const bf16* __restrict__ Q, int64_t q_bs, int64_t q_hs, const bf16* __restrict__ Kc,
                                                                int64_t k_bs, int64_t k_rs, int64_t k_hs, const bf16* __restrict__ Vt,
                                                                int64_t vt_bs, int spad, const int* __restrict__ krange, int Hq, int Hkv,
                                                                float scale, float* __restrict__ ws) {
    constexpr int LPK = D / 8;        // lanes per key row
    constexpr int KPP = 256 / LPK;    // keys scored per pass of the block
    constexpr int PARTS = 256 / D;    // threads per output feature
    __shared__ float sc[MAXCHUNK];
    __shared__ float red[8];
    __shared__ float part[256];
    const int split = blockIdx.x, nsplit = gridDim.x, h = blockIdx.y, b = blockIdx.z, hk = h / (Hq / Hkv);
    const int t = threadIdx.x;
    const int lo = krange[2 * b], hi = krange[2 * b + 1];
    // chunk boundaries on multiples of 8 keys (16-byte alignment of the transposed value rows)
    const int a0 = lo & ~7, total = max(hi - a0, 0);
    const int chunk = min((((total + nsplit - 1) / nsplit) + 7) & ~7, MAXCHUNK);
    const int c0 = a0 + split * chunk, c1 = min(c0 + chunk, hi);  // keys [max(c0, lo), c1)
    const int n = max(c1 - c0, 0);
    float* out = ws + ((int64_t)(b * Hq + h) * nsplit + split) * (D + 2);
    if (n <= 0 || c1 <= lo) {
        if (t < D) out[2 + t] = 0.f;
        if (t == 0) { out[0] = NEG_INF; out[1] = 0.f; }
        return;
    }
    // ---- scores
    const int sub = t % LPK;
    const bf16x8 qv = *(const bf16x8*)(Q + b * q_bs + h * q_hs + sub * 8);
    const bf16* kbase = Kc + b * k_bs + hk * k_hs + sub * 8;
    const float c2 = scale * LOG2E;
    float mx = NEG_INF;
    for (int i0 = 0; i0 < n; i0 += KPP) {
        const int i = i0 + t / LPK;
        const int key = c0 + i;
        float s = 0.f;
        if (i < n) {
            const bf16x8 kv = *(const bf16x8*)(kbase + (int64_t)key * k_rs);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16x2 a = {qv[2 * e], qv[2 * e + 1]}, c = {kv[2 * e], kv[2 * e + 1]};
                s = __builtin_amdgcn_fdot2_f32_bf16(a, c, s, false);
            }
        }
#pragma unroll
        for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (i < n && sub == 0) {
            const float v = (key >= lo) ? s * c2 : NEG_INF;  // log2 domain
            sc[i] = v;
        }
        if (i < n && key >= lo) mx = fmaxf(mx, s * c2);
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float msafe = (m == NEG_INF) ? 0.f : m;
    float ls = 0.f;
    for (int i = t; i < n; i += 256) {
        const float p = __builtin_amdgcn_exp2f(sc[i] - msafe);
        sc[i] = p;
        ls += p;
    }
    ls = wave_sum(ls);
    if ((t & 63) == 0) red[4 + (t >> 6)] = ls;
    __syncthreads();
    const float l = red[4] + red[5] + red[6] + red[7];
    // ---- partial output: thread -> feature d, part -> every PARTS-th group of 8 keys
    const int d = t % D, pt = t / D;
    const bf16* vrow = Vt + b * vt_bs + ((int64_t)hk * D + d) * spad + c0;  // c0 % 8 == 0
    float acc = 0.f;
    for (int g = pt * 8; g < n; g += PARTS * 8) {
        const bf16x8 vv = *(const bf16x8*)(vrow + g);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (g + e < n) acc += sc[g + e] * (float)vv[e];
    }
    part[t] = acc;
    __syncthreads();
    if (t < D) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) o += part[t + q * D];
        out[2 + t] = o;
    }
    if (t == 0) { out[0] = m; out[1] = l; }
}

template <int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float* __restrict__ ws, int nsplit, bf16* __restrict__ O, int64_t o_bs,
                                                               int64_t o_hs, int Hq) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* base = ws + (int64_t)(b * Hq + h) * nsplit * (D + 2);
    float M = NEG_INF;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[s * (D + 2)]);
    float L = 0.f, o = 0.f;
    if (M != NEG_INF) {
        for (int s = 0; s < nsplit; ++s) {
            const float w = __builtin_amdgcn_exp2f(base[s * (D + 2)] - M);  // m = -inf -> 0
            L += w * base[s * (D + 2) + 1];
            o += w * base[s * (D + 2) + 2 + d];
        }
    }
    O[b * o_bs + h * o_hs + d] = (bf16)(L > 0.f ? o / L : 0.f);
}

}  // namespace

extern "C" int afk_attn_decode_workspace_floats(int B, int Hq, int D, int nsplit) { return B * Hq * nsplit * (D + 2); }

extern "C" int afk_attn_decode(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                               const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                               int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    AFK_REQUIRE(Q && Kc && Vt && O && krange && workspace, "afk_attn_decode: null pointer");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && (D == 64 || D == 128) && nsplit >= 1 && nsplit <= 64, "afk_attn_decode: bad shape");
    AFK_REQUIRE(q_bs % 8 == 0 && q_hs % 8 == 0 && k_bs % 8 == 0 && k_rs % 8 == 0 && k_hs % 8 == 0 && vt_bs % 8 == 0 && spad % 8 == 0,
                "afk_attn_decode: strides must keep 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nsplit, (unsigned)Hq, (unsigned)B);
    if (D == 128) {
        hipLaunchKernelGGL(attn_decode_split_kernel<128>, grid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs, k_hs,
                           (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace);
        hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3((unsigned)Hq, (unsigned)B), dim3(128), 0, st, workspace, nsplit, (bf16*)O, o_bs,
                           o_hs, Hq);
    } else {
        hipLaunchKernelGGL(attn_decode_split_kernel<64>, grid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs, k_hs,
                           (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace);
        hipLaunchKernelGGL(attn_decode_combine_kernel<64>, dim3((unsigned)Hq, (unsigned)B), dim3(64), 0, st, workspace, nsplit, (bf16*)O, o_bs,
                           o_hs, Hq);
    }
    AFK_LAUNCH_CHECK("afk_attn_decode");
    return AFK_OK;
}
