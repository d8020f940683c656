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
#include <atomic>
#include "common.h"
#include "../../include/afk.h"

namespace {

constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int MAXCHUNK = 4096;  // keys per split held in LDS

#ifdef AFK_PROBES
// timing probe (PROBES=1 builds only, tools/probes/probe_attn_decode.py): thread 0 of every block stamps the 100 MHz wall clock at the phase boundaries
__device__ long long* g_stamps = nullptr;
#define AFK_STAMP(k)                                                                                                      \
    do {                                                                                                                  \
        if (g_stamps && threadIdx.x == 0) g_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define AFK_STAMP(k)
#endif

// FINAL (round 4, afk_attn_decode_fused): no combine launch - the LAST chunk block of a (batch, head) pair to finish merges the chunks itself.
// The chunks of one head run on different XCDs whose L2s are not coherent, so the hand-over goes through memory:
//   sync 1 (default): every (m, l, o) word is an agent-scope atomic store (write-through, `sc1`), each wave drains its stores (vmcnt 0) before the block
//           barrier, thread 0 then bumps the pair's counter; the block that reads nsplit - 1 loads the slots with agent-scope atomic loads (they bypass
//           the non-coherent cache levels).  No cache-wide operation anywhere.
//   sync 0: plain stores, thread 0 brackets the counter update with agent-scope fences after / before the block barrier (L2 write-back + invalidate of
//           the whole XCD L2: measured 29 us per launch when EVERY thread fenced, profiles/r04_decode_chain.md).
// The merge folds the partials in chunk order with the arithmetic of attn_decode_combine_kernel - the result does not depend on which block came last -
// and resets the counter for the next call.  Nobody waits for anybody.
// WPE: waves per SIMD the register allocation aims for (amdgpu_waves_per_eu).  Left alone the D = 128 hand-over form takes 142 VGPRs = 3 blocks per CU; a batched step
// launches 7 blocks per CU (B = 8: 1 792) that each live ~10 us of memory round trips, so residency, not bytes, sets its time (round 6 probe: a quarter of the
// blocks entered at once, the last after 25 us).  4 -> 102 VGPRs without a spill, 5 -> 96 (+ 6 spilled), 6 -> 80 (+ 18 spilled).
template <int D, int FINAL, int WPE = 3, int BK = 128>   // BK: keys per register batch; FINAL: 0 = partials only (afk_attn_decode), 1 = merge by the last block, write-through hand-over, 2 = the same with agent-scope fences
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void attn_decode_split_kernel(const bf16* __restrict__ Q, int64_t q_bs, int64_t q_hs, const bf16* __restrict__ Kc,
                                                                int64_t k_bs, int64_t k_rs, int64_t k_hs, const bf16* __restrict__ Vt,
                                                                int64_t vt_bs, int spad, const int* __restrict__ krange, int Hq, int Hkv,
                                                                float scale, float* __restrict__ ws, bf16* __restrict__ O, int64_t o_bs, int64_t o_hs) {
    constexpr int LPK = D / 8;        // lanes per key row
    constexpr int KPP = 256 / LPK;    // keys scored per pass of the block
    constexpr int PARTS = 256 / D;    // threads per output feature
    __shared__ __attribute__((aligned(16))) float sc[MAXCHUNK + 8];
    __shared__ float red[8];
    __shared__ float part[256];
    const int split = blockIdx.x, nsplit = gridDim.x, h = blockIdx.y, b = blockIdx.z, hk = h / (Hq / Hkv);
    const int t = threadIdx.x;
    AFK_STAMP(0);
    const int lo = krange[2 * b], hi = krange[2 * b + 1];
    // chunk boundaries on multiples of 8 keys (16-byte alignment of the transposed value rows)
    const int a0 = lo & ~7, total = max(hi - a0, 0);
    const int chunk = min((((total + nsplit - 1) / nsplit) + 7) & ~7, MAXCHUNK);
    const int c0 = a0 + split * chunk, c1 = min(c0 + chunk, hi);  // keys [max(c0, lo), c1)
    const int n = max(c1 - c0, 0);
    float* out = ws + ((int64_t)(b * Hq + h) * nsplit + split) * (D + 2);
    __shared__ int last_flag;
    // FINAL: the merge by the last-arriving block of this (batch, head) pair
    constexpr bool wt = FINAL == 1;
    auto put = [&](float* dst, float v) {
        if (wt) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = v;
    };
    auto get = [&](const float* src) -> float {
        return wt ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(src);
    };
    auto finish = [&]() {
        if (!FINAL) return;
        int* counters = (int*)(ws + (int64_t)gridDim.z * Hq * nsplit * (D + 2));
        // every storing wave drains its agent-scope write-through stores itself (vmcnt counts stores on gfx9 and falls when the write has been acknowledged at the
        // coherence point).  A workgroup-scope release fence does NOT do this on gfx950 outside tgsplit mode - it emits no vmcnt wait (round-4 ISA: store sc1 ->
        // s_barrier -> atomic add, nothing in between) - so the wait is written out; the block barrier below then orders all of the block's stores before
        // thread 0's counter bump.
        if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        AFK_STAMP(5);
        if (t == 0) {
            if (!wt) __threadfence();                      // release: the block's (m, l, o) leave this XCD's L2 before the counter moves
            const int prev = __hip_atomic_fetch_add(&counters[b * Hq + h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (prev == nsplit - 1);
            if (last_flag) {
                __hip_atomic_store(&counters[b * Hq + h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-resetting: the next call (a HIP-graph replay) starts from zero
                if (!wt) __threadfence();                  // acquire: drop what this CU / XCD cached of the other blocks' slots
            }
        }
        __syncthreads();
        AFK_STAMP(6);
        if (!last_flag) return;
        // all nsplit x (D + 2) words in parallel (4 independent loads per thread and round: a loop of dependent round trips to memory costs ~1 us each),
        // staged in the score buffer - nobody reads scores any more
        const float* base = ws + (int64_t)(b * Hq + h) * nsplit * (D + 2);
        const int tot = nsplit * (D + 2);
        constexpr int NLM = 5;   // 8 chunks x (128 + 2) words = 1 040: with four loads per thread and round (1 024 words) the last 16 words cost a second memory round trip
        for (int i = t; i < tot; i += 256 * NLM) {
            float v[NLM];
#pragma unroll
            for (int u = 0; u < NLM; ++u) v[u] = get(base + min(i + 256 * u, tot - 1));   // clamped: all loads of a round in flight, no branch between them
#pragma unroll
            for (int u = 0; u < NLM; ++u)
                if (i + 256 * u < tot) sc[i + 256 * u] = v[u];
        }
        __syncthreads();
        if (t < D) {
            float M = NEG_INF;
            for (int s = 0; s < nsplit; ++s) M = fmaxf(M, sc[s * (D + 2)]);
            float L = 0.f, o = 0.f;
            if (M != NEG_INF) {
                for (int s = 0; s < nsplit; ++s) {
                    const float w = __builtin_amdgcn_exp2f(sc[s * (D + 2)] - M);
                    L += w * sc[s * (D + 2) + 1];
                    o += w * sc[s * (D + 2) + 2 + t];
                }
            }
            O[b * o_bs + h * o_hs + t] = (bf16)(L > 0.f ? o / L : 0.f);
        }
        AFK_STAMP(7);
    };
    if (n <= 0 || c1 <= lo) {
        if (t < D) put(out + 2 + t, 0.f);
        if (t == 0) { put(out, NEG_INF); put(out + 1, 0.f); }
        finish();
        return;
    }
    // ---- every load of the first 128 keys of the chunk is issued before anything is computed (round 4: the one-load-per-pass loops spent a memory
    // round trip per 16 keys - 10 us for a 100-key chunk); longer chunks continue in batches of the same depth
    constexpr int UK = BK / KPP;     // key passes per batch
    constexpr int UV = BK / (PARTS * 8);   // value loads per batch and thread
    const int sub = t % LPK, krow = t / LPK;
    const bf16x8 qv = *(const bf16x8*)(Q + b * q_bs + h * q_hs + sub * 8);
    const bf16* kbase = Kc + b * k_bs + hk * k_hs + sub * 8 + (int64_t)c0 * k_rs;
    const int d = t % D, pt = t / D;
    const bf16* vrow = Vt + b * vt_bs + ((int64_t)hk * D + d) * spad + c0;  // c0 % 8 == 0
    bf16x8 kv[UK], vv[UV];
    auto load_k = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UK; ++u) kv[u] = *(const bf16x8*)(kbase + (int64_t)min(i0 + u * KPP + krow, n - 1) * k_rs);   // clamped: valid memory, masked below
    };
    auto load_v = [&](int g0) {
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int g = g0 + (u * PARTS + pt) * 8;
            vv[u] = *(const bf16x8*)(vrow + (g < n ? g : 0));
        }
    };
    AFK_STAMP(1);
    load_k(0);
    load_v(0);
    const float c2 = scale * LOG2E;
    float mx = NEG_INF;
    for (int i0 = 0; i0 < n; i0 += BK) {
        if (i0) load_k(i0);
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int i = i0 + u * KPP + krow;
            const int key = c0 + i;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16x2 a = {qv[2 * e], qv[2 * e + 1]}, c = {kv[u][2 * e], kv[u][2 * e + 1]};
                s = __builtin_amdgcn_fdot2_f32_bf16(a, c, s, false);
            }
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (i < n && sub == 0) sc[i] = (key >= lo) ? s * c2 : NEG_INF;  // log2 domain
            if (i < n && key >= lo) mx = fmaxf(mx, s * c2);
        }
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    AFK_STAMP(2);
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
    AFK_STAMP(3);
    // ---- partial output: thread -> feature d, part -> every PARTS-th group of 8 keys
    // (branch-free: written as `if (g + e < n) acc += ...` the compiler emitted one branch + one LDS round trip per key - 2.7 us of a 11 us launch)
    float acc = 0.f;
    for (int g0 = 0; g0 < n; g0 += BK) {
        if (g0) load_v(g0);
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int g = g0 + (u * PARTS + pt) * 8;   // wave-uniform
            const int gg = g < n ? g : 0;
            const f32x4 p0 = *(const f32x4*)&sc[gg], p1 = *(const f32x4*)&sc[gg + 4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = g + e < n;
                const float pe = in ? (e < 4 ? p0[e & 3] : p1[e & 3]) : 0.f;
                const float ve = in ? (float)vv[u][e] : 0.f;   // what lies behind the last key may be anything
                acc = fmaf(pe, ve, acc);
            }
        }
    }
    part[t] = acc;
    __syncthreads();
    AFK_STAMP(4);
    if (t < D) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) o += part[t + q * D];
        put(out + 2 + t, o);
    }
    if (t == 0) { put(out, m); put(out + 1, l); }
    finish();
}

// Round 5 - batched decode (afk_attn_decode_fused at B x Hkv x nsplit >= GROUP_MIN_BLOCKS): ONE block per (sample, KV head, key chunk) serves ALL G = Hq / Hkv
// query heads of the group.  The per-head form above launches G blocks that each fetch the same K / V chunk - at B = 8 that was 1 792 blocks and 26 us per
// layer for 13 MB of cache (profiles/r04_decode_kernel_stats.md).  Here a key row is loaded once and scored against the G query rows the thread holds in
// registers, the transposed value rows are loaded once and accumulated into G outputs.  The arithmetic of every head - dot-product order, shuffle reduction,
// per-thread strided sums, fold order of the parts, the merge - is that of attn_decode_split_kernel element for element, so the results are BIT-IDENTICAL to
// the per-head form (tests/test_ops_gpu.py::test_attn_decode_group_kernel_bit_equal).  Same workspace layout; the arrival counter of a group is the slot of
// its first head.  Hand-over: write-through stores drained by every storing wave (s_waitcnt vmcnt(0)) before the barrier and the counter bump.
constexpr int GCHUNK = 1024;         // keys per chunk the group form holds in LDS (G score rows)
constexpr int GROUP_MIN_BLOCKS = 128;
constexpr int GROUP_LDS_FLOATS = 8 * (GCHUNK + 24);
template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_group_kernel(const bf16* __restrict__ Q, int64_t q_bs, int64_t q_hs, const bf16* __restrict__ Kc,
                                                                int64_t k_bs, int64_t k_rs, int64_t k_hs, const bf16* __restrict__ Vt,
                                                                int64_t vt_bs, int spad, const int* __restrict__ krange, int Hq, int Hkv,
                                                                float scale, float* __restrict__ ws, bf16* __restrict__ O, int64_t o_bs, int64_t o_hs) {
    constexpr int LPK = D / 8, KPP = 256 / LPK, PARTS = 256 / D;
    constexpr int SROW = GCHUNK + 8;
    __shared__ __attribute__((aligned(16))) float sc[GROUP_LDS_FLOATS];   // [G][SROW] scores / probabilities; later the merge's staging area
    __shared__ float red[2][G][4];
    __shared__ float part[G][256];
    __shared__ int last_flag;
    const int split = blockIdx.x, nsplit = gridDim.x, hk = blockIdx.y, b = blockIdx.z, h0 = hk * G;
    const int t = threadIdx.x;
    const int lo = krange[2 * b], hi = krange[2 * b + 1];
    const int a0 = lo & ~7, total = max(hi - a0, 0);
    const int chunk = min((((total + nsplit - 1) / nsplit) + 7) & ~7, GCHUNK);
    const int c0 = a0 + split * chunk, c1 = min(c0 + chunk, hi);
    const int n = max(c1 - c0, 0);
    auto slot = [&](int g) { return ws + ((int64_t)(b * Hq + h0 + g) * nsplit + split) * (D + 2); };
    auto put = [&](float* dst, float v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto get = [&](const float* src) -> float { return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto finish = [&]() {
        int* counters = (int*)(ws + (int64_t)gridDim.z * Hq * nsplit * (D + 2));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through partials have landed (see attn_decode_split_kernel)
        __syncthreads();
        if (t == 0) {
            const int prev = __hip_atomic_fetch_add(&counters[b * Hq + h0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (prev == nsplit - 1);
            if (last_flag) __hip_atomic_store(&counters[b * Hq + h0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_flag) return;
        // the G heads' partials are contiguous in the workspace: [h0 .. h0 + G)[nsplit][D + 2]
        const float* base = ws + (int64_t)(b * Hq + h0) * nsplit * (D + 2);
        const int tot = G * nsplit * (D + 2);
        // G x nsplit x (D + 2) words (7 280 at the AF3 geometry): SIXTEEN independent agent-scope loads per thread and round - these loads bypass the cache
        // levels, a round costs a memory round trip (~2 us), and four per round (the per-head kernel's depth) made the merge 8 dependent trips
        constexpr int NL = 16;
        for (int i = t; i < tot; i += 256 * NL) {
            float v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) v[u] = get(base + min(i + 256 * u, tot - 1));
#pragma unroll
            for (int u = 0; u < NL; ++u)
                if (i + 256 * u < tot) sc[i + 256 * u] = v[u];
        }
        __syncthreads();
        for (int g = t / D; g < G; g += PARTS) {   // PARTS heads at a time, thread -> feature d
            const int d = t % D;
            const float* sg = sc + g * nsplit * (D + 2);
            float M = NEG_INF;
            for (int s = 0; s < nsplit; ++s) M = fmaxf(M, sg[s * (D + 2)]);
            float L = 0.f, o = 0.f;
            if (M != NEG_INF) {
                for (int s = 0; s < nsplit; ++s) {
                    const float w = __builtin_amdgcn_exp2f(sg[s * (D + 2)] - M);
                    L += w * sg[s * (D + 2) + 1];
                    o += w * sg[s * (D + 2) + 2 + d];
                }
            }
            O[b * o_bs + (h0 + g) * o_hs + d] = (bf16)(L > 0.f ? o / L : 0.f);
        }
    };
    if (n <= 0 || c1 <= lo) {
        for (int g = 0; g < G; ++g) {
            if (t < D) put(slot(g) + 2 + t, 0.f);
            if (t == 0) { put(slot(g), NEG_INF); put(slot(g) + 1, 0.f); }
        }
        finish();
        return;
    }
    constexpr int UK = 128 / KPP, UV = 128 / (PARTS * 8);
    const int sub = t % LPK, krow = t / LPK;
    bf16x8 qv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) qv[g] = *(const bf16x8*)(Q + b * q_bs + (h0 + g) * q_hs + sub * 8);
    const bf16* kbase = Kc + b * k_bs + hk * k_hs + sub * 8 + (int64_t)c0 * k_rs;
    const int d = t % D, pt = t / D;
    const bf16* vrow = Vt + b * vt_bs + ((int64_t)hk * D + d) * spad + c0;
    bf16x8 kv[UK], vv[UV];
    auto load_k = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UK; ++u) kv[u] = *(const bf16x8*)(kbase + (int64_t)min(i0 + u * KPP + krow, n - 1) * k_rs);
    };
    auto load_v = [&](int g0) {
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int gi = g0 + (u * PARTS + pt) * 8;
            vv[u] = *(const bf16x8*)(vrow + (gi < n ? gi : 0));
        }
    };
    load_k(0);
    load_v(0);
    const float c2 = scale * LOG2E;
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = NEG_INF;
    for (int i0 = 0; i0 < n; i0 += 128) {
        if (i0) load_k(i0);
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int i = i0 + u * KPP + krow;
            const int key = c0 + i;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 a = {qv[g][2 * e], qv[g][2 * e + 1]}, c = {kv[u][2 * e], kv[u][2 * e + 1]};
                    s = __builtin_amdgcn_fdot2_f32_bf16(a, c, s, false);
                }
#pragma unroll
                for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (i < n && sub == 0) sc[g * SROW + i] = (key >= lo) ? s * c2 : NEG_INF;
                if (i < n && key >= lo) mx[g] = fmaxf(mx[g], s * c2);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float w = wave_max(mx[g]);
        if ((t & 63) == 0) red[0][g][t >> 6] = w;
    }
    __syncthreads();
    float m[G], msafe[G], ls[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = fmaxf(fmaxf(red[0][g][0], red[0][g][1]), fmaxf(red[0][g][2], red[0][g][3]));
        msafe[g] = (m[g] == NEG_INF) ? 0.f : m[g];
        ls[g] = 0.f;
    }
    for (int i = t; i < n; i += 256) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float p = __builtin_amdgcn_exp2f(sc[g * SROW + i] - msafe[g]);
            sc[g * SROW + i] = p;
            ls[g] += p;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float w = wave_sum(ls[g]);
        if ((t & 63) == 0) red[1][g][t >> 6] = w;
    }
    __syncthreads();
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
    for (int g0 = 0; g0 < n; g0 += 128) {
        if (g0) load_v(g0);
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int gi = g0 + (u * PARTS + pt) * 8;   // wave-uniform
            const int gg = gi < n ? gi : 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 p0 = *(const f32x4*)&sc[g * SROW + gg], p1 = *(const f32x4*)&sc[g * SROW + gg + 4];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool in = gi + e < n;
                    const float pe = in ? (e < 4 ? p0[e & 3] : p1[e & 3]) : 0.f;
                    const float ve = in ? (float)vv[u][e] : 0.f;
                    acc[g] = fmaf(pe, ve, acc[g]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) part[g][t] = acc[g];
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        if (t < D) {
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) o += part[g][t + q * D];
            put(slot(g) + 2 + t, o);
        }
        if (t == 0) { put(slot(g), m[g]); put(slot(g) + 1, red[1][g][0] + red[1][g][1] + red[1][g][2] + red[1][g][3]); }
    }
    finish();
}

// Round 6 - the group form on the MATRIX PIPE (afk_attn_decode_set_group(3) / AFK_ATTN_DECODE_GROUP=3; default for batched steps, see attn_decode_impl).  The per-head
// form brings 7 blocks per CU to a B = 8 step and is bound by wave dispatch (224 blocks per XCD, ~36 ns each) + one 10 us block life; the VALU group form above has 256
// blocks but G times the dot-product work per thread.  Here a block = (sample, KV head, key chunk) serves the G <= 8 query heads of the group with two small GEMMs:
//   S[key][head]  = K[key][:] . Q[head][:]      v_mfma_f32_32x32x16_bf16, A = 32 key rows (16-byte pieces straight from the cache rows), B = the G query rows
//   O[d][head]   += Vt[d][key] . P[key][head]   A = 32 rows of the transposed value cache (8 consecutive keys per lane), B = the bf16 probabilities
// wave w owns key tiles w, w + 4, ... of the scores and feature tile(s) w (, w + 4) of the output; softmax (chunk maximum, exp2, sum of the bf16-rounded
// probabilities - what the P.V product sees, as the training kernels count) goes through the score rows in LDS exactly as above, and so do the partials, the
// hand-over and the merge (same workspace layout).  Summation order differs from the per-head kernel: equal within fp32 / bf16 rounding, not bit for bit.
template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_gmma_kernel(const bf16* __restrict__ Q, int64_t q_bs, int64_t q_hs, const bf16* __restrict__ Kc,
                                                               int64_t k_bs, int64_t k_rs, int64_t k_hs, const bf16* __restrict__ Vt,
                                                               int64_t vt_bs, int spad, const int* __restrict__ krange, int Hq, int Hkv,
                                                               float scale, float* __restrict__ ws, bf16* __restrict__ O, int64_t o_bs, int64_t o_hs) {
    constexpr int PARTS = 256 / D;
    constexpr int SROW = GCHUNK + 8;
    constexpr int KS = D / 16;          // k-steps of the score GEMM
    constexpr int DT = D / 32;          // 32-feature tiles of the output
    __shared__ __attribute__((aligned(16))) float sc[GROUP_LDS_FLOATS];   // [G][SROW] scores / probabilities; later the merge's staging area
    __shared__ float red[2][G][4];
    __shared__ int last_flag;
    const int split = blockIdx.x, nsplit = gridDim.x, hk = blockIdx.y, b = blockIdx.z, h0 = hk * G;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int lo = krange[2 * b], hi_key = krange[2 * b + 1];
    const int a0 = lo & ~7, total = max(hi_key - a0, 0);
    const int chunk = min((((total + nsplit - 1) / nsplit) + 7) & ~7, GCHUNK);
    const int c0 = a0 + split * chunk, c1 = min(c0 + chunk, hi_key);
    const int n = max(c1 - c0, 0);
    auto slot = [&](int g) { return ws + ((int64_t)(b * Hq + h0 + g) * nsplit + split) * (D + 2); };
    auto put = [&](float* dst, float v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto get = [&](const float* src) -> float { return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto finish = [&]() {   // hand-over + merge: attn_decode_group_kernel's, word for word
        int* counters = (int*)(ws + (int64_t)gridDim.z * Hq * nsplit * (D + 2));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const int prev = __hip_atomic_fetch_add(&counters[b * Hq + h0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (prev == nsplit - 1);
            if (last_flag) __hip_atomic_store(&counters[b * Hq + h0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_flag) return;
        const float* base = ws + (int64_t)(b * Hq + h0) * nsplit * (D + 2);
        const int tot = G * nsplit * (D + 2);
        constexpr int NL = 16;
        for (int i = t; i < tot; i += 256 * NL) {
            float v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) v[u] = get(base + min(i + 256 * u, tot - 1));
#pragma unroll
            for (int u = 0; u < NL; ++u)
                if (i + 256 * u < tot) sc[i + 256 * u] = v[u];
        }
        __syncthreads();
        for (int g = t / D; g < G; g += PARTS) {
            const int d = t % D;
            const float* sg = sc + g * nsplit * (D + 2);
            float M = NEG_INF;
            for (int s = 0; s < nsplit; ++s) M = fmaxf(M, sg[s * (D + 2)]);
            float L = 0.f, o = 0.f;
            if (M != NEG_INF) {
                for (int s = 0; s < nsplit; ++s) {
                    const float wgt = __builtin_amdgcn_exp2f(sg[s * (D + 2)] - M);
                    L += wgt * sg[s * (D + 2) + 1];
                    o += wgt * sg[s * (D + 2) + 2 + d];
                }
            }
            O[b * o_bs + (h0 + g) * o_hs + d] = (bf16)(L > 0.f ? o / L : 0.f);
        }
    };
    if (n <= 0 || c1 <= lo) {
        for (int g = 0; g < G; ++g) {
            if (t < D) put(slot(g) + 2 + t, 0.f);
            if (t == 0) { put(slot(g), NEG_INF); put(slot(g) + 1, 0.f); }
        }
        finish();
        return;
    }
    // ---- operands of the first key tile and the first value batch are requested before anything is computed
    const int gq = min(l31, G - 1);                                             // lanes beyond the group repeat its last head: columns nobody reads
    const bf16* qrow = Q + b * q_bs + (h0 + gq) * q_hs + hi * 8;
    bf16x8 qf[KS];
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) qf[s4] = *(const bf16x8*)(qrow + s4 * 16);
    const bf16* kbase = Kc + b * k_bs + hk * k_hs + hi * 8 + (int64_t)c0 * k_rs;
    auto load_k = [&](bf16x8(&dst)[KS], int tile) {
        const bf16* kr = kbase + (int64_t)min(tile * 32 + l31, n - 1) * k_rs;  // clamped: valid memory, masked below
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) dst[s4] = *(const bf16x8*)(kr + s4 * 16);
    };
    const int ntile = (n + 31) >> 5;
    bf16x8 kf[KS];
    if (w < ntile) load_k(kf, w);
    constexpr int VB = 8;   // value k-steps (16 keys each) per register batch
    const int nks = (n + 15) >> 4;
    const bf16* vbase = Vt + b * vt_bs + ((int64_t)hk * D + l31) * spad + c0 + hi * 8;   // + dtile * 32 * spad, + ks * 16
    auto load_v = [&](bf16x8(&dst)[VB], int dt, int ks0) {
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int ks = min(ks0 + u, nks - 1);
            const int koff = ks * 16 + hi * 8 < n ? ks * 16 : -hi * 8;   // a half-step entirely behind the last key re-reads the chunk's first keys (valid memory; zeroed below)
            dst[u] = *(const bf16x8*)(vbase + (int64_t)dt * 32 * spad + koff);
        }
    };
    bf16x8 vf[VB];
    if (w < DT) load_v(vf, w, 0);
    const float c2 = scale * LOG2E;
    // ---- scores: S tile = 32 keys x 32 (G used) heads per MFMA chain; lane (head l31, hi) holds keys 8 q + 4 hi + e of the tile
    float mx = NEG_INF;
    for (int tile = w; tile < ntile; tile += 4) {
        if (tile != w) load_k(kf, tile);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s4], qf[s4], acc, 0, 0, 0);
        if (l31 < G) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = tile * 32 + 8 * q + 4 * hi + e;
                    const int key = c0 + i;
                    const float sv = (i < n && key >= lo) ? acc[4 * q + e] * c2 : NEG_INF;   // log2 domain
                    if (i < n) sc[l31 * SROW + i] = sv;
                    mx = fmaxf(mx, sv);
                }
        }
    }
    // chunk maximum per head: the two lane halves of a wave, then the four waves
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane < G) red[0][lane][w] = mx;
    __syncthreads();
    // ---- probabilities (every thread a strided share of the G x n scores), rounded to bf16 as the P.V product reads them; zeros up to a whole k-step
    float ls[G];
#pragma unroll
    for (int g = 0; g < G; ++g) ls[g] = 0.f;
    const int npad = nks << 4;
    for (int i = t; i < npad; i += 256) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float m = fmaxf(fmaxf(red[0][g][0], red[0][g][1]), fmaxf(red[0][g][2], red[0][g][3]));
            const float msafe = (m == NEG_INF) ? 0.f : m;
            const float pr = i < n ? rbf(__builtin_amdgcn_exp2f(sc[g * SROW + i] - msafe)) : 0.f;
            sc[g * SROW + i] = pr;
            ls[g] += pr;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float v = wave_sum(ls[g]);
        if (lane == 0) red[1][g][w] = v;
    }
    __syncthreads();
    // ---- output: wave w owns feature tiles w, w + 4, ...; O tile = 32 features x 32 (G used) heads, k = keys
    for (int dt = w; dt < DT; dt += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int ks0 = 0; ks0 < nks; ks0 += VB) {
            if (ks0 || dt != w) load_v(vf, dt, ks0);
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int ks = ks0 + u;
                if (ks < nks) {   // block-uniform
                    const int k0 = ks * 16 + hi * 8;
                    bf16x8 pv, vv = vf[u];
                    const float* pp = &sc[gq * SROW + k0];
                    const f32x4 p0 = *(const f32x4*)pp, p1 = *(const f32x4*)(pp + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pv[e] = (bf16)(e < 4 ? p0[e & 3] : p1[e & 3]);
                        if (k0 + e >= n) vv[e] = (bf16)0.f;   // what lies behind the last key may be anything (0 x NaN)
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vv, pv, acc, 0, 0, 0);
                }
            }
        }
        if (l31 < G) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) put(slot(l31) + 2 + dt * 32 + 8 * q + 4 * hi + e, acc[4 * q + e]);
        }
    }
    if (t < G) {
        const float m = fmaxf(fmaxf(red[0][t][0], red[0][t][1]), fmaxf(red[0][t][2], red[0][t][3]));
        put(slot(t), m);
        put(slot(t) + 1, red[1][t][0] + red[1][t][1] + red[1][t][2] + red[1][t][3]);
    }
    finish();
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

extern "C" int afk_attn_decode_workspace_floats(int B, int Hq, int D, int nsplit) { return B * Hq * nsplit * (D + 2) + B * Hq; }   // + arrival counters (afk_attn_decode_fused)

static std::atomic<int> g_group_mode{[] { const char* e = getenv("AFK_ATTN_DECODE_GROUP"); return e ? atoi(e) : -1; }()};
extern "C" int afk_attn_decode_set_group(int mode) {
    AFK_REQUIRE(mode >= -1 && mode <= 3,
                "afk_attn_decode_set_group: mode %d (-1 default: per-head, matrix-pipe group form for large launches; 0 per-head; 1 / 2 VALU group form from 128 blocks on / whenever it fits; 3 matrix-pipe group form)", mode);
    g_group_mode.store(mode, std::memory_order_relaxed);
    return AFK_OK;
}

static int attn_decode_impl(bool fused, const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                            const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                            int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    AFK_REQUIRE(Q && Kc && Vt && O && krange && workspace, "afk_attn_decode: null pointer");
    AFK_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && (D == 64 || D == 128) && nsplit >= 1 && nsplit <= 64, "afk_attn_decode: bad shape");
    AFK_REQUIRE(!fused || nsplit * (D + 2) <= MAXCHUNK, "afk_attn_decode_fused: nsplit * (D + 2) <= %d (the merge stages the partials in the score buffer)", MAXCHUNK);
    AFK_REQUIRE(q_bs % 8 == 0 && q_hs % 8 == 0 && k_bs % 8 == 0 && k_rs % 8 == 0 && k_hs % 8 == 0 && vt_bs % 8 == 0 && spad % 8 == 0,
                "afk_attn_decode: strides must keep 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nsplit, (unsigned)Hq, (unsigned)B);
    static const int sync = [] { const char* e = getenv("AFK_ATTN_DECODE_SYNC"); return e ? atoi(e) : 1; }();
    // group form (one block per (sample, KV head, chunk) for all Hq / Hkv query heads).  OFF by default (mode 0): built for the batched decode step on the
    // argument that the G per-head blocks re-fetch the same K / V chunk, bit-identical to the per-head form - and MEASURED 1-2 % SLOWER on the B = 8 step
    // (4.50 / 4.54 vs 4.45 / 4.46 ms, alternating runs on one box, profiles/r05_kernel_ab.md): with nsplit = 8 the blocks of one chunk index already share
    // an XCD (linear block id mod 8 = chunk index), so the L2 serves the G - 1 re-reads, and 256 blocks of 4 waves hide less latency than 1 792.
    // afk_attn_decode_set_group / AFK_ATTN_DECODE_GROUP: -1 = default (below), 0 = per-head, 1 = VALU group form from 128 blocks on, 2 = whenever its limits hold, 3 = matrix-pipe group form
    const int group_mode = g_group_mode.load(std::memory_order_relaxed);
    const int G = Hq / Hkv;
    const bool group_fits = fused && sync == 1 && G >= 2 && (int64_t)spad <= (int64_t)nsplit * GCHUNK && G * nsplit * (D + 2) <= GROUP_LDS_FLOATS &&
                            (G == 2 || G == 4 || G == 7 || G == 8);
    // matrix-pipe group form (round 6): on request, or by itself (mode -1, the default) when the per-head form would bring more than four blocks per CU - measured at the
    // AF3-7B geometry, 800 keys, us per launch: B = 8  21.1 per-head / 17.6 here (1 500 keys: 27.4 / 21.7), B = 4  15.7 / 17.0, B = 2  13.0 / 15.6
    const bool gmma_auto = group_mode == -1 && (int64_t)nsplit * Hq * B > 4 * 256;
    if (group_fits && (group_mode == 3 || gmma_auto) && G <= 8 && spad % 8 == 0) {
        const dim3 ggrid((unsigned)nsplit, (unsigned)Hkv, (unsigned)B);
#define AFK_ADM(DD, GG)                                                                                                                                   \
    hipLaunchKernelGGL((attn_decode_gmma_kernel<DD, GG>), ggrid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs, k_hs,         \
                       (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace, (bf16*)O, o_bs, o_hs)
#define AFK_ADM_D(DD)                                      \
    do {                                                   \
        if (G == 2) AFK_ADM(DD, 2);                        \
        else if (G == 4) AFK_ADM(DD, 4);                   \
        else if (G == 7) AFK_ADM(DD, 7);                   \
        else AFK_ADM(DD, 8);                               \
    } while (0)
        if (D == 128) AFK_ADM_D(128); else AFK_ADM_D(64);
#undef AFK_ADM_D
#undef AFK_ADM
        AFK_LAUNCH_CHECK("afk_attn_decode_fused (matrix-pipe group form)");
        return AFK_OK;
    }
    if (group_fits && group_mode > 0 && group_mode != 3 && (group_mode == 2 || B * Hkv * nsplit >= GROUP_MIN_BLOCKS)) {
        const dim3 ggrid((unsigned)nsplit, (unsigned)Hkv, (unsigned)B);
#define AFK_ADG(DD, GG)                                                                                                                                   \
    hipLaunchKernelGGL((attn_decode_group_kernel<DD, GG>), ggrid, dim3(256), 0, st, (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs, k_hs,        \
                       (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace, (bf16*)O, o_bs, o_hs)
#define AFK_ADG_D(DD)                                      \
    do {                                                   \
        if (G == 2) AFK_ADG(DD, 2);                        \
        else if (G == 4) AFK_ADG(DD, 4);                   \
        else if (G == 7) AFK_ADG(DD, 7);                   \
        else AFK_ADG(DD, 8);                               \
    } while (0)
        if (D == 128) AFK_ADG_D(128); else AFK_ADG_D(64);
#undef AFK_ADG_D
#undef AFK_ADG
        AFK_LAUNCH_CHECK("afk_attn_decode_fused (group form)");
        return AFK_OK;
    }
    // AFK_ATTN_DECODE_WPE (A/B knob): 0 = by block count, 3 = the compiler's allocation (130 VGPRs, 3 blocks per CU), 4 / 5 / 6 = that many waves per SIMD with
    // 128-key register batches, 7 = 64-key batches in 66 VGPRs (7 blocks per CU), 8 = 64-key batches in 76.  Round 6 (tools/bench_decode_chain_batched.py, us per launch,
    // 800 keys): B = 8  29.1 / 26.3 / 28.9 / 35.8 / 21.3 / 22.7 for 3 / 4 / 5 / 6 / 7 / 8;  B = 4  19.7 / 16.2 / 17.4 / - / 15.9 / 15.1;  B = 1  10.9 / 11.7 / 12.1 / - / 11.5 / 11.2
    // -> more than three blocks per CU: 7; otherwise the compiler's allocation.
    static const int wpe_env = [] { const char* e = getenv("AFK_ATTN_DECODE_WPE"); return e ? atoi(e) : 0; }();
    const int wpe = wpe_env ? wpe_env : ((int64_t)nsplit * Hq * B > 3 * 256 ? 7 : 3);
#define AFK_AD_ARGS (const bf16*)Q, q_bs, q_hs, (const bf16*)Kc, k_bs, k_rs, k_hs, (const bf16*)Vt, vt_bs, spad, krange, Hq, Hkv, scale, workspace, (bf16*)O, o_bs, o_hs
#define AFK_AD(DD)                                                                                                                       \
    do {                                                                                                                                 \
        if (fused && sync == 1) {                                                                                                        \
            if (wpe == 4) hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1, 4>), grid, dim3(256), 0, st, AFK_AD_ARGS);               \
            else if (wpe == 5) hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1, 5>), grid, dim3(256), 0, st, AFK_AD_ARGS);          \
            else if (wpe == 6) hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1, 6>), grid, dim3(256), 0, st, AFK_AD_ARGS);          \
            else if (wpe == 7) hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1, 7, 64>), grid, dim3(256), 0, st, AFK_AD_ARGS);      \
            else if (wpe == 8) hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1, 5, 64>), grid, dim3(256), 0, st, AFK_AD_ARGS);      \
            else hipLaunchKernelGGL((attn_decode_split_kernel<DD, 1>), grid, dim3(256), 0, st, AFK_AD_ARGS);                           \
        } else if (fused) {                                                                                                              \
            hipLaunchKernelGGL((attn_decode_split_kernel<DD, 2>), grid, dim3(256), 0, st, AFK_AD_ARGS);                                  \
        } else {                                                                                                                         \
            hipLaunchKernelGGL((attn_decode_split_kernel<DD, 0>), grid, dim3(256), 0, st, AFK_AD_ARGS);                                  \
            hipLaunchKernelGGL(attn_decode_combine_kernel<DD>, dim3((unsigned)Hq, (unsigned)B), dim3(DD), 0, st, workspace, nsplit, (bf16*)O, o_bs, \
                               o_hs, Hq);                                                                                                \
        }                                                                                                                                \
    } while (0)
    if (D == 128) AFK_AD(128); else AFK_AD(64);
#undef AFK_AD
#undef AFK_AD_ARGS
    AFK_LAUNCH_CHECK("afk_attn_decode");
    return AFK_OK;
}

extern "C" int afk_attn_decode(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                               const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                               int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    return attn_decode_impl(false, Q, q_bs, q_hs, Kc, k_bs, k_rs, k_hs, Vt, vt_bs, spad, O, o_bs, o_hs, krange, B, Hq, Hkv, D, scale, nsplit, workspace, stream);
}

#ifdef AFK_PROBES
extern "C" int afk_probe_attn_decode_stamps(long long* device_buffer) {   // [blocks][8] int64, or null to switch the stamps off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &device_buffer, sizeof(device_buffer)) == hipSuccess ? AFK_OK : AFK_ERR_LAUNCH;
}
#endif

// one launch: the last chunk block of every (batch, head) pair merges the chunks.  workspace: afk_attn_decode_workspace_floats(...) floats whose LAST B * Hq words are the
// per-pair arrival counters - they must read ZERO before the first call (the kernel leaves them at zero)
extern "C" int afk_attn_decode_fused(const void* Q, int64_t q_bs, int64_t q_hs, const void* Kc, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                                     const void* Vt, int64_t vt_bs, int spad, void* O, int64_t o_bs, int64_t o_hs, const int* krange, int B,
                                     int Hq, int Hkv, int D, float scale, int nsplit, float* workspace, void* stream) {
    return attn_decode_impl(true, Q, q_bs, q_hs, Kc, k_bs, k_rs, k_hs, Vt, vt_bs, spad, O, o_bs, o_hs, krange, B, Hq, Hkv, D, scale, nsplit, workspace, stream);
}
