// bf16 NT GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . B[N,K]^T )   (fp32 accumulate on MFMA)
//
// This one kernel carries every dense contraction on the Audio Flamingo 3 training path
// (oracle call sites: nn.Linear in modeling_audioflamingo3.py:109-115,209-210,427-433,576 and
// modeling_qwen2.py:40-42,189-192; conv stem :328-329 via im2col).  nn.Linear stores W as [N,K]
// row-major, so the forward is NT with both operands K-contiguous - exactly the MFMA fragment order.
// The backward contractions are brought to the same NT form by the host (W^T shadow copies and
// transposed activations, see ops.py), so one tuned kernel serves fwd, dgrad and wgrad.
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles,
// global->LDS by direct LDS-DMA (global_load_lds_dwordx4), two LDS stages (64 KiB => 2 blocks/CU),
// one barrier per K-step.  LDS rows are 128 B (64 bf16); the 16-B chunk c of row r is stored at
// chunk position c ^ ((r>>1)&7): the LDS-DMA destination is lane-linear, so the permutation is applied
// to the per-lane SOURCE address and again on the ds_read_b128 side (guide rule 21).  With the
// 32x32x16 fragment (lane -> row l&31, chunk 2s+(l>>5)) every ds_read_b128 lane group then covers all
// 16 slots of the 256-B bank row exactly once.
// Operands are fed swapped (mfma(Bfrag, Afrag)) so that each lane ends up with 4 consecutive N for one
// M row: 8-byte bf16 stores, and bias / residual loads of the same shape.
// Workgroup ids are remapped XCD-contiguously and walk the tile space in 8-row groups so that the
// 64 tiles resident on one XCD share 8 A panels and 8 B panels through that XCD's L2.
#include "gemm_common.h"
#include <vector>
#include <mutex>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROWB = BK * 2;                   // 128 bytes per LDS row
constexpr int STAGE_BYTES = (BM + BN) * ROWB;  // 32 KiB
constexpr int NSTAGE = 2;

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_k128(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    gemm_tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane LDS-DMA source pointers: unit u covers LDS rows [8u, 8u+8) of the tile, 1 KiB
    const bf16* a_src[4];
    const bf16* b_src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int unit = wave + 4 * u;
        const int rl = unit * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        const int ra = min(m0 + rl, p.M - 1);
        const int rb = min(n0 + rl, p.N - 1);
        a_src[u] = p.A + (int64_t)ra * p.lda + chunk * 8;
        b_src[u] = p.B + (int64_t)rb * p.ldb + chunk * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + BM * ROWB;
        const int koff = kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int unit = wave + 4 * u;
            __builtin_amdgcn_global_load_lds((gbl_void*)(a_src[u] + koff), (lds_void*)(sa + unit * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int unit = wave + 4 * u;
            __builtin_amdgcn_global_load_lds((gbl_void*)(b_src[u] + koff), (lds_void*)(sb + unit * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes) inside a tile image; swizzle term is lane-constant
    const int swz_l = (lane >> 1) & 7;
    int a_off[4], b_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = ((2 * s + hi) ^ swz_l) << 4;
        a_off[s] = (wm * 64 + l31) * ROWB + c;
        b_off[s] = (wn * 64 + l31) * ROWB + c;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int kt0 = 0, nkt = p.K / BK;
    if (p.splits > 1) {  // split-K: this block reduces K-tiles [kt0, nkt) of the tile
        const int all = nkt, sp = blockIdx.y;
        kt0 = (int)((int64_t)all * sp / p.splits);
        nkt = (int)((int64_t)all * (sp + 1) / p.splits);
    }
    if (kt0 < nkt) stage(kt0 & 1, kt0);
    __syncthreads();
    for (int kt = kt0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * STAGE_BYTES;
        const char* sb = sa + BM * ROWB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(sa + a_off[s] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(sb + b_off[s] + j * 32 * ROWB);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue (gemm_common.h): every lane takes part in the lane-half exchange of the wide form, so no early exit per lane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gemm_store_block32_body<-1>(p, m0 + wm * 64 + i * 32 + l31, n0 + wn * 64 + j * 32, hi, acc[i][j]);
}

// measured on the AF3-7B decode step (ms/token): M = 1 3.85 (MFMA split-K tiles: 6.5); M = 2 6.3, M = 4 6.4, M = 8 7.6 against 5.3-5.7 on
// MFMA split-K tiles (per launch 34 us at M = 2 vs 22 us at M = 1) - the path is taken for a single row only; the kernel keeps its
// multi-row instantiations for the unit tests and for a later look at why the second row costs 55 %
#define AFK_GEMV_MAX_M 1
template <int MB, int R>
__global__ __launch_bounds__(256) void gemv_nt_bf16_kernel(GemmArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * R;
    if (n0 >= p.N) return;
    const int kb_all = (p.K + 511) >> 9;
    const int kb0 = (int)((int64_t)kb_all * blockIdx.y / p.splits), kb1 = (int)((int64_t)kb_all * (blockIdx.y + 1) / p.splits);
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    const bf16* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = p.B + (int64_t)min(n0 + r, p.N - 1) * p.ldb;
    for (int kb = kb0; kb < kb1; ++kb) {
        const int k = (kb << 9) + lane * 8;
        if (k < p.K) {  // K % 8 == 0: a lane's 8 elements are all in or all out
            bf16x8 wv[R], xv[MB];
#pragma unroll
            for (int r = 0; r < R; ++r) wv[r] = __builtin_nontemporal_load((const bf16x8*)(wrow[r] + k));  // streamed once: keep L2 for x
#pragma unroll
            for (int m = 0; m < MB; ++m) xv[m] = *(const bf16x8*)(p.A + (int64_t)min(m, p.M - 1) * p.lda + k);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2 a = {wv[r][2 * e], wv[r][2 * e + 1]}, b = {xv[m][2 * e], xv[m][2 * e + 1]};
                        acc[r][m] = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc[r][m], false);
                    }
        }
    }
    // cross-lane reduction of the V = R*MB per-lane partial sums as a reduce-scatter: at stride s a lane hands the half it does not
    // keep to lane^s (V-1 exchanges instead of 6V); after log2(V) halvings lane l holds the complete sum number l >> (6 - log2 V)
    constexpr int V = R * MB;
    static_assert((V & (V - 1)) == 0 && V <= 64, "R*MB must be a power of two <= 64");
    float* v = &acc[0][0];
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
                    const float lo = v[i], hi_ = v[i + h];
                    v[i] = (up ? hi_ : lo) + __shfl_xor(up ? lo : hi_, st, 64);
                }
            n = h;
        } else {
            v[0] += __shfl_xor(v[0], st, 64);
        }
    }
    constexpr int LOGV = V == 64 ? 6 : V == 32 ? 5 : V == 16 ? 4 : V == 8 ? 3 : V == 4 ? 2 : V == 2 ? 1 : 0;
    const int idx = lane >> (6 - LOGV);
    const int r = idx / MB, m = idx % MB;
    if ((lane & ((64 >> LOGV) - 1)) == 0 && n0 + r < p.N && m < p.M) p.ws[((int64_t)blockIdx.y * p.M + m) * p.N + n0 + r] = v[0];
}

// split-K second pass: fixed-order sum of the partials (bit-deterministic), then the same fused epilogue as the one-pass kernels
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs p) {
    const int n4 = p.N >> 2;
    const int64_t total = (int64_t)p.M * n4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(t / n4), n = (int)(t % n4) * 4;
        const float* src = p.ws + (int64_t)m * p.N + n;
        f32x4 acc = *(const f32x4*)src;
        for (int sp = 1; sp < p.splits; ++sp) {
            const f32x4 x = *(const f32x4*)(src + (int64_t)sp * p.M * p.N);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += x[e];
        }
        float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        gemm_epilogue_store4(p, m, n, v);
    }
}

// ------------------------------------------------------------------ profiling (HIP events per launch)
struct ProfState {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> pool;  // pairs
    size_t used = 0;
    double flops = 0.0;
    int64_t launches = 0;
    struct Rec { int M, N, K, variant; };
    std::vector<Rec> recs;
};
ProfState g_prof;
int g_variant = 0;  // 0 auto, 1 force 128x128, 2 force 256x256 (8-wave ping-pong), 3 force 256x256 (4-wave, 128x128 per wave)
[[maybe_unused]] int g_256_impl = 2; // which 256x256 NT kernel "auto" uses: 2 = 8-wave ping-pong (gemm256.hip), 3 = 4-wave (gemm256w4.hip), 10 = free-running 8-wave BK=32
                    // (gemm256f8.hip); env AFK_GEMM256 = pp | w4 | f8
int g_gm = 0;       // rasterization group height override (0 = default 8)
int g_wide = 1;     // 16-byte epilogue form allowed (afk_gemm_set_variant bit 4 clears it: A/B experiments)

hipEvent_t prof_next_event() {
    if (g_prof.used == g_prof.pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_prof.pool.push_back(e);
    }
    return g_prof.pool[g_prof.used++];
}

}  // namespace

extern "C" int afk_gemm_set_variant(int v) {
    // v = base + 256*gm : base 0 auto, 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel; bit 4 (16) = 8-byte epilogue stores (A/B knob, same results);
    // gm bits 0..5 = rasterization group height.  Everything else selects a timing probe or a rejected schedule and exists in
    // -DAFK_PROBES builds only (gm bit 6: no epilogue = WRONG results, bit 7: raw dispatch order; base 3..13: gemm256w4 / f8 / p.hip).
    const int gm = (v >> 8) & 255, base = v & 15;
#ifdef AFK_PROBES
    AFK_REQUIRE(base >= 0 && base <= 14, "afk_gemm_set_variant: 14 = persistent tile loop with the next tile's prologue issued ahead of the epilogue stores (round-4 probe), 13 = 256x256 ping-pong as a persistent tile loop, 0 auto, 1 = 128x128, 2 = 256x256 8-wave ping-pong, 3 = 256x256 4-wave (4, 5: its timing probes), 6 = 256x256 8-wave free-running BK=64 (7, 8, 9: probes), 10 = 8-wave free-running BK=32 ring-10 (11: probe)");
#else
    AFK_REQUIRE(base >= 0 && base <= 2, "afk_gemm_set_variant: variant %d is a probe / rejected schedule; this libafk.so was built without -DAFK_PROBES (make PROBES=1)", base);
    AFK_REQUIRE((gm & 0xc0) == 0, "afk_gemm_set_variant: gm bits 6 / 7 are timing probes (wrong results); this libafk.so was built without -DAFK_PROBES");
#endif
    g_gm = gm;
    g_wide = (v & 16) ? 0 : 1;
    g_variant = base;
    return AFK_OK;
}

extern "C" int afk_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return AFK_OK;
}

extern "C" int afk_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.used = 0;
    g_prof.flops = 0.0;
    g_prof.launches = 0;
    g_prof.recs.clear();
    return AFK_OK;
}

extern "C" int afk_prof_collect(double* total_ms, double* total_flops, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        if (hipEventSynchronize(g_prof.pool[i + 1]) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "afk_prof_collect: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.pool[i], g_prof.pool[i + 1]) != hipSuccess)
            return afk_set_error(AFK_ERR_LAUNCH, "afk_prof_collect: elapsed failed");
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = g_prof.flops;
    if (launches) *launches = g_prof.launches;
    return AFK_OK;
}

struct RopeEpi { const void *cos_t, *sin_t; const int* pos; int S, cols; const void *cos_lanes, *sin_lanes; };   // afk_gemm_nt_bf16_rope

static int gemm_impl(int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                     int res_mod, void* preact_out, float alpha, int flags, void* stream, int splits = 1, void* workspace = nullptr,
                     const RopeEpi* rope = nullptr) {
    AFK_REQUIRE(splits >= 1 && splits <= 64 && (splits == 1 || workspace), "afk_gemm_*_splitk: 1..64 splits, workspace required");
    AFK_REQUIRE(((flags & AFK_GEMM_ROPE) != 0) == (rope != nullptr), "afk_gemm: AFK_GEMM_ROPE belongs to afk_gemm_nt_bf16_rope");
    AFK_REQUIRE(splits == 1 || splits <= (K + BK - 1) / BK, "afk_gemm_*_splitk: more splits than K tiles");
    const bool gemv = workspace != nullptr && M <= AFK_GEMV_MAX_M && !trans_a && !trans_b;  // skinny-M weight streaming (decode)
    AFK_REQUIRE(!gemv || splits <= (K + 511) / 512, "afk_gemm_nt_bf16_splitk: M <= 16 streams K in blocks of 512: at most ceil(K/512) splits");
    AFK_REQUIRE(A && B && C, "afk_gemm_nt_bf16: null operand");
    AFK_REQUIRE(M > 0 && N > 0 && K > 0, "afk_gemm_nt_bf16: bad shape %d %d %d", M, N, K);
    AFK_REQUIRE(!trans_a || trans_b, "afk_gemm_bf16: A^T with k-contiguous B is not implemented (NT, NN, TN are)");
    AFK_REQUIRE(trans_a || K % BK == 0, "afk_gemm_nt_bf16: K=%d must be a multiple of %d (pad the operand)", K, BK);
    AFK_REQUIRE(!trans_b || (N % 8 == 0 && N >= 8), "afk_gemm_bf16: N=%d must be a multiple of 8 for a reduction-major B", N);
    AFK_REQUIRE(!trans_a || (M % 8 == 0 && M >= 8), "afk_gemm_bf16: M=%d must be a multiple of 8 for a reduction-major A", M);
    AFK_REQUIRE(N % 4 == 0, "afk_gemm_nt_bf16: N=%d must be a multiple of 4", N);
    AFK_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "afk_gemm_nt_bf16: leading dims must keep 16-B (A,B) / 8-B (C) alignment");
    AFK_REQUIRE(!(flags & AFK_GEMM_BIAS) || bias, "afk_gemm_nt_bf16: BIAS flag without bias");
    AFK_REQUIRE(!(flags & AFK_GEMM_RESIDUAL) || (residual && ldr % 4 == 0), "afk_gemm_nt_bf16: RESIDUAL flag without residual");
    AFK_REQUIRE(!(flags & AFK_GEMM_SWIGLU_FWD) || (flags == AFK_GEMM_SWIGLU_FWD && !trans_a && !trans_b && preact_out && N % 256 == 0 && splits == 1 &&
                                                     ldc % 8 == 0 && (uintptr_t)C % 16 == 0 && (uintptr_t)preact_out % 16 == 0),
                "afk_gemm_nt_bf16: SWIGLU_FWD needs the NT form, preact_out, N = 2I with I %% 128 == 0 and no other flag");
    AFK_REQUIRE(!(flags & AFK_GEMM_SWIGLU_BWD) || (residual && ldr % 4 == 0 && ldc >= 2 * (int64_t)N && flags == AFK_GEMM_SWIGLU_BWD && splits == 1),
                "afk_gemm: SWIGLU_BWD needs the gate|up tensor as `residual`, a [M, 2N] output and no other epilogue flag");
    AFK_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0), "afk_gemm_nt_bf16: misaligned pointer");
    GemmArgs p;
    p.A = (const bf16*)A;
    p.B = (const bf16*)B;
    p.C = C;
    p.C2 = preact_out;
    p.bias = (const bf16*)bias;
    p.R = (const bf16*)residual;
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = ldc;
    p.ldr = ldr;
    p.M = M;
    p.N = N;
    p.K = K;
    p.flags = flags;
    p.res_mod = res_mod;
    p.alpha = alpha;
    p.gm = g_gm;
    p.splits = splits;
    p.ws = (float*)workspace;
    p.rope_cos = rope ? (const bf16*)rope->cos_t : nullptr;
    p.rope_sin = rope ? (const bf16*)rope->sin_t : nullptr;
    p.rope_pos = rope ? rope->pos : nullptr;
    p.rope_S = rope ? rope->S : 1;
    p.rope_cols = rope ? rope->cols : 0;
    // the lane-major copies apply when a 32-row accumulator block never straddles two samples and positions are the row index
    const bool lanes_ok = rope && rope->cos_lanes && rope->sin_lanes && !rope->pos && rope->S % 32 == 0;
    p.rope_cos_lanes = lanes_ok ? (const bf16*)rope->cos_lanes : nullptr;
    p.rope_sin_lanes = lanes_ok ? (const bf16*)rope->sin_lanes : nullptr;
    {
        const bool f32 = (flags & AFK_GEMM_OUT_F32) != 0;
        bool w = N % 8 == 0 && ldc % 8 == 0 && (uintptr_t)C % (f32 ? 32 : 16) == 0;
        if (flags & AFK_GEMM_BIAS) w = w && (uintptr_t)bias % 16 == 0;
        if (flags & (AFK_GEMM_RESIDUAL | AFK_GEMM_SWIGLU_BWD)) w = w && ldr % 8 == 0 && (uintptr_t)residual % 16 == 0;
        if (preact_out) w = w && (uintptr_t)preact_out % 16 == 0;
        p.wide = w && g_wide ? 1 : 0;
    }
    // variant choice: the 256x256 ping-pong kernel halves L2->LDS traffic per flop but needs enough tiles to fill 256 CUs
    const int64_t tiles256 = afk_cdiv(M, 256) * afk_cdiv(N, 256);
    const bool use256 = !gemv && (trans_b || (flags & AFK_GEMM_SWIGLU_FWD) || (splits == 1 && (g_variant >= 2 || (g_variant == 0 && tiles256 >= 192))));
    // the RoPE epilogue exists in the 256 x 256 ping-pong kernel's 16-byte form only: say so instead of running an instantiation without it
    if (rope && (!use256 || !p.wide || trans_a || trans_b || splits != 1))
        return afk_set_error(AFK_ERR_UNSUPPORTED, "afk_gemm_nt_bf16_rope: needs the 256 x 256 NT kernel (>= 192 tiles) in its 16-byte epilogue form; use "
                                                  "afk_gemm_nt_bf16 + afk_rope_inplace for this shape");
#ifdef AFK_PROBES
    static const int env_impl = [] {
        const char* e = getenv("AFK_GEMM256");
        return (e && e[0] == 'p' && e[1] == 'e') ? 13 : (e && e[0] == 'p') ? 2 : (e && e[0] == 'w') ? 3 : (e && e[0] == 'f') ? 10 : 0;  // pp | persist | w4 | f8
    }();
    // the fused SwiGLU forward exists in the ping-pong kernel only: it never takes a probe schedule
    const int impl256 = (flags & (AFK_GEMM_SWIGLU_FWD | AFK_GEMM_ROPE)) ? 2 : g_variant >= 2 ? g_variant : (env_impl ? env_impl : g_256_impl);   // 2 pp | 3..9 w4 family | 10, 11 f8
    const bool w4 = use256 && !trans_b && impl256 >= 3 && impl256 <= 9;
    const bool f8 = use256 && !trans_b && impl256 >= 10 && impl256 <= 12;
    const bool persist = use256 && !trans_b && impl256 == 13;
    const bool persist_q = use256 && !trans_b && impl256 == 14;
#endif
    // SWIGLU_FWD is implemented by gemm_nt_bf16_k256<AFK_GEMM_SWIGLU_FWD> alone, which needs the 16-byte epilogue form: refuse instead of
    // falling through to an instantiation that would leave preact_out unwritten
    if ((flags & AFK_GEMM_SWIGLU_FWD) && !p.wide)
        return afk_set_error(AFK_ERR_UNSUPPORTED, "afk_gemm_nt_bf16: SWIGLU_FWD needs the 16-byte epilogue form (aligned C / preact_out, ldc %% 8 == 0, "
                                                  "afk_gemm_set_variant bit 4 clear)");
    p.ntm = (int)afk_cdiv(M, use256 ? 256 : BM);
    p.ntn = (int)afk_cdiv(N, use256 ? 256 : BN);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_nt_bf16_k128, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t nwg = (int64_t)p.ntm * p.ntn;
    AFK_REQUIRE(nwg < (1ll << 31), "afk_gemm_nt_bf16: grid too large");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool prof = false;
    {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        prof = g_prof.on;
        if (prof) {
            e0 = prof_next_event();
            e1 = prof_next_event();
            g_prof.flops += 2.0 * (double)M * (double)N * (double)K;
            g_prof.launches += 1;
            g_prof.recs.push_back({M, N, K, trans_b ? (trans_a ? 4 : 3) : (use256 ? 2 : 1)});
        }
    }
    afk_count(gemv ? AFK_CNT_GEMV : trans_b ? (trans_a ? AFK_CNT_GEMM_TN256 : AFK_CNT_GEMM_NN256) : use256 ? AFK_CNT_GEMM_NT256 : AFK_CNT_GEMM_NT128);
    if (splits > 1) afk_count(AFK_CNT_GEMM_SPLITK);
    if (prof) hipEventRecord(e0, st);
    if (gemv) {
        const dim3 grid((unsigned)afk_cdiv(N, 32), (unsigned)splits);
#define AFK_GEMV(MB_, R_) hipLaunchKernelGGL((gemv_nt_bf16_kernel<MB_, R_>), grid, dim3(256), 0, st, p)
        AFK_GEMV(1, 8);
#undef AFK_GEMV
        int g = (int)afk_cdiv((int64_t)M * (N / 4), 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p);
    } else if (trans_b) {
        if (int e = afk_launch_gemm256t(p, trans_a, st)) return e;
        if (splits > 1) {
            int g = (int)afk_cdiv((int64_t)M * (N / 4), 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p);
        }
    } else if (use256) {
#ifdef AFK_PROBES
        if (int e = persist_q ? afk_launch_gemm256q(p, st) : persist ? afk_launch_gemm256p(p, st) : f8 ? afk_launch_gemm256f8(p, impl256 - 10, st) : w4 ? afk_launch_gemm256w4(p, impl256 - 3, st) : afk_launch_gemm256(p, st)) return e;
#else
        if (int e = afk_launch_gemm256(p, st)) return e;
#endif
    } else {
        hipLaunchKernelGGL(gemm_nt_bf16_k128, dim3((unsigned)nwg, (unsigned)splits), dim3(256), NSTAGE * STAGE_BYTES, st, p);
        if (splits > 1) {
            int g = (int)afk_cdiv((int64_t)M * (N / 4), 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p);
        }
    }
    if (prof) hipEventRecord(e1, st);
    AFK_LAUNCH_CHECK("afk_gemm_nt_bf16");
    return AFK_OK;
}

extern "C" int afk_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                                int res_mod, void* preact_out, float alpha, int flags, void* stream) {
    return gemm_impl(0, 0, A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, res_mod, preact_out, alpha, flags, stream);
}

extern "C" int afk_gemm_nt_bf16_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                                     const void* cos_t, const void* sin_t, const int* pos, int S, int rope_cols, const void* cos_lanes, const void* sin_lanes,
                                     void* stream) {
    AFK_REQUIRE(cos_t && sin_t && (uintptr_t)cos_t % 16 == 0 && (uintptr_t)sin_t % 16 == 0, "afk_gemm_nt_bf16_rope: cos / sin tables (16-byte aligned) are required");
    AFK_REQUIRE(N % 256 == 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % 256 == 0 && S > 0,
                "afk_gemm_nt_bf16_rope: N and rope_cols must be multiples of 256 (two 128-column heads per tile), S > 0");
    AFK_REQUIRE((cos_lanes == nullptr) == (sin_lanes == nullptr) && (!cos_lanes || ((uintptr_t)cos_lanes % 16 == 0 && (uintptr_t)sin_lanes % 16 == 0)),
                "afk_gemm_nt_bf16_rope: the lane-major tables come as a 16-byte aligned pair");
    const RopeEpi r = {cos_t, sin_t, pos, S, rope_cols, cos_lanes, sin_lanes};
    return gemm_impl(0, 0, A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, 0, nullptr, 1.f, AFK_GEMM_ROPE | (bias ? AFK_GEMM_BIAS : 0), stream, 1, nullptr, &r);
}

extern "C" int afk_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                                       const void* bias, const void* residual, int64_t ldr, int res_mod, void* preact_out, float alpha,
                                       int flags, int splits, void* workspace, void* stream) {
    return gemm_impl(0, 0, A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, res_mod, preact_out, alpha, flags, stream, splits, workspace);
}

// first pass only (decode glue, csrc/decode_glue.hip): fp32 partials ws[splits][M][N] of x[M,K] . W[N,K]^T, M <= AFK_GEMV_MAX_M
extern "C" int afk_gemv_partials(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int splits, float* workspace,
                                 void* stream) {
    AFK_REQUIRE(A && B && workspace && M >= 1 && M <= AFK_GEMV_MAX_M && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0,
                "afk_gemv_partials: bad args (M <= %d, K %% 8 == 0)", AFK_GEMV_MAX_M);
    AFK_REQUIRE(splits >= 1 && splits <= (K + 511) / 512 && splits <= 64, "afk_gemv_partials: 1 <= splits <= ceil(K / 512)");
    GemmArgs p = {};
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.splits = splits; p.ws = workspace;
    const dim3 grid((unsigned)afk_cdiv(N, 32), (unsigned)splits);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((gemv_nt_bf16_kernel<1, 8>), grid, dim3(256), 0, st, p);
    AFK_LAUNCH_CHECK("afk_gemv_partials");
    return AFK_OK;
}

extern "C" int afk_gemm_bf16_splitk(int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                    int M, int N, int K, const void* bias, const void* residual, int64_t ldr, int res_mod, void* preact_out,
                                    float alpha, int flags, int splits, void* workspace, void* stream) {
    return gemm_impl(trans_a, trans_b, A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, res_mod, preact_out, alpha, flags, stream, splits,
                     workspace);
}

extern "C" int afk_gemm_bf16(int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                             int64_t ldc, int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                             int res_mod, void* preact_out, float alpha, int flags, void* stream) {
    return gemm_impl(trans_a, trans_b, A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, res_mod, preact_out, alpha, flags, stream);
}

// per-launch records of the profiling window as CSV: M,N,K,variant(1=nt128,2=nt256,3=nn256,4=tn256),ms
extern "C" int afk_prof_dump(const char* host_path) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    FILE* f = fopen(host_path, "w");
    if (!f) return afk_set_error(AFK_ERR_ARG, "afk_prof_dump: cannot open %s", host_path);
    fprintf(f, "M,N,K,variant,ms\n");
    for (size_t i = 0; i < g_prof.recs.size() && 2 * i + 1 < g_prof.used; ++i) {
        float ms = 0.f;
        hipEventSynchronize(g_prof.pool[2 * i + 1]);
        hipEventElapsedTime(&ms, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]);
        fprintf(f, "%d,%d,%d,%d,%.6f\n", g_prof.recs[i].M, g_prof.recs[i].N, g_prof.recs[i].K, g_prof.recs[i].variant, ms);
    }
    fclose(f);
    return AFK_OK;
}
