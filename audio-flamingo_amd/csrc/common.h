// Shared device/host helpers for the afk (Audio-Flamingo kernels) C-ABI library.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, MFMA bf16 32x32x16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>
#include <utility>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define AFK_OK 0
#define AFK_ERR_ARG (-1)
#define AFK_ERR_LAUNCH (-2)
#define AFK_ERR_UNSUPPORTED (-3)

// error string storage (thread-local, read back through afk_last_error())
int afk_set_error(int code, const char* fmt, ...);
// launch counter of a kernel family (AFK_CNT_* in include/afk.h), read back through afk_kernel_counts()
void afk_count(int id);

#define AFK_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return afk_set_error(AFK_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define AFK_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess)                                                         \
            return afk_set_error(AFK_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int64_t afk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- compile-time loops (immediates for inline asm)
template <typename F, int... I>
__device__ __forceinline__ void afk_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void afk_static_for(F&& f) {
    afk_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------- ds_read_b64_tr_b16 as OPAQUE inline asm
// The compiler's waitcnt pass treats the builtin form (__builtin_amdgcn_ds_read_tr16_b64_*) as "may alias an LDS-DMA
// (global_load_lds) destination" and plants an s_waitcnt vmcnt(0) in front of it - which drains every prefetch in flight and
// serialises a double-buffered pipeline (plain ds_read_b128 loads do not get that wait).  Issued as asm the read is invisible
// to that pass; the price is that the CALLER owns the data hazard: call afk_lds_wait0(frag...) (s_waitcnt lgkmcnt(0) tied to
// the registers) before the first use, and order the read against the DMA that filled the image with a counted vmcnt +
// barrier.  LDS returns in order, so the compiler's own counted lgkmcnt waits stay safe (they can only over-wait).
__device__ __forceinline__ uint32_t afk_lds_addr(const void* generic_ptr_into_lds) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)generic_ptr_into_lds;
}
template <int OFF>
__device__ __forceinline__ bf16x4 afk_lds_tr16_b64(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// two tr-reads -> one MFMA k-operand (8 consecutive k)
template <int OFF>
__device__ __forceinline__ bf16x8 afk_lds_tr_frag(uint32_t addr0, uint32_t addr1) {
    const bf16x4 a = afk_lds_tr16_b64<OFF>(addr0);
    const bf16x4 b = afk_lds_tr16_b64<OFF>(addr1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// ds_read_b128 in the same opaque form (the compiler schedules its own ds_reads just in time - two reads, s_waitcnt lgkmcnt(0), two
// MFMAs - which leaves a lone wave at a fraction of the LDS rate; issued as asm the caller decides how many are in flight)
template <int OFF>
__device__ __forceinline__ bf16x8 afk_lds_b128(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// LDS-DMA of 16 bytes per lane in the SCALAR-BASE form: source = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset (VGPR),
// destination = wave-uniform LDS byte address + 16 x lane (M0).  The builtin (__builtin_amdgcn_global_load_lds) always takes a 64-bit per-lane
// pointer: hipcc zero-extends the offset into a VGPR PAIR and adds the base with one v_lshl_add_u64 per piece - 16 VALU instructions and 16 extra
// VGPRs per 64-key attention tile for addresses whose per-lane part never changes.  Issued as asm the piece costs no VALU at all.  The caller owns
// the wait (counted s_waitcnt vmcnt + barrier before the image is read), exactly as with the builtin in these kernels; compiler-inserted vmcnt waits
// for its own loads can only over-wait because of the extra (unknown to it) operations in the in-order queue.
__device__ __forceinline__ void afk_dma16_saddr(const void* wave_uniform_base, uint32_t lane_byte_offset, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_byte_offset), "s"(wave_uniform_base), "s"(lds_dst) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void afk_lgkmcnt() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void afk_lds_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <typename T0, typename... Ts>
__device__ __forceinline__ void afk_lds_tie(T0& a, Ts&... rest) {
    asm volatile("" : "+v"(a));  // every later use of `a` is ordered after this point (and so after the wait before it)
    if constexpr (sizeof...(rest) > 0) afk_lds_tie(rest...);
}
template <typename... Ts>
__device__ __forceinline__ void afk_lds_wait0(Ts&... frags) {
    afk_lds_wait0();
    afk_lds_tie(frags...);
}
// Two-deep ring of fragment groups (4 fragments = 8 tr-reads per group) read with the opaque asm form: while the MFMAs of group g
// run, the reads of group g+1 are in flight.  issue(integral_constant<g>, dst[4]) emits the 8 reads of group g; use(integral_constant<g>,
// frags[4]) consumes them.  Group 0 must already have been issued into `fa` by the caller (typically ahead of a VALU block, so
// that its latency hides there).  lgkmcnt counts at most 15, hence one group of 8 in flight behind the one being waited for; LDS
// returns in order, so lgkmcnt(8) right after issuing group g+1 means group g has landed.
template <int NG, typename Issue, typename Use>
__device__ __forceinline__ void afk_frag_ring(Issue&& issue, Use&& use, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
    afk_static_for<NG>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        bf16x8(&cur)[4] = (g & 1) ? fb : fa;
        bf16x8(&nxt)[4] = (g & 1) ? fa : fb;
        if constexpr (g + 1 < NG) {
            issue(std::integral_constant<int, g + 1>{}, nxt);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        } else {
            afk_lds_wait0();
        }
        afk_lds_tie(cur[0], cur[1], cur[2], cur[3]);
        use(g_, cur);
    });
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
// round-trip through bf16 (models the oracle's bf16 tensor boundaries)
__device__ __forceinline__ float rbf(float x) { return (float)((bf16)x); }
// rbf whose widening step is integer arithmetic.  hipcc builds with -ffp-contract=fast: for rbf(a * c) + y with bf16-valued a, c, instcombine narrows
// fptrunc(fmul(fpext a, fpext c)) to a bf16 multiply and the DAG combiner then folds fadd(fpext(fmul a, c), y) into fma(a, c, y) - the product's rounding
// is gone (found in round 6: 23 % of the fused-RoPE outputs were one ulp off the two-launch form).  With the bits shifted into place there is no fpext to fold.
__device__ __forceinline__ float rbf_strict(float x) {
    const bf16 h = (bf16)x;
    return __uint_as_float((uint32_t)__builtin_bit_cast(unsigned short, h) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum over NW waves via LDS scratch (scratch must hold >= NW floats). All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// exact-erf GELU (oracle: transformers/activations.py GELUActivation -> F.gelu default)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
