// libafk.so: error reporting + version (the rest of the C ABI lives next to its kernels).
#include "common.h"
#include "../../include/afk.h"

static thread_local char g_err[512] = "";

int afk_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* afk_last_error(void) { return g_err; }

// launch counters per kernel family (tests assert WHICH kernel served a shape: include/afk.h AFK_CNT_*)
#include <atomic>
static std::atomic<int64_t> g_counts[AFK_CNT_MAX];
void afk_count(int id) {
    if (id >= 0 && id < AFK_CNT_MAX) g_counts[id].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int afk_kernel_counts(int64_t* host_out, int n) {
    AFK_REQUIRE(host_out && n > 0, "afk_kernel_counts: bad args");
    for (int i = 0; i < n; ++i) host_out[i] = i < AFK_CNT_MAX ? g_counts[i].load(std::memory_order_relaxed) : 0;
    return AFK_OK;
}
extern "C" int afk_kernel_counts_reset(void) {
    for (int i = 0; i < AFK_CNT_MAX; ++i) g_counts[i].store(0, std::memory_order_relaxed);
    return AFK_OK;
}
extern "C" int afk_version(void) { return 1; }
// sha256 prefix of the sources this binary was built from (Makefile: SRC_HASH); the Python binding recomputes it from the tree and refuses
// a stale library, so a test run can only ever exercise a build of the sources it sits next to
#ifndef AFK_SRC_HASH
#define AFK_SRC_HASH "unknown"
#endif
extern "C" const char* afk_build_id(void) { return AFK_SRC_HASH; }
// 1 when this library was built with -DAFK_PROBES (timing probes with wrong results + rejected GEMM schedules compiled in); the shipped build returns 0
extern "C" int afk_has_probes(void) {
#ifdef AFK_PROBES
    return 1;
#else
    return 0;
#endif
}

// ---- streams with an explicit priority (include/afk.h): the side streams of the training step below the default, its critical path above
extern "C" int afk_stream_priority_range(int* host_least, int* host_greatest) {
    AFK_REQUIRE(host_least && host_greatest, "afk_stream_priority_range: null output");
    hipError_t e = hipDeviceGetStreamPriorityRange(host_least, host_greatest);
    AFK_REQUIRE(e == hipSuccess, "hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
    return AFK_OK;
}
extern "C" int afk_stream_create(int priority, void** host_stream_out) {
    AFK_REQUIRE(host_stream_out, "afk_stream_create: null output");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    AFK_REQUIRE(e == hipSuccess, "hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
    if (priority > least) priority = least;        // numerically greater = lower priority
    if (priority < greatest) priority = greatest;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
    AFK_REQUIRE(e == hipSuccess, "hipStreamCreateWithPriority(%d): %s", priority, hipGetErrorString(e));
    *host_stream_out = (void*)s;
    return AFK_OK;
}
// A stream whose kernels may only run on `n_cus` compute units: logical CUs [first_cu, first_cu + n_cus) of the queue's CU mask.  The kernel driver deals the
// mask bits out XCD-first (bit i -> XCD i % 8, then shader engine, then CU inside it), so a run of 8 k consecutive bits is k CUs on EVERY XCD: the masked
// stream keeps an equal share of each XCD's L2 / fabric port.  Masked streams are created at the default queue priority (the HIP entry point has no priority form).
extern "C" int afk_stream_create_cu_mask(int first_cu, int n_cus, void** host_stream_out) {
    AFK_REQUIRE(host_stream_out, "afk_stream_create_cu_mask: null output");
    int dev = 0, total = 0;
    hipError_t e = hipGetDevice(&dev);
    AFK_REQUIRE(e == hipSuccess, "hipGetDevice: %s", hipGetErrorString(e));
    e = hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev);
    AFK_REQUIRE(e == hipSuccess && total > 0, "hipDeviceGetAttribute(MultiprocessorCount): %s", hipGetErrorString(e));
    AFK_REQUIRE(first_cu >= 0 && n_cus > 0 && first_cu + n_cus <= total, "afk_stream_create_cu_mask: CUs [%d, %d) outside the device's %d", first_cu, first_cu + n_cus, total);
    uint32_t mask[32] = {};
    AFK_REQUIRE(total <= 32 * 32, "afk_stream_create_cu_mask: %d CUs", total);
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    AFK_REQUIRE(e == hipSuccess, "hipExtStreamCreateWithCUMask(%d CUs from %d): %s", n_cus, first_cu, hipGetErrorString(e));
    *host_stream_out = (void*)s;
    return AFK_OK;
}
extern "C" int afk_stream_destroy(void* stream) {
    AFK_REQUIRE(stream, "afk_stream_destroy: null stream");
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    AFK_REQUIRE(e == hipSuccess, "hipStreamDestroy: %s", hipGetErrorString(e));
    return AFK_OK;
}
