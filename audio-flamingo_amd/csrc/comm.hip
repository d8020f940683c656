// afk_comm_*: the data-parallel gradient exchange of the training hot path behind the C ABI (SURVEY.md §8b last row, §8e).
//
// Reference behaviour: torch DistributedDataParallel - bucketed sum-all-reduce of the gradients on a communication stream, overlapped with
// backward (TORCH/nn/parallel/distributed.py:662-666, 828-834; C++ Reducer -> ncclAllReduce).  Here the gradient arena already holds every
// transformer layer's gradients contiguously, so a bucket is (pointer, count): the host calls afk_allreduce_bucket(comm, ptr, n, dtype, stream)
// on a side HIP stream that waits on the event recorded after the layer's last wgrad kernel.
//
// RCCL over xGMI: the 8 GPUs of a node are a full point-to-point mesh (7 links x ~153 GB/s per GPU).  A ring all-reduce is bound by ONE link
// (2 * 7/8 * bytes / 153 GB/s); reduce-scatter + all-gather lets RCCL drive all seven links at once (2 * bytes/8 per link).  Both forms are
// exposed; the exchange is a pure SUM (averaging is folded into the optimizer's grad_scale).
//
// librccl is resolved at first use with dlopen (an already loaded copy - e.g. the one PyTorch ships - is reused), so libafk.so itself carries
// no link-time dependency on it and single-GPU use never touches it.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

#include "common.h"
#include "../../include/afk.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return AFK_OK;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)  // a copy that is already mapped into the process (PyTorch's) wins: one RCCL per process
        if (!h) h = dlopen(n, RTLD_LAZY | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) return afk_set_error(AFK_ERR_UNSUPPORTED, "afk_comm: cannot load librccl.so (%s)", dlerror());
#define AFK_SYM(field, name)                                                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                                                      \
    if (!g_rccl.field) return afk_set_error(AFK_ERR_UNSUPPORTED, "afk_comm: librccl lacks %s", name)
    AFK_SYM(GetUniqueId, "ncclGetUniqueId");
    AFK_SYM(CommInitRank, "ncclCommInitRank");
    AFK_SYM(CommDestroy, "ncclCommDestroy");
    AFK_SYM(AllReduce, "ncclAllReduce");
    AFK_SYM(ReduceScatter, "ncclReduceScatter");
    AFK_SYM(AllGather, "ncclAllGather");
    AFK_SYM(GroupStart, "ncclGroupStart");
    AFK_SYM(GroupEnd, "ncclGroupEnd");
    AFK_SYM(GetErrorString, "ncclGetErrorString");
#undef AFK_SYM
    g_rccl.h = h;
    return AFK_OK;
}

struct AfkComm {
    ncclComm_t comm;
    int rank, world;
};

#define AFK_NCCL(call, what)                                                                              \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != ncclSuccess) return afk_set_error(AFK_ERR_LAUNCH, "%s: %s", what, g_rccl.GetErrorString(r__)); \
    } while (0)

int dtype_of(int dtype, ncclDataType_t* t, size_t* bytes) {
    if (dtype == AFK_COMM_BF16) { *t = ncclBfloat16; *bytes = 2; return AFK_OK; }
    if (dtype == AFK_COMM_F32) { *t = ncclFloat32; *bytes = 4; return AFK_OK; }
    if (dtype == AFK_COMM_I32) { *t = ncclInt32; *bytes = 4; return AFK_OK; }
    return afk_set_error(AFK_ERR_ARG, "afk_comm: dtype %d (0 bf16, 1 f32, 2 i32)", dtype);
}

}  // namespace

extern "C" int afk_comm_unique_id(char* host_out128) {
    AFK_REQUIRE(host_out128, "afk_comm_unique_id: null output");
    if (int e = load_rccl()) return e;
    static_assert(sizeof(ncclUniqueId) == AFK_COMM_UID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    AFK_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(host_out128, &id, sizeof(id));
    return AFK_OK;
}

extern "C" int afk_comm_init(int rank, int world, const char* host_uid128, void** host_comm_out) {
    AFK_REQUIRE(host_uid128 && host_comm_out && world >= 1 && rank >= 0 && rank < world, "afk_comm_init: bad args");
    if (int e = load_rccl()) return e;
    ncclUniqueId id;
    memcpy(&id, host_uid128, sizeof(id));
    AfkComm* c = new AfkComm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);  // uses the calling thread's current HIP device
    if (r != ncclSuccess) {
        delete c;
        return afk_set_error(AFK_ERR_LAUNCH, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
    }
    *host_comm_out = c;
    return AFK_OK;
}

extern "C" int afk_comm_destroy(void* comm) {
    AFK_REQUIRE(comm, "afk_comm_destroy: null communicator");
    AfkComm* c = (AfkComm*)comm;
    AFK_NCCL(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return AFK_OK;
}

// in-place SUM all-reduce of buf[0 .. n) on `stream`
extern "C" int afk_allreduce_bucket(void* comm, void* buf, int64_t n, int dtype, int op_max, void* stream) {
    AFK_REQUIRE(comm && buf && n > 0, "afk_allreduce_bucket: bad args");
    AfkComm* c = (AfkComm*)comm;
    ncclDataType_t t;
    size_t eb;
    if (int e = dtype_of(dtype, &t, &eb)) return e;
    AFK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, t, op_max ? ncclMax : ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
    return AFK_OK;
}

// Partition of a bucket over the ranks: shares start on 128-byte boundaries; what does not divide is a replicated tail.
extern "C" int64_t afk_comm_share(int64_t n, int world, int dtype) {
    if (n <= 0 || world <= 0) return 0;
    const int64_t eb = dtype == AFK_COMM_BF16 ? 2 : 4;
    const int64_t align = 128 / eb;
    return (n / world) / align * align;
}

// in-place reduce-scatter of buf[0 .. n): rank r's share (and the replicated tail, by all-reduce) holds the SUM afterwards
extern "C" int afk_reduce_scatter_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream) {
    AFK_REQUIRE(comm && buf && n > 0, "afk_reduce_scatter_bucket: bad args");
    AfkComm* c = (AfkComm*)comm;
    ncclDataType_t t;
    size_t eb;
    if (int e = dtype_of(dtype, &t, &eb)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int64_t share = afk_comm_share(n, c->world, dtype);
    char* base = (char*)buf;
    if (share > 0) {
        void* mine = base + (size_t)c->rank * share * eb;
        AFK_NCCL(g_rccl.ReduceScatter(buf, mine, (size_t)share, t, ncclSum, c->comm, st), "ncclReduceScatter");
    }
    const int64_t done = share * c->world;
    if (done < n) {
        void* tail = base + (size_t)done * eb;
        AFK_NCCL(g_rccl.AllReduce(tail, tail, (size_t)(n - done), t, ncclSum, c->comm, st), "ncclAllReduce (tail)");
    }
    return AFK_OK;
}

// in-place all-gather of the ranks' own shares of buf[0 .. n); the replicated tail is not touched
extern "C" int afk_allgather_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream) {
    AFK_REQUIRE(comm && buf && n > 0, "afk_allgather_bucket: bad args");
    AfkComm* c = (AfkComm*)comm;
    ncclDataType_t t;
    size_t eb;
    if (int e = dtype_of(dtype, &t, &eb)) return e;
    const int64_t share = afk_comm_share(n, c->world, dtype);
    if (share > 0) {
        void* mine = (char*)buf + (size_t)c->rank * share * eb;
        AFK_NCCL(g_rccl.AllGather(mine, buf, (size_t)share, t, c->comm, (hipStream_t)stream), "ncclAllGather");
    }
    return AFK_OK;
}

// the same result as afk_allreduce_bucket (SUM) by reduce-scatter + all-gather (both in place): every GPU reduces 1/world of the bucket
// and then collects the other shares - on the xGMI mesh all seven links of a GPU work at once.  The tail that does not divide by the
// world size is all-reduced.
extern "C" int afk_reduce_scatter_allgather_bucket(void* comm, void* buf, int64_t n, int dtype, void* stream) {
    if (int e = afk_reduce_scatter_bucket(comm, buf, n, dtype, stream)) return e;
    return afk_allgather_bucket(comm, buf, n, dtype, stream);
}

extern "C" int afk_comm_broadcast(void* comm, void* buf, int64_t n, int dtype, int root, void* stream) {
    AFK_REQUIRE(comm && buf && n > 0, "afk_comm_broadcast: bad args");
    AfkComm* c = (AfkComm*)comm;
    AFK_REQUIRE(root >= 0 && root < c->world, "afk_comm_broadcast: root %d of %d", root, c->world);
    if (int e = load_rccl()) return e;
    ncclDataType_t t;
    size_t eb;
    if (int e = dtype_of(dtype, &t, &eb)) return e;
    // broadcast = all-gather-free form available everywhere: zero the non-root copies is not needed - use ncclBroadcast via dlsym lazily
    static ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    if (!Broadcast) {
        *(void**)(&Broadcast) = dlsym(g_rccl.h, "ncclBroadcast");
        if (!Broadcast) return afk_set_error(AFK_ERR_UNSUPPORTED, "afk_comm: librccl lacks ncclBroadcast");
    }
    AFK_NCCL(Broadcast(buf, buf, (size_t)n, t, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
    return AFK_OK;
}


// ---------------------------------------------------------------------------------------------- CU-contention probe (pre-flight of the 8-GPU run)
// RCCL's collective kernels are persistent workgroups, one per channel, that hold their CU for the whole collective.  A 256x256 GEMM workgroup
// needs an ENTIRE CU (128 KiB of LDS, 8 waves x 256 registers), so every CU a channel sits on is a CU the GEMM rounds lose: 4 736 tiles run 18.5
// rounds on 256 CUs and 19.7 on 240.  Without a second GPU the exchange cannot be run, but its footprint can: afk_cu_hog parks `nblocks`
// workgroups (64 threads + `lds_bytes` of LDS each: above 32 KiB no GEMM workgroup fits beside one) on the stream until *stop_flag becomes non-zero
// (or max_ticks of the 100 MHz clock pass).  bench.py --hog-cus N times the training step beside it; tools/cu_contention.py sweeps N.
namespace {
__global__ __launch_bounds__(64) void cu_hog_kernel(const int* stop_flag, long long max_ticks, long long* report) {
    extern __shared__ char hog_lds[];
    if (threadIdx.x == 0) hog_lds[0] = 0;   // the allocation is what matters
    unsigned long long start, now;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(start)::"memory");
    now = start;
    while (true) {
        // system-scope ACQUIRE: the flag is written by the host (pinned memory) or by a kernel on another XCD - L2s are not coherent with each other,
        // a relaxed load kept hitting this XCD's stale line and the parked workgroups only left at max_ticks
        if (__hip_atomic_load(stop_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        if ((long long)(now - start) > max_ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
    if (report && threadIdx.x == 0) {   // [block][3]: start tick, ticks alive (100 MHz), XCC id << 16 | HW_ID (CU / SE bits)
        const unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
        report[3 * blockIdx.x + 0] = (long long)start;
        report[3 * blockIdx.x + 1] = (long long)(now - start);
        report[3 * blockIdx.x + 2] = ((long long)xcc << 32) | hwid;
    }
}
}  // namespace

extern "C" int afk_cu_hog(int nblocks, int lds_bytes, const int* stop_flag, int64_t max_ticks, int64_t* report, void* stream) {
    AFK_REQUIRE(nblocks > 0 && nblocks <= 256 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && stop_flag && max_ticks > 0, "afk_cu_hog: bad arguments");
    static int once = hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : 1;
    (void)once;
    hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)nblocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, stop_flag, (long long)max_ticks, (long long*)report);
    AFK_LAUNCH_CHECK("afk_cu_hog");
    return AFK_OK;
}
