// comm.hip -- the one exchange of the multi-GPU path: ranks own
// disjoint blocks of frame pairs, estimate them with no data-path collective, and
// all-gather the recovered poses -- ncclAllGather from RCCL over xGMI, one process
// per GPU.  A few doubles of all-reduce cover the bench's max / sum bookkeeping
// and the barrier.
//
// librccl.so (0.5 GB) is opened lazily with dlopen on the first tdk_comm_* call:
// single-GPU users of libtadataka_hip.so never map it.
#include "tdk_runtime.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};

Rccl g_rccl = {};

tdk_status load_rccl() {
    if (g_rccl.lib) return TDK_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) {
        tdk::set_error("cannot open librccl.so: %s", dlerror());
        return TDK_ERR_HIP;
    }
#define RCCL_SYM(field, name)                                        \
    do {                                                             \
        *(void **)(&g_rccl.field) = dlsym(lib, name);                \
        if (!g_rccl.field) {                                         \
            tdk::set_error("librccl.so has no symbol %s", name);     \
            dlclose(lib);                                            \
            return TDK_ERR_HIP;                                      \
        }                                                            \
    } while (0)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    g_rccl.lib = lib;
    return TDK_OK;
}

#define TDK_RCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            tdk::set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return TDK_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

}  // namespace

struct tdk_comm {
    ncclComm_t comm;
    int rank, world;
    hipStream_t stream;      // host-buffer collectives run here
    double *d_buf;           // [send | recv] staging on the device
    double *h_buf;           // pinned twin
    size_t cap;              // doubles in each half
    // device-resident pose gather (tdk_dvo_gather_poses_start / _finish)
    double *d_poses_all, *h_poses_all;
    size_t poses_cap;        // doubles
    int64_t pending_count;   // doubles of the gather in flight
    bool pending;            // a device-resident gather has been started and not collected
    hipEvent_t done;         // recorded behind it on the batch's stream
};

namespace {

// NCCL wants the operations of one communicator issued in one order on every rank and not
// overlapping on the device: the device-resident pose gather runs on a batch's stream, the
// host-buffer collectives on c->stream -- the latter wait for the former's event first.
tdk_status order_after_pending_gather(tdk_comm *c) {
    if (c->pending) TDK_HIP(hipStreamWaitEvent(c->stream, c->done, 0));
    return TDK_OK;
}

tdk_status comm_reserve(tdk_comm *c, size_t doubles) {
    if (c->cap >= doubles) return TDK_OK;
    TDK_HIP(hipStreamSynchronize(c->stream));
    if (c->d_buf) { (void)hipFree(c->d_buf); c->d_buf = nullptr; }
    if (c->h_buf) { (void)hipHostFree(c->h_buf); c->h_buf = nullptr; }
    c->cap = 0;
    size_t cap = doubles + doubles / 2 + 64;
    TDK_HIP(hipMalloc(&c->d_buf, 2 * cap * sizeof(double)));
    TDK_HIP(hipHostMalloc(&c->h_buf, 2 * cap * sizeof(double), hipHostMallocDefault));
    c->cap = cap;
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_comm_unique_id(uint8_t *id128) {
    TDK_API_GUARD;
    TDK_REQUIRE(id128 != nullptr, "id is NULL");
    TDK_TRY(load_rccl());
    ncclUniqueId id;
    TDK_RCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return TDK_OK;
}

tdk_status tdk_comm_available(void) {
    TDK_API_GUARD;                       // (load_rccl keeps static state)
    return load_rccl();
}

tdk_status tdk_comm_destroy(tdk_comm *c) {
    TDK_API_GUARD;
    if (!c) return TDK_OK;
    if (c->pending && c->done) (void)hipEventSynchronize(c->done);   // a gather in flight still writes d_poses_all / h_poses_all
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->d_poses_all) (void)hipFree(c->d_poses_all);
    if (c->h_poses_all) (void)hipHostFree(c->h_poses_all);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TDK_OK;
}

tdk_status tdk_comm_create(const uint8_t *id128, int rank, int world, tdk_comm **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(id128 && out, "null pointer");
    TDK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank out of range");
    TDK_TRY(tdk::ensure_device());
    TDK_TRY(load_rccl());
    tdk_comm *c = new tdk_comm();   // value-initialised
    c->rank = rank; c->world = world;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        tdk::set_error("stream / event creation failed: %s", hipGetErrorString(e));
        tdk_comm_destroy(c);
        return TDK_ERR_HIP;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);   // collective: every rank calls it
    if (r != ncclSuccess) {
        tdk::set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        c->comm = nullptr;
        tdk_comm_destroy(c);
        return TDK_ERR_HIP;
    }
    *out = c;
    return TDK_OK;
}

tdk_status tdk_comm_rank(tdk_comm *c, int *rank, int *world) {
    TDK_API_GUARD;
    TDK_REQUIRE(c != nullptr, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return TDK_OK;
}

tdk_status tdk_comm_all_gather(tdk_comm *c, const double *send, int64_t count, double *recv) {
    TDK_API_GUARD;
    TDK_REQUIRE(c && send && recv && count >= 0, "bad argument");
    if (count == 0) return TDK_OK;
    const size_t n = (size_t)count, total = n * (size_t)c->world;
    TDK_TRY(comm_reserve(c, total));
    double *d_send = c->d_buf, *d_recv = c->d_buf + c->cap;
    memcpy(c->h_buf, send, n * sizeof(double));
    TDK_TRY(order_after_pending_gather(c));
    TDK_HIP(hipMemcpyAsync(d_send, c->h_buf, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    TDK_RCCL(g_rccl.AllGather(d_send, d_recv, n, ncclDouble, c->comm, c->stream));
    TDK_HIP(hipMemcpyAsync(c->h_buf + c->cap, d_recv, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TDK_HIP(hipStreamSynchronize(c->stream));
    memcpy(recv, c->h_buf + c->cap, total * sizeof(double));
    return TDK_OK;
}

tdk_status tdk_comm_all_reduce(tdk_comm *c, double *values, int64_t count, int op) {
    TDK_API_GUARD;
    TDK_REQUIRE(c && values && count >= 0 && (op == 0 || op == 1), "bad argument");
    if (count == 0) return TDK_OK;
    const size_t n = (size_t)count;
    TDK_TRY(comm_reserve(c, n));
    double *d_send = c->d_buf, *d_recv = c->d_buf + c->cap;
    memcpy(c->h_buf, values, n * sizeof(double));
    TDK_TRY(order_after_pending_gather(c));
    TDK_HIP(hipMemcpyAsync(d_send, c->h_buf, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    TDK_RCCL(g_rccl.AllReduce(d_send, d_recv, n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    TDK_HIP(hipMemcpyAsync(c->h_buf + c->cap, d_recv, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TDK_HIP(hipStreamSynchronize(c->stream));
    memcpy(values, c->h_buf + c->cap, n * sizeof(double));
    return TDK_OK;
}

tdk_status tdk_comm_barrier(tdk_comm *c) {
    TDK_API_GUARD;
    double one = 1.0;
    return tdk_comm_all_reduce(c, &one, 1, 0);
}

tdk_status tdk_dvo_gather_poses_start(tdk_dvo *h, tdk_comm *c) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && c, "null pointer");
    TDK_REQUIRE(!c->pending, "finish the previous gather first");
    tdk::DvoLevel0 L;
    TDK_TRY(tdk::dvo_level0(h, &L));
    const size_t n = (size_t)L.n_pairs * 12, total = n * (size_t)c->world;
    if (c->poses_cap < total || !c->d_poses_all) {
        if (c->d_poses_all) { (void)hipFree(c->d_poses_all); c->d_poses_all = nullptr; }
        if (c->h_poses_all) { (void)hipHostFree(c->h_poses_all); c->h_poses_all = nullptr; }
        c->poses_cap = 0;
        TDK_HIP(hipMalloc(&c->d_poses_all, total * sizeof(double)));
        TDK_HIP(hipHostMalloc(&c->h_poses_all, total * sizeof(double), hipHostMallocDefault));
        c->poses_cap = total;
    }
    // on the batch's own stream, right behind the estimation that produced the poses -- and behind
    // whatever host-buffer collective is still running on the communicator's stream
    TDK_HIP(hipEventRecord(c->done, c->stream));
    TDK_HIP(hipStreamWaitEvent(L.stream, c->done, 0));
    if (n > 0) {
        TDK_RCCL(g_rccl.AllGather(L.poses, c->d_poses_all, n, ncclDouble, c->comm, L.stream));
        TDK_HIP(hipMemcpyAsync(c->h_poses_all, c->d_poses_all, total * sizeof(double), hipMemcpyDeviceToHost, L.stream));
    }
    TDK_HIP(hipEventRecord(c->done, L.stream));
    c->pending_count = (int64_t)total;
    c->pending = true;
    return TDK_OK;
}

tdk_status tdk_dvo_gather_poses_finish(tdk_comm *c, double *poses_all) {
    TDK_API_GUARD;
    TDK_REQUIRE(c && poses_all, "null pointer");
    TDK_REQUIRE(c->pending, "no gather in flight");
    TDK_HIP(hipEventSynchronize(c->done));
    memcpy(poses_all, c->h_poses_all, (size_t)c->pending_count * sizeof(double));
    c->pending_count = 0;
    c->pending = false;
    return TDK_OK;
}

}  // extern "C"
