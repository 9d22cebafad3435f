// Collectives for ONE process that drives several GPUs (one host thread + one SkmCtx per GPU): RCCL communicators made
// with ncclCommInitAll, every collective issued for all devices inside one ncclGroupStart / ncclGroupEnd from the calling
// thread, each on the stream its device's work runs on.  This is the reduction step of the reference's multi-process
// loops (sidekit/sidekit_mpi.py:455-481 Reduce + bcast of the EM accumulators, :186-214 of the TV accumulators,
// :313-314 Gatherv of the i-vectors) for a caller that is a single plain Python process (the reference's
// num_thread / multiprocessing fan-out: sidekit/mixture.py:570-588, sidekit/sidekit_wrappers.py:141-214).
//
// librccl is opened at run time (the copy the process already holds -- PyTorch ships one -- else the system one): libskm
// has no link-time dependency on it and a box without RCCL only loses these entry points.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

#include "skm_common.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;          // filled once, under the lock of the first skm_comm_init_all (callers serialise init themselves)

int load_rccl()
{
    if (g_rccl.handle) return SKM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                         // a copy that is already in the process first
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        skm_set_error("librccl not found (%s): single-process multi-GPU collectives are unavailable", dlerror());
        return SKM_ERR_UNSUPPORTED;
    }
#define SKM_SYM(field, name)                                                       \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));       \
    if (!g_rccl.field) {                                                           \
        skm_set_error("librccl has no symbol %s", name);                           \
        return SKM_ERR_UNSUPPORTED;                                                \
    }
    SKM_SYM(CommInitAll, "ncclCommInitAll")
    SKM_SYM(CommDestroy, "ncclCommDestroy")
    SKM_SYM(GroupStart, "ncclGroupStart")
    SKM_SYM(GroupEnd, "ncclGroupEnd")
    SKM_SYM(AllReduce, "ncclAllReduce")
    SKM_SYM(ReduceScatter, "ncclReduceScatter")
    SKM_SYM(AllGather, "ncclAllGather")
    SKM_SYM(GetErrorString, "ncclGetErrorString")
#undef SKM_SYM
    g_rccl.handle = h;
    return SKM_OK;
}

struct SkmComm {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

#define SKM_NCCL(call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            skm_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);  \
            return SKM_ERR_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

int check(SkmComm *c, void *const *a, void *const *b, void *const *streams)
{
    SKM_REQUIRE(c && a && streams, "skm_comm: NULL communicator, buffer list or stream list");
    for (size_t i = 0; i < c->comms.size(); ++i)
        SKM_REQUIRE(a[i] && (!b || b[i]), "skm_comm: NULL buffer for device %d", c->devices[i]);
    return SKM_OK;
}

}   // namespace

extern "C" int skm_comm_init_all(int ndev, const int *devices, void **comm_out)
{
    SKM_REQUIRE(ndev >= 1 && devices && comm_out, "skm_comm_init_all: ndev >= 1, a device list and an output pointer");
    int have = 0;
    SKM_HIP(hipGetDeviceCount(&have));
    for (int i = 0; i < ndev; ++i) {
        SKM_REQUIRE(devices[i] >= 0 && devices[i] < have, "skm_comm_init_all: device %d of %d visible", devices[i], have);
        for (int j = 0; j < i; ++j)
            SKM_REQUIRE(devices[j] != devices[i], "skm_comm_init_all: device %d listed twice (RCCL wants one rank per GPU)",
                        devices[i]);
    }
    int rc = load_rccl();
    if (rc) return rc;
    SkmComm *c = new SkmComm;
    c->devices.assign(devices, devices + ndev);
    c->comms.resize(ndev);
    int prev = 0;
    (void)hipGetDevice(&prev);
    ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, devices);
    (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        skm_set_error("ncclCommInitAll over %d devices failed: %s", ndev, g_rccl.GetErrorString(r));
        delete c;
        return SKM_ERR_HIP;
    }
    *comm_out = c;
    return SKM_OK;
}

extern "C" int skm_comm_size(void *comm) { return comm ? (int)static_cast<SkmComm *>(comm)->comms.size() : 0; }

extern "C" int skm_comm_destroy(void *comm)
{
    SkmComm *c = static_cast<SkmComm *>(comm);
    if (!c) return SKM_OK;
    for (ncclComm_t k : c->comms) (void)g_rccl.CommDestroy(k);
    delete c;
    return SKM_OK;
}

// In-place sum (op 0) or maximum (op 1) of `count` float64 values over the devices; bufs_dev[i] and streams[i] belong to
// the i-th device of skm_comm_init_all.  Asynchronous: ordered behind the work already on each stream.
extern "C" int skm_allreduce_f64(void *comm, void *const *bufs_dev, size_t count, int op, void *const *streams)
{
    SkmComm *c = static_cast<SkmComm *>(comm);
    int rc = check(c, bufs_dev, nullptr, streams);
    if (rc) return rc;
    SKM_REQUIRE(op == 0 || op == 1, "skm_allreduce_f64: op 0 (sum) or 1 (max)");
    SKM_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < c->comms.size(); ++i) {
        ncclResult_t r = g_rccl.AllReduce(bufs_dev[i], bufs_dev[i], count, ncclFloat64, op ? ncclMax : ncclSum, c->comms[i],
                                          static_cast<hipStream_t>(streams[i]));
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            skm_set_error("ncclAllReduce on device %d failed: %s", c->devices[i], g_rccl.GetErrorString(r));
            return SKM_ERR_HIP;
        }
    }
    SKM_NCCL(g_rccl.GroupEnd());
    return SKM_OK;
}

// Sum over the devices of [ndev * recv_count] float64 values; device i keeps values [i * recv_count, (i + 1) * recv_count)
// in recv_dev[i] (the component-sharded TV M-step: each byte crosses the links once instead of twice).
extern "C" int skm_reduce_scatter_f64(void *comm, void *const *send_dev, void *const *recv_dev, size_t recv_count,
                                      void *const *streams)
{
    SkmComm *c = static_cast<SkmComm *>(comm);
    int rc = check(c, send_dev, recv_dev, streams);
    if (rc) return rc;
    SKM_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < c->comms.size(); ++i) {
        ncclResult_t r = g_rccl.ReduceScatter(send_dev[i], recv_dev[i], recv_count, ncclFloat64, ncclSum, c->comms[i],
                                              static_cast<hipStream_t>(streams[i]));
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            skm_set_error("ncclReduceScatter on device %d failed: %s", c->devices[i], g_rccl.GetErrorString(r));
            return SKM_ERR_HIP;
        }
    }
    SKM_NCCL(g_rccl.GroupEnd());
    return SKM_OK;
}

// recv_dev[i] = the send_bytes of every device, in device order ([ndev * send_bytes] bytes on each device): per-session
// rows (statistics, i-vectors) back to every rank, whatever their element type.
extern "C" int skm_allgather_bytes(void *comm, void *const *send_dev, void *const *recv_dev, size_t send_bytes,
                                   void *const *streams)
{
    SkmComm *c = static_cast<SkmComm *>(comm);
    int rc = check(c, send_dev, recv_dev, streams);
    if (rc) return rc;
    SKM_NCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < c->comms.size(); ++i) {
        ncclResult_t r = g_rccl.AllGather(send_dev[i], recv_dev[i], send_bytes, ncclChar, c->comms[i],
                                          static_cast<hipStream_t>(streams[i]));
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            skm_set_error("ncclAllGather on device %d failed: %s", c->devices[i], g_rccl.GetErrorString(r));
            return SKM_ERR_HIP;
        }
    }
    SKM_NCCL(g_rccl.GroupEnd());
    return SKM_OK;
}
