// collective.hip -- the path's one collective behind the C ABI: sum of the six ProcessingStats counters over the ranks of a
// one-process-per-GPU job (the reference merges them under a mutex between its worker threads, src/local_filter.rs:388-396;
// across processes that merge is an all-reduce).  SURVEY.md 8e: reads are sharded, the index is replicated, nothing else is
// exchanged -- 48 bytes once per run, so the link bandwidth is irrelevant and no kernel of ours is involved.
//
// RCCL is bound at first use with dlopen("librccl.so.1"), not at link time: a process that has PyTorch in it already holds
// a librccl of that SONAME (its own copy), and the loader hands back the one that is loaded -- the same reason
// libdeacon_hip.so binds to the HIP runtime the process loaded first.  A host without RCCL can use everything else.
#include "dcn_internal.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#define DCN_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DCN_OK) return _rc; \
    } while (0)

namespace {

// the few declarations of rccl.h that are used (rccl.h:40-43, 187, 220, 260, 339, 448-464, 611), so that the library
// builds where the header is not installed and never links against a particular copy
constexpr int kUniqueIdBytes = 128;
struct RcclUniqueId {
    char internal[kUniqueIdBytes];
};
typedef struct ncclComm *RcclComm;
constexpr int kRcclSuccess = 0, kRcclSum = 0, kRcclUint64 = 5;

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("DCN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.error = dlerror();
        }
        if (!r.handle) return;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p) r.error = std::string("librccl has no ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

int need_rccl(Rccl **out) {
    Rccl &r = rccl();
    if (!r.handle) return dcn_fail(DCN_ERR_HIP, "RCCL is not available (librccl.so.1): " + r.error);
    *out = &r;
    return DCN_OK;
}

int rccl_fail(Rccl *r, const char *what, int code) {
    return dcn_fail(DCN_ERR_HIP, std::string(what) + ": " + r->GetErrorString(code));
}

} // namespace

struct dcn_comm {
    RcclComm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    uint64_t *d_buf = nullptr; // DCN_N_STATS words on the device
};

extern "C" int dcn_comm_available(void) {
    Rccl *r = nullptr;
    return need_rccl(&r);
}

extern "C" int dcn_comm_unique_id(uint8_t id[DCN_COMM_ID_BYTES]) {
    if (!id) return dcn_fail(DCN_ERR_ARG, "id is NULL");
    static_assert(DCN_COMM_ID_BYTES == kUniqueIdBytes, "ncclUniqueId is 128 bytes");
    Rccl *r = nullptr;
    DCN_TRY(need_rccl(&r));
    RcclUniqueId u;
    int rc = r->GetUniqueId(&u);
    if (rc != kRcclSuccess) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id, u.internal, kUniqueIdBytes);
    return DCN_OK;
}

extern "C" void dcn_comm_destroy(dcn_comm *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->d_buf) hipFree(c->d_buf);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int dcn_comm_create(const uint8_t id[DCN_COMM_ID_BYTES], int world_size, int rank, int device, dcn_comm **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!id) return dcn_fail(DCN_ERR_ARG, "id is NULL");
    if (world_size < 1 || rank < 0 || rank >= world_size) return dcn_fail(DCN_ERR_ARG, "rank must be in 0..world_size");
    int ndev = 0;
    DCN_TRY(dcn_device_count(&ndev));
    if (device < 0 || device >= ndev) return dcn_fail(DCN_ERR_ARG, "no such HIP device");
    Rccl *r = nullptr;
    DCN_TRY(need_rccl(&r));
    dcn_comm *c = new (std::nothrow) dcn_comm();
    if (!c) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    c->world = world_size;
    c->rank = rank;
    c->device = device;
    auto fail = [&](int rc) {
        dcn_comm_destroy(c);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) return fail(dcn_fail(DCN_ERR_HIP, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(dcn_fail(DCN_ERR_HIP, "stream creation failed"));
    if (hipMalloc((void **)&c->d_buf, DCN_N_STATS * sizeof(uint64_t)) != hipSuccess) return fail(dcn_fail(DCN_ERR_NOMEM, "hipMalloc failed"));
    RcclUniqueId u;
    memcpy(u.internal, id, kUniqueIdBytes);
    int rc = r->CommInitRank(&c->comm, world_size, u, rank); // collective: every rank of the job is in here together
    if (rc != kRcclSuccess) {
        c->comm = nullptr;
        return fail(rccl_fail(r, "ncclCommInitRank", rc));
    }
    *out = c;
    return DCN_OK;
}

extern "C" int dcn_stats_allreduce_rccl(dcn_comm *comm, dcn_ctx *const *ctxs, int n_ctx, uint64_t counters[DCN_N_STATS]) {
    if (!comm) return dcn_fail(DCN_ERR_ARG, "comm is NULL");
    uint64_t local[DCN_N_STATS];
    DCN_TRY(dcn_stats_allreduce(ctxs, n_ctx, local)); // this process's contexts first (host sum), then the ranks
    Rccl *r = nullptr;
    DCN_TRY(need_rccl(&r));
    DCN_HIP(hipSetDevice(comm->device));
    DCN_HIP(hipMemcpyAsync(comm->d_buf, local, sizeof local, hipMemcpyHostToDevice, comm->stream));
    int rc = r->AllReduce(comm->d_buf, comm->d_buf, DCN_N_STATS, kRcclUint64, kRcclSum, comm->comm, comm->stream);
    if (rc != kRcclSuccess) return rccl_fail(r, "ncclAllReduce", rc);
    DCN_HIP(hipMemcpyAsync(counters, comm->d_buf, sizeof local, hipMemcpyDeviceToHost, comm->stream));
    DCN_HIP(hipStreamSynchronize(comm->stream));
    return DCN_OK;
}
