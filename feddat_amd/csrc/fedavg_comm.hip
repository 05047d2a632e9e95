// K7, collective half: FedAvg of the shared adapter as ONE RCCL all-reduce over xGMI (replaces get_average_net's
// host-driven K x 48 tensor loop, src/train/main.py:50-65; call site main.py:510).
//
// RCCL is bound at run time (dlopen of the librccl the process already has -- PyTorch-ROCm ships one -- or the system
// one), so libfeddat_hip.so carries no link-time dependency on it and the single-GPU path never touches it.
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "common.hip.h"

namespace {
// the slice of rccl.h this file uses (opaque handles; enum values are fixed by the NCCL ABI)
struct ncclUniqueId_ { char internal[128]; };
typedef int (*GetUniqueIdFn)(ncclUniqueId_*);
typedef int (*CommInitRankFn)(void**, int, ncclUniqueId_, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, void*, hipStream_t);
typedef int (*GetVersionFn)(int*);
typedef int (*CommQueryFn)(void*, int*);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    CommDestroyFn destroy = nullptr;
    CommDestroyFn abort = nullptr;           // optional: ncclCommAbort (releases a communicator without waiting for its peers)
    AllReduceFn all_reduce = nullptr;
    GetVersionFn get_version = nullptr;      // optional (diagnostics only)
    CommQueryFn comm_count = nullptr, comm_user_rank = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

const Rccl& rccl() {
    std::call_once(g_once, [] {
        // First the librccl the process has ALREADY mapped (RTLD_NOLOAD: PyTorch-ROCm ships its own torch/lib/librccl.so and
        // its process group uses it -- binding a second, system RCCL next to it would put two RCCLs with separate state in
        // one process); only if none is mapped, load one by name.
        const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (g_rccl.h) break;
        }
        if (!g_rccl.h && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) g_rccl.h = dlopen(nullptr, RTLD_NOW);   // mapped under another name
        for (const char* name : names) {
            if (g_rccl.h) break;
            g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!g_rccl.h) return;
        g_rccl.get_id = (GetUniqueIdFn)dlsym(g_rccl.h, "ncclGetUniqueId");
        g_rccl.init_rank = (CommInitRankFn)dlsym(g_rccl.h, "ncclCommInitRank");
        g_rccl.destroy = (CommDestroyFn)dlsym(g_rccl.h, "ncclCommDestroy");
        g_rccl.abort = (CommDestroyFn)dlsym(g_rccl.h, "ncclCommAbort");
        g_rccl.all_reduce = (AllReduceFn)dlsym(g_rccl.h, "ncclAllReduce");
        g_rccl.get_version = (GetVersionFn)dlsym(g_rccl.h, "ncclGetVersion");
        g_rccl.comm_count = (CommQueryFn)dlsym(g_rccl.h, "ncclCommCount");
        g_rccl.comm_user_rank = (CommQueryFn)dlsym(g_rccl.h, "ncclCommUserRank");
        g_rccl.ok = g_rccl.get_id && g_rccl.init_rank && g_rccl.destroy && g_rccl.all_reduce;
    });
    return g_rccl;
}
}  // namespace

extern "C" int feddat_comm_unique_id(void* id_128_bytes) {
    FD_CHECK_ARG(id_128_bytes);
    const Rccl& r = rccl();
    if (!r.ok) return FEDDAT_ELAUNCH;
    return r.get_id((ncclUniqueId_*)id_128_bytes) == 0 ? FEDDAT_OK : FEDDAT_ELAUNCH;
}

extern "C" int feddat_comm_create(const void* id_128_bytes, int world, int rank, void** comm_out) {
    return feddat_comm_create_timeout(id_128_bytes, world, rank, 0, comm_out);
}

// ncclCommInitRank is collective: a rank whose peer died (or never loaded RCCL) would wait in the bootstrap forever.  With
// timeout_ms > 0 the call runs on a helper thread (same HIP device as the caller) and the caller gives up after the
// timeout: FEDDAT_ETIMEOUT, *comm_out = NULL, the helper thread is abandoned (detached; its state block is kept alive by
// the shared_ptr it holds), so the surviving ranks can agree on a fallback exchange instead of hanging.
extern "C" int feddat_comm_create_timeout(const void* id_128_bytes, int world, int rank, int timeout_ms, void** comm_out) {
    FD_CHECK_ARG(id_128_bytes && comm_out && world > 0 && rank >= 0 && rank < world && timeout_ms >= 0);
    *comm_out = nullptr;
    const Rccl& r = rccl();
    if (!r.ok) return FEDDAT_ELAUNCH;
    ncclUniqueId_ id;
    __builtin_memcpy(&id, id_128_bytes, sizeof(id));
    if (timeout_ms == 0) return r.init_rank(comm_out, world, id, rank) == 0 ? FEDDAT_OK : FEDDAT_ELAUNCH;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FEDDAT_ELAUNCH;
    struct Job {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        bool abandoned = false;      // the caller gave up (timeout): a communicator that still appears belongs to nobody
        int rc = -1;
        void* comm = nullptr;
    };
    auto job = std::make_shared<Job>();
    const CommInitRankFn init = r.init_rank;
    const CommDestroyFn drop = r.abort ? r.abort : r.destroy;      // ncclCommAbort does not wait for the peers
    std::thread([job, init, drop, id, world, rank, dev] {
        void* c = nullptr;
        int rc = hipSetDevice(dev) == hipSuccess ? init(&c, world, id, rank) : -1;
        bool orphan;
        {
            std::lock_guard<std::mutex> lk(job->mu);
            job->rc = rc;
            job->comm = c;
            job->done = true;
            orphan = job->abandoned;
            job->cv.notify_all();
        }
        // the bootstrap completed after the caller had timed out and moved on to the fallback exchange: nobody will ever
        // use (or destroy) this communicator -- release it here instead of leaking it next to the fallback's collectives
        if (orphan && rc == 0 && c) drop(c);
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
        job->abandoned = true;
        return FEDDAT_ETIMEOUT;
    }
    if (job->rc != 0) return FEDDAT_ELAUNCH;
    *comm_out = job->comm;
    return FEDDAT_OK;
}

extern "C" int feddat_comm_destroy(void* comm) {
    if (!comm) return FEDDAT_OK;
    const Rccl& r = rccl();
    if (!r.ok) return FEDDAT_ELAUNCH;
    return r.destroy(comm) == 0 ? FEDDAT_OK : FEDDAT_ELAUNCH;
}

extern "C" int feddat_comm_info(void* comm, int* rccl_version, int* n_ranks, int* rank) {
    const Rccl& r = rccl();
    if (!r.ok) return FEDDAT_ELAUNCH;
    if (rccl_version) {
        *rccl_version = 0;
        if (r.get_version && r.get_version(rccl_version) != 0) return FEDDAT_ELAUNCH;
    }
    if (n_ranks || rank) FD_CHECK_ARG(comm);
    if (n_ranks) {
        *n_ranks = 0;
        if (r.comm_count && r.comm_count(comm, n_ranks) != 0) return FEDDAT_ELAUNCH;
    }
    if (rank) {
        *rank = -1;
        if (r.comm_user_rank && r.comm_user_rank(comm, rank) != 0) return FEDDAT_ELAUNCH;
    }
    return FEDDAT_OK;
}

extern "C" int feddat_fedavg_allreduce(void* comm, float* flat, float* scratch, long n, float num, float total,
                                       hipStream_t stream) {
    FD_CHECK_ARG(comm && flat && scratch && n > 0 && total > 0.f);
    const Rccl& r = rccl();
    if (!r.ok) return FEDDAT_ELAUNCH;
    // scratch = flat * num / total in the reference's operation order (main.py:62), SUM over the clients, write back
    int rc = feddat_fedavg_accumulate(scratch, flat, n, num, total, 1, stream);
    if (rc != FEDDAT_OK) return rc;
    if (r.all_reduce(scratch, scratch, (size_t)n, kNcclFloat32, kNcclSum, comm, stream) != 0) return FEDDAT_ELAUNCH;
    if (hipMemcpyAsync(flat, scratch, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return FEDDAT_ELAUNCH;
    return FEDDAT_OK;
}
