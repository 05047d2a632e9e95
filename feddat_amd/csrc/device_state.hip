// Per-device launch state of libfeddat_hip.so: CU count, "max dynamic LDS" function attributes, ablation flags.
//
// Nothing here is keyed on "the first device that called": every cache is indexed by the CURRENT HIP device of the calling
// thread and guarded by a mutex, so one process may drive several GPUs (one feddat_ctx per device, or none at all: the
// stateless entry points look the device up themselves).  The ablation flags (tools/ only) are set explicitly through
// feddat_set_debug_flags() (kernel-selection bits only in the production build: common.hip.h FD_ABL); the launch path never
// reads the environment.
#include <atomic>
#include <mutex>
#include <unordered_set>

#include "common.hip.h"

namespace {
constexpr int FD_MAX_DEV = 64;
struct DevState {
    int n_cu = 0;
    std::unordered_set<const void*> lds_attr_done;
};
std::mutex g_mu;
DevState g_dev[FD_MAX_DEV];
std::atomic<int> g_debug{0};
}  // namespace

int fd_device_cus(int* n_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FD_MAX_DEV) return FEDDAT_ELAUNCH;
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& d = g_dev[dev];
    if (!d.n_cu) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return FEDDAT_ELAUNCH;
        d.n_cu = v > 0 ? v : 256;
    }
    *n_cu = d.n_cu;
    return FEDDAT_OK;
}

int fd_set_max_lds(const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FD_MAX_DEV) return FEDDAT_ELAUNCH;
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& d = g_dev[dev];
    if (d.lds_attr_done.count(kernel)) return FEDDAT_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return FEDDAT_ELAUNCH;
    d.lds_attr_done.insert(kernel);
    return FEDDAT_OK;
}

int fd_prepare_all_kernels() {
    int rc = fd_prepare_gemm_kernels();
    return rc != FEDDAT_OK ? rc : fd_prepare_attn_kernels();
}

int fd_debug_flags() { return g_debug.load(std::memory_order_relaxed); }

struct feddat_ctx {
    int device;
    int n_cu;
};

extern "C" int feddat_set_debug_flags(int flags) {
#ifndef FEDDAT_ABLATE
    if ((unsigned)flags & ~FD_DEBUG_SELECT_BITS) return FEDDAT_EINVAL;      // timing-only ablations are not in this build
#endif
    g_debug.store(flags, std::memory_order_relaxed);
    return FEDDAT_OK;
}

extern "C" int feddat_ctx_create(int device, feddat_ctx** out) {
    FD_CHECK_ARG(out && device >= 0 && device < FD_MAX_DEV);
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return FEDDAT_ELAUNCH;
    if (hipSetDevice(device) != hipSuccess) return FEDDAT_EINVAL;
    int n_cu = 0;
    int rc = fd_device_cus(&n_cu);
    if (rc == FEDDAT_OK) rc = fd_prepare_all_kernels();     // every kernel's LDS attribute, for THIS device
    (void)hipSetDevice(prev);
    if (rc != FEDDAT_OK) return rc;
    feddat_ctx* c = new feddat_ctx{device, n_cu};
    *out = c;
    return FEDDAT_OK;
}

extern "C" int feddat_ctx_destroy(feddat_ctx* ctx) {
    delete ctx;
    return FEDDAT_OK;
}

extern "C" int feddat_ctx_device(const feddat_ctx* ctx, int* device, int* compute_units) {
    FD_CHECK_ARG(ctx);
    if (device) *device = ctx->device;
    if (compute_units) *compute_units = ctx->n_cu;
    return FEDDAT_OK;
}
