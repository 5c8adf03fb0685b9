// bgs_multi.hip -- several GPUs of one node from ONE host process, for hosts without torch.distributed
// (include/bgs.h, bgs_multi_connect_rollout).  The path shards trivially: device r plays global game ids
// [r * n, (r + 1) * n) (RNG streams are keyed by global game id, so the union equals the unsharded run); the only
// exchange is the reward gather -- every device's 2-bit outcome codes to the first device with RCCL point-to-point
// calls over xGMI (ncclSend / ncclRecv in one group = a gather), one copy of the gathered codes to the host and the
// host-side expansion into the one reward array.  RCCL is loaded at first use (dlopen), so libbgs.so carries no
// link-time dependency on it and shares the copy a framework in the same process may already have mapped.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgs_capi_util.h"
#include "bgs_common.h"
#include "bgs_internal.h"

namespace {

using bgs::fail;

typedef void* ncclComm_t;
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static const Rccl api = [] {
        Rccl r;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return r;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
        return r;
    }();
    return api;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        int r_ = (expr);                                                                                 \
        if (r_ != 0) {                                                                                   \
            rc = fail(BGS_ERR_RUNTIME, "%s failed: %s", #expr, rccl().GetErrorString(r_));              \
            goto done;                                                                                   \
        }                                                                                                \
    } while (0)

#define HIP_GO(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            rc = fail(BGS_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_));                   \
            goto done;                                                                                   \
        }                                                                                                \
    } while (0)

}  // namespace

extern "C" int bgs_multi_connect_rollout(const int* devices, int n_devices, int height, int width, int count,
                                         int64_t n_per_device, uint64_t seed, int8_t* host_reward, uint64_t* steps) {
    NEED(devices != nullptr && host_reward != nullptr, "NULL argument");
    NEED(n_devices >= 1 && n_devices <= 64, "n_devices must be in 1..64");
    NEED(n_per_device >= 4 && (n_per_device & 3) == 0, "n_per_device must be a positive multiple of 4 (4 outcome codes per byte)");
    NEED(rccl().ok, "RCCL (librccl.so) is not available: %s", dlerror() ? dlerror() : "symbols missing");
    const size_t code_bytes = (size_t)n_per_device / 4;
    std::vector<bgs_batch*> batch(n_devices, nullptr);
    std::vector<hipStream_t> stream(n_devices, nullptr);
    std::vector<uint8_t*> codes(n_devices, nullptr);
    std::vector<ncclComm_t> comm(n_devices, nullptr);
    uint8_t* gathered = nullptr;   // on devices[0]: the codes of all devices, in global game order
    uint8_t* host_codes = nullptr; // page-locked
    bool have_comms = false;
    uint64_t total = 0;
    int rc = BGS_OK;

    for (int r = 0; r < n_devices && rc == BGS_OK; ++r) {
        HIP_GO(hipSetDevice(devices[r]));
        HIP_GO(hipStreamCreateWithFlags(&stream[r], hipStreamNonBlocking));
        rc = bgs_connect_create(height, width, count, n_per_device, devices[r], nullptr, 0, &batch[r]);
        if (rc) goto done;
        if ((rc = bgs_set_stream(batch[r], stream[r]))) goto done;
        if ((rc = bgs_set_first_game(batch[r], (uint64_t)r * (uint64_t)n_per_device))) goto done;
        HIP_GO(hipMalloc(reinterpret_cast<void**>(&codes[r]), code_bytes));
    }
    HIP_GO(hipSetDevice(devices[0]));
    HIP_GO(hipMalloc(reinterpret_cast<void**>(&gathered), code_bytes * n_devices));
    HIP_GO(hipHostMalloc(reinterpret_cast<void**>(&host_codes), code_bytes * n_devices, hipHostMallocDefault));
    NCCL_TRY(rccl().CommInitAll(comm.data(), n_devices, devices));
    have_comms = true;

    // every device plays its shard and packs its outcomes; all launches are enqueued before anything is waited for
    for (int r = 0; r < n_devices; ++r) {
        if ((rc = bgs_rollout(batch[r], seed, 0x7FFFFFFF, BGS_ROLLOUT_FROM_INITIAL))) goto done;
        if ((rc = bgs_pack_outcomes(batch[r], codes[r]))) goto done;
    }
    // the gather: one group of point-to-point calls, device r -> device 0, each on its own stream behind its rollout
    NCCL_TRY(rccl().GroupStart());
    for (int r = 0; r < n_devices; ++r) {
        HIP_GO(hipSetDevice(devices[r]));
        NCCL_TRY(rccl().Send(codes[r], code_bytes, kNcclUint8, 0, comm[r], stream[r]));
    }
    HIP_GO(hipSetDevice(devices[0]));
    for (int r = 0; r < n_devices; ++r)
        NCCL_TRY(rccl().Recv(gathered + (size_t)r * code_bytes, code_bytes, kNcclUint8, r, comm[0], stream[0]));
    NCCL_TRY(rccl().GroupEnd());
    HIP_GO(hipMemcpyAsync(host_codes, gathered, code_bytes * n_devices, hipMemcpyDeviceToHost, stream[0]));
    for (int r = 0; r < n_devices; ++r) {
        HIP_GO(hipSetDevice(devices[r]));
        HIP_GO(hipStreamSynchronize(stream[r]));
    }
    if ((rc = bgs_expand_outcomes_host(host_codes, 0, n_per_device * n_devices, host_reward))) goto done;
    for (int r = 0; r < n_devices; ++r) {
        uint64_t s = 0;
        if ((rc = bgs_steps(batch[r], &s))) goto done;
        total += s;
    }
    if (steps) *steps = total;

done:
    for (int r = 0; r < n_devices; ++r) {
        (void)hipSetDevice(devices[r]);
        if (have_comms && comm[r]) (void)rccl().CommDestroy(comm[r]);
        if (batch[r]) (void)bgs_destroy(batch[r]);
        if (codes[r]) (void)hipFree(codes[r]);
        if (stream[r]) (void)hipStreamDestroy(stream[r]);
    }
    (void)hipSetDevice(devices[0]);
    if (gathered) (void)hipFree(gathered);
    if (host_codes) (void)hipHostFree(host_codes);
    return rc;
}
