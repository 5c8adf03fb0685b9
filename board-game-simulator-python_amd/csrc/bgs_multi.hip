// bgs_multi.hip -- the path's only exchange between GPUs: every rank's outcome codes to rank 0 (RCCL over xGMI).
//   * bgs_gather_*: one process per GPU (torch.distributed.run or any other launcher): a PERSISTENT communicator
//     (ncclCommInitRank), a communication stream and a communication thread per rank, so the thread that launches the
//     rollouts never calls into RCCL: it enqueues the rollout, records an event and goes on; the communication thread
//     makes its stream wait for that event, enqueues the send (and, on rank 0, the receives and the hand-over of the
//     gathered codes to the reward sink);
//   * bgs_multi_connect_rollout: several GPUs of one node from ONE host process, for hosts without torch.distributed.  The path shards trivially: device r plays global game ids
// [r * n, (r + 1) * n) (RNG streams are keyed by global game id, so the union equals the unsharded run); the only
// exchange is the reward gather -- every device's 2-bit outcome codes to the first device with RCCL point-to-point
// calls over xGMI (ncclSend / ncclRecv in one group = a gather), one copy of the gathered codes to the host and the
// host-side expansion into the one reward array.  RCCL is loaded at first use (dlopen), so libbgs.so carries no
// link-time dependency on it and shares the copy a framework in the same process may already have mapped.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bgs_capi_util.h"
#include "bgs_common.h"
#include "bgs_internal.h"

namespace {

using bgs::fail;

typedef void* ncclComm_t;
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct NcclUniqueId {
    char internal[128];  // NCCL_UNIQUE_ID_BYTES
};
static_assert(sizeof(NcclUniqueId) == BGS_UNIQUE_ID_BYTES, "bgs.h promises 128 bytes");

struct Rccl {
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;      // optional: what the communicator itself says (bgs_gather_comm)
    int (*CommUserRank)(ncclComm_t, int*) = nullptr;   // optional
    bool ok = false;
    std::string why;   // when !ok
    std::string name;  // what was loaded
};

const Rccl& rccl() {
    static const Rccl api = [] {
        Rccl r;
        void* h = nullptr;
        // BGS_RCCL_LIB=<path>: another library with the same nine entry points -- tests/c/fake_rccl.hip, a test-only
        // transport over shared memory that lets several processes SHARING ONE GPU run the world > 1 code below (RCCL
        // itself refuses two ranks on one device)
        if (const char* path = getenv("BGS_RCCL_LIB")) {
            h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                const char* e = dlerror();
                r.why = std::string("BGS_RCCL_LIB: ") + (e ? e : path);
                return r;
            }
            r.name = path;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) r.name = name;
        }
        if (!h) {
            const char* e = dlerror();  // (one call: a second one returns NULL)
            r.why = e ? e : "librccl.so not found";
            return r;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send &&
               r.Recv && r.GetErrorString;
        if (!r.ok) r.why = "symbols missing";
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

// Several GPUs of one node from ONE process: batches, streams, code buffers and the communicators (ncclCommInitAll) live
// as long as the handle; a rollout is launches on every device, one group of point-to-point calls, one copy, one host
// expansion.
struct bgs_multi {
    std::vector<int> devices;
    int64_t n = 0;               // games per device
    size_t code_bytes = 0;
    std::vector<bgs_batch*> batch;
    std::vector<hipStream_t> stream;
    std::vector<uint8_t*> codes;      // per device: this device's outcome codes
    std::vector<ncclComm_t> comm;
    uint8_t* gathered = nullptr;      // on devices[0]: the codes of all devices, in global game order
    uint8_t* host_codes = nullptr;    // page-locked
    bool have_comms = false;
};

extern "C" int bgs_multi_destroy(bgs_multi* m) {
    if (!m) return BGS_OK;
    for (size_t r = 0; r < m->devices.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        if (r < m->stream.size() && m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
        if (m->have_comms && r < m->comm.size() && m->comm[r]) (void)rccl().CommDestroy(m->comm[r]);
        if (r < m->batch.size() && m->batch[r]) (void)bgs_destroy(m->batch[r]);
        if (r < m->codes.size() && m->codes[r]) (void)hipFree(m->codes[r]);
        if (r < m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->gathered) (void)hipFree(m->gathered);
    if (m->host_codes) (void)hipHostFree(m->host_codes);
    delete m;
    return BGS_OK;
}

extern "C" int bgs_multi_create(const int* devices, int n_devices, int height, int width, int count, int64_t n_per_device,
                                bgs_multi** out) {
    NEED(out != nullptr && devices != nullptr, "NULL argument");
    *out = nullptr;
    NEED(n_devices >= 1 && n_devices <= 64, "n_devices must be in 1..64");
    NEED(n_per_device >= 4 && (n_per_device & 3) == 0, "n_per_device must be a positive multiple of 4 (4 outcome codes per byte)");
    NEED(rccl().ok, "RCCL (librccl.so) is not available: %s", rccl().why.c_str());
    bgs_multi* m = new (std::nothrow) bgs_multi();
    NEED(m != nullptr, "out of host memory");
    m->devices.assign(devices, devices + n_devices);
    m->n = n_per_device;
    m->code_bytes = (size_t)n_per_device / 4;
    m->batch.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->codes.assign(n_devices, nullptr);
    m->comm.assign(n_devices, nullptr);
    int rc = BGS_OK;
    for (int r = 0; r < n_devices; ++r) {
        HIP_GO(hipSetDevice(devices[r]));
        HIP_GO(hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking));
        if ((rc = bgs_connect_create(height, width, count, n_per_device, devices[r], nullptr, 0, &m->batch[r]))) goto done;
        if ((rc = bgs_set_stream(m->batch[r], m->stream[r]))) goto done;
        if ((rc = bgs_set_first_game(m->batch[r], (uint64_t)r * (uint64_t)n_per_device))) goto done;
        HIP_GO(hipMalloc(reinterpret_cast<void**>(&m->codes[r]), (size_t)(n_per_device + 63) / 64 * 16));
    }
    HIP_GO(hipSetDevice(devices[0]));
    HIP_GO(hipMalloc(reinterpret_cast<void**>(&m->gathered), m->code_bytes * n_devices));
    HIP_GO(hipHostMalloc(reinterpret_cast<void**>(&m->host_codes), m->code_bytes * n_devices, hipHostMallocDefault));
    NCCL_TRY(rccl().CommInitAll(m->comm.data(), n_devices, devices));
    m->have_comms = true;
done:
    if (rc != BGS_OK) {
        (void)bgs_multi_destroy(m);
        return rc;
    }
    *out = m;
    return BGS_OK;
}

extern "C" int bgs_multi_rollout(bgs_multi* m, uint64_t seed, int8_t* host_reward, uint64_t* steps) {
    NEED(m != nullptr && host_reward != nullptr, "NULL argument");
    const int n_devices = (int)m->devices.size();
    uint64_t total = 0;
    int rc = BGS_OK;
    // every device plays its shard (the rollout kernel writes the outcome codes itself); all launches are enqueued before
    // anything is waited for
    for (int r = 0; r < n_devices; ++r) {
        if ((rc = bgs_reset_steps(m->batch[r]))) goto done;
        if ((rc = bgs_rollout_pack(m->batch[r], seed, 0x7FFFFFFF, BGS_ROLLOUT_FROM_INITIAL, m->codes[r]))) goto done;
    }
    // the gather: one group of point-to-point calls, device r -> device 0, each on its own stream behind its rollout
    NCCL_TRY(rccl().GroupStart());
    for (int r = 0; r < n_devices; ++r) {
        HIP_GO(hipSetDevice(m->devices[r]));
        NCCL_TRY(rccl().Send(m->codes[r], m->code_bytes, kNcclUint8, 0, m->comm[r], m->stream[r]));
    }
    HIP_GO(hipSetDevice(m->devices[0]));
    for (int r = 0; r < n_devices; ++r)
        NCCL_TRY(rccl().Recv(m->gathered + (size_t)r * m->code_bytes, m->code_bytes, kNcclUint8, r, m->comm[0], m->stream[0]));
    NCCL_TRY(rccl().GroupEnd());
    HIP_GO(hipMemcpyAsync(m->host_codes, m->gathered, m->code_bytes * n_devices, hipMemcpyDeviceToHost, m->stream[0]));
    for (int r = 0; r < n_devices; ++r) {
        HIP_GO(hipSetDevice(m->devices[r]));
        HIP_GO(hipStreamSynchronize(m->stream[r]));
    }
    if ((rc = bgs_expand_outcomes_host(m->host_codes, 0, m->n * n_devices, host_reward))) goto done;
    for (int r = 0; r < n_devices; ++r) {
        uint64_t s = 0;
        if ((rc = bgs_steps(m->batch[r], &s))) goto done;
        total += s;
    }
    if (steps) *steps = total;
done:
    return rc;
}

extern "C" int bgs_multi_connect_rollout(const int* devices, int n_devices, int height, int width, int count,
                                         int64_t n_per_device, uint64_t seed, int8_t* host_reward, uint64_t* steps) {
    NEED(host_reward != nullptr, "NULL argument");
    bgs_multi* m = nullptr;
    int rc = bgs_multi_create(devices, n_devices, height, width, count, n_per_device, &m);
    if (rc) return rc;
    rc = bgs_multi_rollout(m, seed, host_reward, steps);
    (void)bgs_multi_destroy(m);
    return rc;
}


// ------------------------------------------------------------------------------------------------
// bgs_gather: one process per GPU, persistent communicator, communication thread
//
// Rank 0's own codes are not transported: its rollout kernel writes them where they belong among the gathered codes.
// (Round 3 first sent them to itself like everybody else's: RCCL turns a self send / receive into ~25 small fill / copy
// dispatches per group on the communication stream.)
//
// What a step costs (round 4: the machinery is per GROUP of steps, not per step):
//   launching thread  the rollout, ONE event record behind it (as the plain sink's hand-over), a mutex;
//   comm thread       per group of `batch` steps: one hipStreamWaitEvent per distinct launch stream of the group (the last
//                     rollout on each), one group of point-to-point calls, rank 0 without direct receives: ONE copy kernel
//                     for the whole group, ONE event record -- rank 0's sink is told that the group's jobs have arrived when
//                     the LAST job's slot event fires (sink_publish with an event ticket);
//   a one-rank world  has nothing to gather: no communication thread, no communication stream, the communicator is
//                     created (the library loads, the id is good) and destroyed again; a step is exactly the sink's
//                     bgs_sink_rollout.
// Groups are the same on every rank as long as every rank enqueues the same number of steps between two waits for the
// newest ticket (what bgs_pipeline_* and bench.py do): a group closes when `batch` steps are there, and a partial group
// when somebody waits for one of its steps -- at the end of a region, with all of its steps submitted -- or, by itself,
// BGS_GATHER_FLUSH_US (1000) microseconds after its first step was submitted.  Ranks whose groups are cut at different
// steps still match: point-to-point messages between a pair of ranks are matched in the order they were posted.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kMaxGroup = 32;   // steps per group of point-to-point calls (slots <= 64, default batch = slots / 2)

struct CopyList {
    const uint4* src[kMaxGroup];
    uint4* dst[kMaxGroup];
};

// rank 0 without direct receives: the gathered codes of a whole group of steps (device) -> their sink slots as the GPU
// sees them (device-mapped page-locked memory): 16-byte stores over PCIe, one launch per group
__global__ void __launch_bounds__(256) k_codes_to_slots(CopyList list, size_t units) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < units) list.dst[blockIdx.y][i] = list.src[blockIdx.y][i];
}

inline uint8_t check_byte(int rank, size_t i) { return (uint8_t)(rank * 37 + (int)(i * 11 % 251) + 5); }

}  // namespace

struct bgs_gather {
    int device = 0, rank = 0, world = 1, slots = 0;
    int64_t n = 0;            // games per rank
    size_t code_bytes = 0;    // n / 4: what a rank contributes per step
    // rank 0 receives straight into the sink's device-mapped slot (BGS_GATHER_DIRECT=1) or into device memory, from where a
    // copy kernel takes the gathered codes of a group to their slots (the default with peers: until a run on real xGMI has
    // been seen to deliver into device-mapped host memory, the conservative form is the default -- round-3 advisor)
    bool direct = false;
    int transport_check = 0;  // 0: none (one rank); 1: the create-time message arrived intact in the mode asked for;
                              // 2: it did not arrive intact in direct mode, the gather fell back to the copy kernel
    int batch = 0;            // steps per group of point-to-point calls: slots / 2 (BGS_GATHER_BATCH overrides; the launching
                              // thread runs `slots` steps ahead, so half of them can wait for their group to fill)
    int64_t flush_upto = 0;   // somebody waits for a step below this: send partial groups
    int64_t flush_us = 1000;  // a partial group leaves by itself this long after its first step was submitted (BGS_GATHER_FLUSH_US)
    int comm_ranks = -1, comm_rank = -1;   // ncclCommCount / ncclCommUserRank of the communicator (-1: the transport has no such query)
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;        // everything RCCL does for this rank is enqueued here
    std::vector<uint8_t*> codes;         // [slots] device: the rollout kernel writes a step's codes here (ranks != 0)
    std::vector<uint8_t*> gathered;      // [slots] device, rank 0 without direct receives: the codes of all ranks
    std::vector<hipEvent_t> rolled;      // [slots] on the batch's stream, behind the rollout
    std::vector<hipEvent_t> sent;        // [slots] on the communication stream, behind a GROUP (recorded for its last step)
    std::vector<int8_t*> host;           // [slots] rank 0: destination of the step's rewards
    std::vector<hipStream_t> step_stream;  // [slots] the stream the step's rollout was enqueued on
    std::vector<char> step_ok;           // [slots] the rollout and its event were enqueued
    std::vector<char> step_all_end;      // [slots] every game of the step must have ended (uncapped Connect rollout from the start):
                                         //         rank 0's sink then takes a "still running" code for what it is -- a rank whose
                                         //         step failed and whose message was zeros -- and fails the step
    std::vector<int64_t> cover_seq;      // [slots] sequence number of the group the slot's last step left in ...
    std::vector<int> cover_ev;           // [slots] ... and the slot whose `sent` event was recorded behind that group
    int64_t group_seq = 0;
    struct Waited {
        hipStream_t stream;
        int64_t seq;    // newest group the stream has been told to wait for
        int64_t step;   // ... at this step (an entry older than `slots` steps is not trusted: the handle may be a new stream's)
    };
    std::vector<Waited> waited;
    bgs_reward_sink* sink = nullptr;     // rank 0
    std::mutex mu;
    std::condition_variable cv;
    int64_t submitted = 0;               // steps handed to the communication thread
    int64_t enqueued = 0;                // steps whose send / receives are on the communication stream
    bool stop = false;
    bool failed = false;                 // (under mu) a step or the transport failed: every later call reports `error`
    bool comm_broken = false;            // (communication thread only) RCCL or the communication stream failed: nothing more is posted
    std::string error;
    std::thread worker;

    void fail_with(const char* what, const char* detail) {
        std::lock_guard<std::mutex> lock(mu);
        if (!failed) error = std::string(what) + ": " + detail;
        failed = true;
    }
    bool has_failed() {
        std::lock_guard<std::mutex> lock(mu);
        return failed;
    }
    bool own_buffer() const { return !(rank == 0 && direct); }   // the step's codes live in a buffer of the gather's

    // One group: steps [t, t + k) -- stream waits, point-to-point calls, rank 0's copy, one record, rank 0's publishes.
    // A step whose rollout could not be enqueued on THIS rank (step_ok down) still takes part in the group's point-to-point
    // calls: the peers have posted -- or will post -- the matching receives / sends, and a rank that skipped its half would
    // leave them waiting on their communication streams for ever (round-4 advisor).  Such a step's message carries zeros
    // ("every game still running": rank 0 delivers 0 / 0 for those rows), the step is published as failed on the rank that
    // failed, and that rank's gather refuses every later call.  Only a failure of the transport itself (an RCCL or HIP
    // error on the communication stream: comm_broken) stops the posting.
    void run_group(int64_t t, int k, std::vector<int64_t>& st) {
        const Rccl& api = rccl();
        bool ok = !comm_broken;
        bool all_steps_ok = true;
        hipError_t he = hipSuccess;
        int ne = 0;
        for (int i = 0; i < k; ++i) {
            st[i] = t + i;  // rank 0: the launching thread claimed the sink ticket of this step (same numbers)
            if (!step_ok[(t + i) % slots]) all_steps_ok = false;   // (its rollout could not be enqueued: nothing to wait for)
        }
        // the communication stream waits for the LAST rollout of the group on every launch stream it used
        hipStream_t seen[kMaxGroup];
        int n_seen = 0;
        for (int i = k - 1; i >= 0 && ok; --i) {
            const int slot = (int)((t + i) % slots);
            if (!step_ok[slot]) continue;   // (no event was recorded behind it)
            bool dup = false;
            for (int j = 0; j < n_seen; ++j) dup = dup || seen[j] == step_stream[slot];
            if (dup) continue;
            seen[n_seen++] = step_stream[slot];
            if ((he = hipStreamWaitEvent(stream, rolled[slot], 0)) != hipSuccess) ok = false;
        }
        // a failed step's codes: zeros (ranks other than 0 send them; rank 0's own rows of that step are never delivered)
        for (int i = 0; i < k && ok && rank != 0; ++i) {
            const int slot = (int)((t + i) % slots);
            if (!step_ok[slot] && (he = hipMemsetAsync(codes[slot], 0, code_bytes, stream)) != hipSuccess) ok = false;
        }
        // The gather: one group of point-to-point calls, every OTHER rank -> rank 0.
        if (ok) {
            if ((ne = api.GroupStart()) == 0) {
                for (int i = 0; i < k && ne == 0; ++i) {
                    const int slot = (int)((t + i) % slots);
                    if (rank != 0) {
                        ne = api.Send(codes[slot], code_bytes, kNcclUint8, 0, comm, stream);
                    } else {
                        uint8_t* dst = direct ? bgs::sink_slot_device(sink, st[i]) : gathered[slot];
                        for (int r = 1; r < world && ne == 0; ++r)
                            ne = api.Recv(dst + (size_t)r * code_bytes, code_bytes, kNcclUint8, r, comm, stream);
                    }
                }
                const int ge = api.GroupEnd();
                if (ne == 0) ne = ge;
            }
            if (ne != 0) ok = false;
        }
        if (rank == 0 && ok && !direct) {
            CopyList list;
            for (int i = 0; i < k; ++i) {
                list.src[i] = reinterpret_cast<const uint4*>(gathered[(t + i) % slots]);
                list.dst[i] = reinterpret_cast<uint4*>(bgs::sink_slot_device(sink, st[i]));
            }
            const size_t units = (code_bytes * (size_t)world + 15) / 16;  // (both buffers are whole 16-byte units)
            hipLaunchKernelGGL(k_codes_to_slots, dim3((unsigned)((units + 255) / 256), (unsigned)k), dim3(256), 0, stream, list, units);
            if ((he = hipGetLastError()) != hipSuccess) ok = false;
        }
        const int last = (int)((t + k - 1) % slots);
        // `sent`: the code buffers of the group (rank 0: its gathered codes) may be written again -- nobody waits for it
        // on rank 0 when the codes go straight into the sink's slots
        if (ok && own_buffer() && (he = hipEventRecord(sent[last], stream)) != hipSuccess) ok = false;
        if (rank == 0) {
            // ONE arrival event for the group -- the last job's -- and every job of the group points at it
            if (ok && (he = hipEventRecord(bgs::sink_slot_event(sink, st[k - 1]), stream)) != hipSuccess) ok = false;
            for (int i = 0; i < k; ++i)
                bgs::sink_publish(sink, st[i], n * world, host[(t + i) % slots], ok && step_ok[(t + i) % slots], st[k - 1],
                                  step_all_end[(t + i) % slots] != 0);
        }
        if (!ok) comm_broken = true;
        if ((!ok || !all_steps_ok) && !has_failed()) {
            if (ne != 0) fail_with("RCCL", api.GetErrorString(ne));
            else if (he != hipSuccess) fail_with("HIP", hipGetErrorString(he));
            else fail_with("rollout", "a step of the group could not be enqueued");
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            ++group_seq;
            for (int i = 0; i < k; ++i) {
                cover_seq[(t + i) % slots] = group_seq;
                cover_ev[(t + i) % slots] = last;
            }
            enqueued = t + k;
        }
        cv.notify_all();
    }

    // The communication thread (worlds of two ranks and more).  A group closes when `batch` steps are there; a partial
    // group goes out as soon as somebody waits for one of its steps (flush_upto) -- or by itself flush_us microseconds after
    // its first step arrived (round-4 advisor: a rank that submits fewer than `batch` steps and then blocks on something
    // outside the library -- a host barrier, a message from rank 0 -- without waiting for its newest ticket would never post
    // its sends, and rank 0's wait for that step would never return).  In a loop that keeps submitting, a group of 6 fills
    // in ~0.2 ms and the timer never fires.
    void run() {
        (void)hipSetDevice(device);
        std::vector<int64_t> st((size_t)kMaxGroup, -1);
        for (int64_t t = 0;;) {
            int k;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return stop || submitted > t; });
                if (submitted <= t) return;  // stop, nothing left
                auto closed = [&] { return stop || submitted >= t + batch || flush_upto > t; };
                if (!closed()) cv.wait_for(lock, std::chrono::microseconds(flush_us), closed);
                k = (int)std::min<int64_t>(batch, submitted - t);
            }
            run_group(t, k, st);
            t += k;
        }
    }

    // Create-time check of the transport in the mode the gather will use (worlds of two ranks and more; collective):
    // every other rank sends one step's worth of a known pattern, rank 0 receives it where a step's codes would go and
    // compares what arrives in the sink's page-locked slot; rank 0 then tells everybody the verdict (one 16-byte message
    // each).  A direct receive that does not deliver falls back to the copy kernel, once, and says so on stderr.
    int check_transport() {
        const Rccl& api = rccl();
        int rc = BGS_OK;
        uint8_t* verdict_dev = nullptr;
        std::vector<uint8_t> pattern(code_bytes);
        HIP_GO(hipMalloc(reinterpret_cast<void**>(&verdict_dev), 16));
        for (int round = 0; round < 2; ++round) {
            int verdict = 0;  // 0 fine, 1 again without direct receives, 2 give up
            if (rank != 0) {
                for (size_t i = 0; i < code_bytes; ++i) pattern[i] = check_byte(rank, i);
                HIP_GO(hipMemcpyAsync(codes[0], pattern.data(), code_bytes, hipMemcpyHostToDevice, stream));
                NCCL_TRY(api.GroupStart());
                NCCL_TRY(api.Send(codes[0], code_bytes, kNcclUint8, 0, comm, stream));
                NCCL_TRY(api.GroupEnd());
                NCCL_TRY(api.GroupStart());
                NCCL_TRY(api.Recv(verdict_dev, 16, kNcclUint8, 0, comm, stream));
                NCCL_TRY(api.GroupEnd());
                uint8_t got[16] = {0};
                HIP_GO(hipMemcpyAsync(got, verdict_dev, 16, hipMemcpyDeviceToHost, stream));
                HIP_GO(hipStreamSynchronize(stream));
                verdict = got[0];
            } else {
                uint8_t* slot_host = bgs::sink_slot_host(sink, 0);
                uint8_t* slot_dev = bgs::sink_slot_device(sink, 0);
                const size_t all = code_bytes * (size_t)world;
                memset(slot_host, 0, all);
                if (!direct) HIP_GO(hipMemsetAsync(gathered[0], 0, (all + 15) / 16 * 16, stream));
                uint8_t* dst = direct ? slot_dev : gathered[0];
                NCCL_TRY(api.GroupStart());
                for (int r = 1; r < world; ++r) NCCL_TRY(api.Recv(dst + (size_t)r * code_bytes, code_bytes, kNcclUint8, r, comm, stream));
                NCCL_TRY(api.GroupEnd());
                if (!direct) {
                    CopyList list;
                    list.src[0] = reinterpret_cast<const uint4*>(gathered[0]);
                    list.dst[0] = reinterpret_cast<uint4*>(slot_dev);
                    const size_t units = (all + 15) / 16;
                    hipLaunchKernelGGL(k_codes_to_slots, dim3((unsigned)((units + 255) / 256), 1u), dim3(256), 0, stream, list, units);
                    HIP_GO(hipGetLastError());
                }
                HIP_GO(hipStreamSynchronize(stream));
                size_t wrong = 0;
                for (int r = 1; r < world; ++r)
                    for (size_t i = 0; i < code_bytes; ++i) wrong += slot_host[(size_t)r * code_bytes + i] != check_byte(r, i);
                memset(slot_host, 0, all);
                if (wrong == 0) {
                    transport_check = round == 0 ? 1 : 2;
                } else if (direct && round == 0) {
                    verdict = 1;
                    fprintf(stderr, "libbgs: RCCL gather: %zu of %zu bytes received straight into the device-mapped host slot differ "
                                    "from what was sent; falling back to receives into device memory + a copy kernel\n",
                            wrong, code_bytes * (size_t)(world - 1));
                } else {
                    verdict = 2;
                }
                uint8_t msg[16] = {(uint8_t)verdict};
                HIP_GO(hipMemcpyAsync(verdict_dev, msg, 16, hipMemcpyHostToDevice, stream));
                NCCL_TRY(api.GroupStart());
                for (int r = 1; r < world; ++r) NCCL_TRY(api.Send(verdict_dev, 16, kNcclUint8, r, comm, stream));
                NCCL_TRY(api.GroupEnd());
                HIP_GO(hipStreamSynchronize(stream));
                if (verdict == 2) rc = fail(BGS_ERR_RUNTIME, "RCCL gather: %zu bytes of the create-time message did not arrive intact", wrong);
            }
            if (verdict == 0) break;
            if (verdict == 2) {
                if (rc == BGS_OK) rc = fail(BGS_ERR_RUNTIME, "RCCL gather: rank 0 reports that the create-time message did not arrive intact");
                break;
            }
            // again, without direct receives (rank 0 needs the device buffers it did not allocate)
            direct = false;
            if (rank == 0) {
                while ((int)gathered.size() < slots) {
                    void* p = nullptr;
                    HIP_GO(hipMalloc(&p, (code_bytes * (size_t)world + 15) / 16 * 16));
                    gathered.push_back(static_cast<uint8_t*>(p));
                }
            }
        }
    done:
        if (verdict_dev) (void)hipFree(verdict_dev);
        return rc;
    }
};

extern "C" {

int bgs_gather_unique_id(uint8_t* id) {
    NEED(id != nullptr, "id is NULL");
    NEED(rccl().ok, "RCCL (librccl.so) is not available: %s", rccl().why.c_str());
    NcclUniqueId u;
    const int r = rccl().GetUniqueId(&u);
    if (r != 0) return fail(BGS_ERR_RUNTIME, "ncclGetUniqueId failed: %s", rccl().GetErrorString(r));
    memcpy(id, u.internal, sizeof u.internal);
    return BGS_OK;
}

int bgs_gather_create(int device, int rank, int world, const uint8_t* id, int64_t n_per_rank, int slots, int host_threads,
                      bgs_gather** out) {
    NEED(out != nullptr && id != nullptr, "NULL argument");
    *out = nullptr;
    NEED(world >= 1 && world <= 4096 && rank >= 0 && rank < world, "bad rank %d of %d", rank, world);
    NEED(n_per_rank >= 4 && (n_per_rank & 3) == 0, "n_per_rank must be a positive multiple of 4 (4 outcome codes per byte)");
    NEED(slots >= 1 && slots <= 64 && host_threads >= 1, "need 1 <= slots <= 64 and host_threads >= 1");
    NEED(rccl().ok, "RCCL (librccl.so) is not available: %s", rccl().why.c_str());
    HIP_TRY(hipSetDevice(device));
    bgs_gather* g = new (std::nothrow) bgs_gather();
    NEED(g != nullptr, "out of host memory");
    g->device = device;
    g->rank = rank;
    g->world = world;
    g->slots = slots;
    g->n = n_per_rank;
    g->code_bytes = (size_t)n_per_rank / 4;
    g->direct = world == 1;   // (a one-rank world: rank 0's own kernel writes the slot, nothing is received)
    if (const char* e = getenv("BGS_GATHER_DIRECT")) g->direct = atoi(e) != 0 || world == 1;
    if (const char* e = getenv("BGS_GATHER_BATCH")) {
        const int v = atoi(e);
        if (v >= 1) g->batch = v;
    }
    if (g->batch <= 0) g->batch = slots / 2;
    if (g->batch > slots) g->batch = slots;
    if (g->batch > kMaxGroup) g->batch = kMaxGroup;
    if (g->batch < 1) g->batch = 1;
    if (const char* e = getenv("BGS_GATHER_FLUSH_US")) {
        const long long v = atoll(e);
        if (v >= 1) g->flush_us = v;
    }
    g->host.assign(slots, nullptr);
    g->step_stream.assign(slots, nullptr);
    g->step_ok.assign(slots, 0);
    g->step_all_end.assign(slots, 0);
    g->cover_seq.assign(slots, 0);
    g->cover_ev.assign(slots, 0);
    int rc = BGS_OK;
    hipError_t he = hipSuccess;
    const size_t padded = (size_t)(n_per_rank + 63) / 64 * 16;  // the rollout kernels store whole 16-byte units
    if (world > 1) {
        he = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
        for (int k = 0; k < slots && he == hipSuccess; ++k) {
            void* p = nullptr;
            hipEvent_t e = nullptr;
            if (rank != 0 && (he = hipMalloc(&p, padded)) == hipSuccess) g->codes.push_back(static_cast<uint8_t*>(p));
            if (he == hipSuccess && rank == 0 && !g->direct &&
                (he = hipMalloc(&p, (g->code_bytes * (size_t)world + 15) / 16 * 16)) == hipSuccess)
                g->gathered.push_back(static_cast<uint8_t*>(p));
            if (he == hipSuccess && (he = hipEventCreateWithFlags(&e, hipEventDisableTiming)) == hipSuccess) g->rolled.push_back(e);
            if (he == hipSuccess && (he = hipEventCreateWithFlags(&e, hipEventDisableTiming)) == hipSuccess) g->sent.push_back(e);
        }
    }
    if (he != hipSuccess) rc = fail(BGS_ERR_RUNTIME, "gather allocation failed: %s", hipGetErrorString(he));
    if (rc == BGS_OK && rank == 0) rc = bgs_sink_create(device, n_per_rank * world, slots, host_threads, &g->sink);
    static const bool comm_alone = [] { const char* e = bgs::experiment("gather_comm_alone"); return !(e && e[0] == '0'); }();
    if (rc == BGS_OK && (world > 1 || comm_alone)) {
        // collective: every rank of the world is inside this call at the same time
        NcclUniqueId u;
        memcpy(u.internal, id, sizeof u.internal);
        const int r = rccl().CommInitRank(&g->comm, world, u, rank);
        if (r != 0) rc = fail(BGS_ERR_RUNTIME, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    if (rc == BGS_OK && g->comm) {
        // what the communicator says about itself (the first line of an N > 1 bench run carries it: "did RCCL see N ranks?")
        int v = -1;
        if (rccl().CommCount && rccl().CommCount(g->comm, &v) == 0) g->comm_ranks = v;
        v = -1;
        if (rccl().CommUserRank && rccl().CommUserRank(g->comm, &v) == 0) g->comm_rank = v;
    }
    if (rc == BGS_OK && world == 1 && g->comm) {
        // nothing will ever be sent: the library loaded and the id was good, which is all a one-rank world can show
        (void)rccl().CommDestroy(g->comm);
        g->comm = nullptr;
    }
    if (rc == BGS_OK && world > 1) rc = g->check_transport();
    if (rc != BGS_OK) {
        // (bgs_gather_destroy must not overwrite the message of the failure)
        const std::string why = bgs_last_error();
        (void)bgs_gather_destroy(g);
        return fail(rc, "%s", why.c_str());
    }
    if (world > 1) g->worker = std::thread([g] { g->run(); });
    *out = g;
    return BGS_OK;
}

const char* bgs_gather_transport(void) { return rccl().ok ? rccl().name.c_str() : ""; }

int bgs_gather_info(const bgs_gather* g, int* direct, int* batch, int* transport_check) {
    NEED(g != nullptr, "gather is NULL");
    if (direct) *direct = g->direct ? 1 : 0;
    if (batch) *batch = g->world > 1 ? g->batch : 1;
    if (transport_check) *transport_check = g->transport_check;
    return BGS_OK;
}

int bgs_gather_comm(const bgs_gather* g, int* ranks, int* rank) {
    NEED(g != nullptr, "gather is NULL");
    if (ranks) *ranks = g->comm_ranks;
    if (rank) *rank = g->comm_rank;
    return BGS_OK;
}

int bgs_gather_rollout(bgs_gather* g, bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, int8_t* host_reward,
                       int64_t* ticket) {
    // everything that can be refused is refused BEFORE a ticket exists (round-3 advisor: a claimed sink ticket that is
    // never published stalls the sink, and with it bgs_gather_destroy)
    NEED(g != nullptr && b != nullptr, "NULL argument");
    NEED(b->device == g->device, "batch lives on device %d, the gather on device %d", b->device, g->device);
    NEED(b->n == g->n, "batch of %lld games, the gather was made for %lld per rank", (long long)b->n, (long long)g->n);
    NEED(g->rank != 0 || host_reward != nullptr, "rank 0 needs the host array int8[world * n][2]");
    NEED(max_plies >= 0, "max_plies must be >= 0");
#ifdef BGS_TEST_HOOKS
    // fault injection (TEST build only; experiment gather_inject_failure=<step>): that step "cannot be enqueued" after its
    // ticket was claimed -- the path a device error would take
    static const long long inject = bgs::experiment("gather_inject_failure") ? atoll(bgs::experiment("gather_inject_failure")) : -1;
    static const int inject_rank = bgs::experiment("gather_inject_rank") ? atoi(bgs::experiment("gather_inject_rank")) : -1;  // -1: every rank
#endif
    if (g->world == 1) return bgs_sink_rollout(g->sink, b, seed, max_plies, flags, host_reward, ticket);
    HIP_TRY(hipSetDevice(g->device));
    int64_t t;
    int64_t wait_seq = 0;
    int wait_ev = -1;
    {
        // the code buffer of this step was last used `slots` steps ago: its group must be on the communication stream
        // before the batch's stream can be told to wait for it
        std::unique_lock<std::mutex> lock(g->mu);
        t = g->submitted;
        if (g->enqueued <= t - g->slots) {
            g->flush_upto = std::max(g->flush_upto, t - g->slots + 1);
            g->cv.notify_all();
        }
        g->cv.wait(lock, [&] { return g->enqueued > t - g->slots; });
        if (g->failed) return fail(BGS_ERR_RUNTIME, "the reward gather failed earlier: %s", g->error.c_str());
        const int slot = (int)(t % g->slots);
        if (g->own_buffer() && t >= g->slots) {
            // one wait per launch stream and GROUP, not per step: a stream that has been told to wait for a group has
            // waited for every earlier one (the communication stream runs them in order)
            wait_seq = g->cover_seq[slot];
            auto it = std::find_if(g->waited.begin(), g->waited.end(), [&](const bgs_gather::Waited& w) { return w.stream == b->stream; });
            if (it == g->waited.end()) {
                if (g->waited.size() >= 256) g->waited.clear();   // (streams come and go: forgetting one costs a redundant wait)
                g->waited.push_back({b->stream, wait_seq, t});
                wait_ev = g->cover_ev[slot];
            } else {
                if (it->seq < wait_seq || t - it->step > g->slots) wait_ev = g->cover_ev[slot];
                it->seq = std::max(it->seq, wait_seq);
                it->step = t;
            }
        }
    }
    const int slot = (int)(t % g->slots);
    // (rank 0 receiving straight into the sink's slot needs no such wait: the slot is the sink's, and claiming it below
    // blocks until its previous job has been expanded)
    bool ok = true;
    int rc = BGS_OK;
    hipError_t he = hipSuccess;
    if (wait_ev >= 0 && (he = hipStreamWaitEvent(b->stream, g->sent[wait_ev], 0)) != hipSuccess) ok = false;
    // Rank 0 needs no transport for its own codes: its rollout kernel writes them straight to where the gathered codes of
    // games [0, n) belong -- the device buffer the copy kernel reads, or the sink's device-mapped slot.  The sink ticket
    // is claimed here, by the launching thread (claim blocks while the slot's previous job is still being expanded: the
    // same back-pressure as the plain sink's), and carries the gather's own ticket number.
    uint8_t* codes_out = g->rank != 0 ? g->codes[slot] : nullptr;
    if (g->rank == 0) {
        const int64_t st = bgs::sink_claim(g->sink);
        if (st != t) {
            // cannot happen while the gather owns its sink; if it does, the ticket is still published (by the
            // communication thread, in order) and the gather reports it
            ok = false;
            rc = fail(BGS_ERR_RUNTIME, "the gather's sink handed out ticket %lld for step %lld", (long long)st, (long long)t);
        }
        codes_out = g->direct ? bgs::sink_slot_device(g->sink, t) : g->gathered[slot];
    }
    // From here on the step is submitted whatever happens -- a step that could not be enqueued travels through the
    // communication thread with its flag down, so that rank 0's sink ticket is published in order and nothing stalls.
#ifdef BGS_TEST_HOOKS
    if (ok && t == inject && (inject_rank < 0 || inject_rank == g->rank)) {
        ok = false;
        rc = fail(BGS_ERR_RUNTIME, "injected failure at step %lld (BGS_GATHER_INJECT_FAILURE)", (long long)t);
    }
#endif
    if (ok && (rc = bgs::rollout_with_codes(b, seed, max_plies, flags, codes_out)) != BGS_OK) ok = false;
    if (ok && (he = hipEventRecord(g->rolled[slot], b->stream)) != hipSuccess) ok = false;
    if (!ok && rc == BGS_OK) rc = fail(BGS_ERR_RUNTIME, "the step could not be enqueued: %s", hipGetErrorString(he));
    bool wake;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        g->host[slot] = host_reward;
        g->step_stream[slot] = b->stream;
        g->step_ok[slot] = ok ? 1 : 0;
        // (a Connect game cannot outlast height x width plies; Bounce games can run to any cap)
        g->step_all_end[slot] = (flags & BGS_ROLLOUT_FROM_INITIAL) && b->game == BGS_GAME_CONNECT && !b->generic &&
                                        (int64_t)max_plies >= (int64_t)b->cg.h * b->cg.w ? 1 : 0;
        g->submitted = t + 1;
        if (!ok) g->flush_upto = std::max(g->flush_upto, t + 1);   // (let the failed step's group leave at once)
        // the communication thread is woken for the FIRST step of a group (it starts the group's flush timer and goes
        // back to sleep), when the group is full, and when somebody asks for a flush -- not for the steps in between
        wake = g->submitted == g->enqueued + 1 || g->submitted >= g->enqueued + g->batch || g->flush_upto > g->enqueued;
    }
    if (wake) g->cv.notify_all();
    if (ticket) *ticket = t;
    return rc;
}

int bgs_gather_wait(bgs_gather* g, int64_t ticket) { return bgs::gather_wait(g, ticket, false); }

}  // extern "C"

int bgs::gather_wait(bgs_gather* g, int64_t ticket, bool urgent) {
    NEED(g != nullptr, "gather is NULL");
    if (g->world == 1) return bgs::sink_wait(g->sink, ticket, urgent);
    int64_t enq;
    {
        std::unique_lock<std::mutex> lock(g->mu);
        NEED(ticket >= 0 && ticket < g->submitted, "unknown ticket %lld", (long long)ticket);
        if (g->enqueued <= ticket) {
            g->flush_upto = std::max(g->flush_upto, ticket + 1);
            g->cv.notify_all();
        }
        g->cv.wait(lock, [&] { return g->enqueued > ticket; });
        if (g->failed) return fail(BGS_ERR_RUNTIME, "the reward gather failed: %s", g->error.c_str());
        enq = g->enqueued;
    }
    if (g->rank == 0) return bgs::sink_wait(g->sink, ticket, urgent);  // the sink belongs to the gather: same ticket numbers
    // other ranks: "my codes have left".  A slot's event is reused `slots` steps later, and a step that old has been
    // sent long ago (its successor could not have been enqueued otherwise)
    if (ticket + g->slots >= enq) {
        int ev;
        {
            std::lock_guard<std::mutex> lock(g->mu);
            ev = g->cover_ev[ticket % g->slots];
        }
        HIP_TRY(hipEventSynchronize(g->sent[ev]));
    }
    return BGS_OK;
}

extern "C" {

int bgs_gather_destroy(bgs_gather* g) {
    if (!g) return BGS_OK;
    (void)hipSetDevice(g->device);
    if (g->worker.joinable()) {
        {
            std::unique_lock<std::mutex> lock(g->mu);
            g->flush_upto = g->submitted;
            g->cv.notify_all();
            g->cv.wait(lock, [&] { return g->enqueued == g->submitted; });
            g->stop = true;
        }
        g->cv.notify_all();
        g->worker.join();
    }
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->sink) (void)bgs_sink_destroy(g->sink);
    if (g->comm) (void)rccl().CommDestroy(g->comm);
    for (auto p : g->codes) (void)hipFree(p);
    for (auto p : g->gathered) (void)hipFree(p);
    for (auto e : g->rolled) (void)hipEventDestroy(e);
    for (auto e : g->sent) (void)hipEventDestroy(e);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return BGS_OK;
}

}  // extern "C"
