// fake_rccl.hip -- TEST-ONLY stand-in for the nccl* entry points (nine required, two optional queries) libbgs.so loads at run time (csrc/bgs_multi.hip,
// `BGS_RCCL_LIB=<this library>`).  RCCL refuses two ranks on one device, and the test box has one GPU: with this transport
// several PROCESSES SHARING THE ONE GPU (or several logical devices of one process) execute libbgs's real world > 1 code --
// the communication thread's groups of ncclSend / ncclRecv, partial groups, receives into the sink's device-mapped
// page-locked slot, the create-time transport check, bgs_multi_rollout's gather -- against the oracle.
//
// It is NOT part of the product and nothing under board-game-simulator-python_amd/ refers to it.  What it keeps of the real
// thing is the contract libbgs relies on: point-to-point messages between a pair of ranks are delivered in the order
// they were posted, a send reads its buffer in stream order (behind the kernels that fill it), a receive's bytes are in
// the destination in stream order (ahead of whatever the stream runs next), and the destination may be any address the
// GPU can write -- device memory or device-mapped host memory.  What it does not keep: it is not fast, and a receive
// blocks the calling HOST thread (inside ncclGroupEnd) until the matching send has been posted and its bytes have left
// the sender's GPU -- which only ever holds up libbgs's communication thread, as a slow network would.
//
// Transport: a shared-memory file (named in the "unique id") with one ring of mailboxes per ordered pair of ranks.
//   send:  stream-ordered copy device -> mailbox entry (the file is registered with HIP), an event, and a helper thread
//          raises the entry's `ready` count once the event has fired;
//   recv:  the host waits for `ready`, a copy KERNEL mailbox entry -> destination on the caller's stream (a kernel, so
//          that device-mapped host destinations work like any other), an event, the helper raises `consumed`.
// Every wait is bounded (60 s; BGS_FAKE_RCCL_TIMEOUT_MS) and ends in an error code, never in a hang.
//
// What it REFUSES (round 6) -- the mistakes that RCCL answers with an error, a hang or silent corruption end in an error
// code here, so that a test sees them:
//   * a receive whose element count or data type differs from the matching send's (RCCL: undefined -- truncation or a hang);
//   * a point-to-point call outside ncclGroupStart / ncclGroupEnd (libbgs posts every send / receive inside a group:
//     ungrouped blocking calls between two ranks that both send first dead-lock on the real thing), a ncclGroupEnd without
//     a ncclGroupStart, a communicator call with a group left open by the same thread at ncclCommDestroy;
//   * a peer outside the communicator, a NULL buffer with a non-zero count, an unknown data type, a destroyed communicator;
//   * a peer that never posts its half: the wait runs into the time limit and the call fails ("never sent" / "never took").
// A test hook of its own: BGS_FAKE_RCCL_MUTE_AFTER=<rank>:<k> -- rank <rank> stops posting its sends after its first k
// messages (they are accepted and dropped), i.e. a peer that has gone silent without an error.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kEntries = 16;         // messages in flight per ordered pair of ranks
constexpr int kMaxWorld = 8;

int64_t timeout_ms() {
    static const int64_t value = [] {
        const char* e = getenv("BGS_FAKE_RCCL_TIMEOUT_MS");
        const long long v = e ? atoll(e) : 0;
        return (int64_t)(v > 0 ? v : 60000);
    }();
    return value;
}

struct alignas(64) Mailbox {
    std::atomic<uint64_t> ready;     // messages whose bytes are in their entry
    std::atomic<uint64_t> consumed;  // messages whose bytes have been copied out
    uint64_t meta[kEntries];         // per entry: (element count << 8) | data type of the message, written by the sender
                                     // before `ready` is raised for it
};

struct alignas(64) Header {
    std::atomic<int> arrived;
    std::atomic<int> departed;
    int world;
    int entries;
    uint64_t msg_bytes;
    Mailbox box[kMaxWorld * kMaxWorld];  // [src * world + dst]
};

constexpr size_t kHeaderBytes = 16384;   // the payload starts on a page boundary (it is registered with HIP)
static_assert(sizeof(Header) <= kHeaderBytes, "header must fit its pages");

thread_local std::string g_error = "no error";
int fail(const char* what) {
    g_error = what;
    fprintf(stderr, "fake_rccl: %s\n", what);
    return 1;
}

struct Op {
    hipEvent_t event;
    std::atomic<uint64_t>* counter;
    uint64_t value;
};

struct Region {   // one per process (ncclCommInitRank) or shared by the comms of ncclCommInitAll
    Header* header = nullptr;
    uint8_t* payload = nullptr;      // host view
    uint8_t* payload_dev = nullptr;  // device view of the same bytes
    size_t bytes = 0;
    bool mapped_file = false;
    int users = 0;
};

struct Comm {
    Region* region = nullptr;
    int rank = 0, world = 1, device = 0;
    uint64_t sent[kMaxWorld] = {0}, received[kMaxWorld] = {0};
    std::thread helper;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Op> ops;
    bool stop = false;
    bool broken = false;
    int mute_after = -1;             // BGS_FAKE_RCCL_MUTE_AFTER: sends beyond this many are dropped
    uint64_t posted = 0;

    uint8_t* entry(int src, int dst, uint64_t seq, bool device_view) const {
        const Header* h = region->header;
        const size_t index = ((size_t)(src * world + dst) * h->entries + seq % h->entries) * h->msg_bytes;
        return (device_view ? region->payload_dev : region->payload) + index;
    }

    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            Op op;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return stop || !ops.empty(); });
                if (ops.empty()) return;
                op = ops.front();
                ops.pop_front();
            }
            if (hipEventSynchronize(op.event) != hipSuccess) broken = true;
            op.counter->store(op.value, std::memory_order_release);
            (void)hipEventDestroy(op.event);
        }
    }
    int after(hipStream_t stream, std::atomic<uint64_t>* counter, uint64_t value) {
        Op op{nullptr, counter, value};
        if (hipEventCreateWithFlags(&op.event, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
        if (hipEventRecord(op.event, stream) != hipSuccess) return fail("hipEventRecord failed");
        {
            std::lock_guard<std::mutex> lock(mu);
            ops.push_back(op);
        }
        cv.notify_one();
        return 0;
    }
};

// the communicators that exist (a destroyed one's memory is gone: its handle is looked up, never dereferenced)
std::mutex g_live_mu;
std::vector<Comm*> g_live;
Comm* live(void* comm) {
    std::lock_guard<std::mutex> lock(g_live_mu);
    for (Comm* c : g_live)
        if (c == comm) return c;
    return nullptr;
}

bool wait_for(const std::atomic<uint64_t>& counter, uint64_t at_least) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms());
    while (counter.load(std::memory_order_acquire) < at_least) {
        if (std::chrono::steady_clock::now() > until) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return true;
}

__global__ void __launch_bounds__(256) k_deliver(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < bytes) dst[i] = src[i];
}

struct Pending {
    bool send;
    void* buf;
    size_t bytes;
    uint64_t meta;   // (count << 8) | type
    int peer;
    Comm* comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;

int execute(const Pending& p) {
    Comm* c = p.comm;
    Header* h = c->region->header;
    if (p.peer < 0 || p.peer >= c->world) return fail("peer out of range");
    if (p.bytes > h->msg_bytes) return fail("message larger than BGS_FAKE_RCCL_MSG_BYTES");
    if (c->broken) return fail("an earlier operation failed");
    if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice failed");
    if (p.send) {
        if (c->mute_after >= 0 && c->posted++ >= (uint64_t)c->mute_after) return 0;   // (the test hook: a peer gone silent)
        Mailbox& box = h->box[c->rank * c->world + p.peer];
        const uint64_t seq = c->sent[p.peer]++;
        if (seq >= (uint64_t)h->entries && !wait_for(box.consumed, seq + 1 - h->entries)) return fail("send: the peer never took the earlier messages (timeout)");
        box.meta[seq % h->entries] = p.meta;   // (published by the release store that raises `ready`)
        if (p.bytes && hipMemcpyAsync(c->entry(c->rank, p.peer, seq, false), p.buf, p.bytes, hipMemcpyDeviceToHost, p.stream) != hipSuccess)
            return fail("send: hipMemcpyAsync failed");
        return c->after(p.stream, &box.ready, seq + 1);
    }
    Mailbox& box = h->box[p.peer * c->world + c->rank];
    const uint64_t seq = c->received[p.peer]++;
    if (!wait_for(box.ready, seq + 1)) return fail("recv: the peer never sent the message (timeout)");
    if (box.meta[seq % h->entries] != p.meta) {
        static thread_local char text[160];
        const uint64_t m = box.meta[seq % h->entries];
        snprintf(text, sizeof text, "recv: count / data type mismatch with the matching send (sent %llu x type %d, receiving %llu x type %d)",
                 (unsigned long long)(m >> 8), (int)(m & 255), (unsigned long long)(p.meta >> 8), (int)(p.meta & 255));
        c->broken = true;
        return fail(text);
    }
    if (p.bytes) {
        hipLaunchKernelGGL(k_deliver, dim3((unsigned)((p.bytes + 255) / 256)), dim3(256), 0, p.stream,
                           c->entry(p.peer, c->rank, seq, true), static_cast<uint8_t*>(p.buf), p.bytes);
        if (hipGetLastError() != hipSuccess) return fail("recv: the copy kernel could not be launched");
    }
    return c->after(p.stream, &box.consumed, seq + 1);
}

size_t region_bytes(int world, size_t msg_bytes) { return kHeaderBytes + (size_t)world * world * kEntries * msg_bytes; }

size_t message_bytes() {
    const char* e = getenv("BGS_FAKE_RCCL_MSG_BYTES");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (size_t)v : (size_t)256 * 1024;
}

int start(Comm* c) {
    if (hipGetDevice(&c->device) != hipSuccess) return fail("hipGetDevice failed");
    if (const char* e = getenv("BGS_FAKE_RCCL_MUTE_AFTER")) {
        int rank = -1, k = -1;
        if (sscanf(e, "%d:%d", &rank, &k) == 2 && rank == c->rank && k >= 0) c->mute_after = k;
    }
    c->helper = std::thread([c] { c->run(); });
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        g_live.push_back(c);
    }
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : g_error.c_str(); }

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return fail("id is NULL");
    memset(id->internal, 0, sizeof id->internal);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/bgs_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)now);
    return 0;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return fail("bad arguments to ncclCommInitRank");
    id.internal[sizeof id.internal - 1] = 0;
    const size_t msg = message_bytes();
    const size_t bytes = region_bytes(world, msg);
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open failed");
    if (ftruncate(fd, (off_t)bytes) != 0) {   // (every rank sets the same size; the file starts out as zeros)
        close(fd);
        return fail("ftruncate failed (is /dev/shm large enough?)");
    }
    void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return fail("mmap failed");
    Region* r = new Region();
    r->header = static_cast<Header*>(base);
    r->payload = static_cast<uint8_t*>(base) + kHeaderBytes;
    r->bytes = bytes;
    r->mapped_file = true;
    r->users = 1;
    if (hipHostRegister(r->payload, bytes - kHeaderBytes, hipHostRegisterMapped) != hipSuccess) return fail("hipHostRegister failed");
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, r->payload, 0) != hipSuccess) return fail("hipHostGetDevicePointer failed");
    r->payload_dev = static_cast<uint8_t*>(dev);
    if (rank == 0) {
        r->header->world = world;
        r->header->entries = kEntries;
        r->header->msg_bytes = msg;
    }
    // the collective part: everybody has mapped the file before anybody goes on (and then its name can go)
    r->header->arrived.fetch_add(1, std::memory_order_acq_rel);
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms());
    while (r->header->arrived.load(std::memory_order_acquire) < world) {
        if (std::chrono::steady_clock::now() > until) return fail("ncclCommInitRank: not every rank arrived (timeout)");
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (rank == 0) (void)shm_unlink(id.internal);
    Comm* c = new Comm();
    c->region = r;
    c->rank = rank;
    c->world = world;
    if (start(c)) return 1;
    *comm = c;
    return 0;
}

int ncclCommInitAll(void** comms, int n, const int* devices) {
    if (!comms || n < 1 || n > kMaxWorld) return fail("bad arguments to ncclCommInitAll");
    const size_t msg = message_bytes();
    const size_t bytes = region_bytes(n, msg);
    int before = 0;
    (void)hipGetDevice(&before);
    void* base = nullptr;
    if (hipHostMalloc(&base, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return fail("hipHostMalloc failed");
    memset(base, 0, sizeof(Header));
    Region* r = new Region();
    r->header = new (base) Header();
    r->payload = static_cast<uint8_t*>(base) + kHeaderBytes;
    r->bytes = bytes;
    r->users = n;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, r->payload, 0) != hipSuccess) return fail("hipHostGetDevicePointer failed");
    r->payload_dev = static_cast<uint8_t*>(dev);
    r->header->world = n;
    r->header->entries = kEntries;
    r->header->msg_bytes = msg;
    for (int k = 0; k < n; ++k) {
        if (devices && hipSetDevice(devices[k]) != hipSuccess) return fail("hipSetDevice failed");
        Comm* c = new Comm();
        c->region = r;
        c->rank = k;
        c->world = n;
        if (start(c)) return 1;
        comms[k] = c;
    }
    (void)hipSetDevice(before);
    return 0;
}


int ncclCommDestroy(void* comm) {
    if (!comm) return 0;
    Comm* c = live(comm);
    if (!c) return fail("ncclCommDestroy: not a live communicator (destroyed twice?)");
    if (g_depth > 0) return fail("ncclCommDestroy inside an open group (ncclGroupEnd is missing)");
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        for (size_t i = 0; i < g_live.size(); ++i)
            if (g_live[i] == c) { g_live.erase(g_live.begin() + i); break; }
    }
    {
        std::lock_guard<std::mutex> lock(c->mu);
        c->stop = true;
    }
    c->cv.notify_one();
    if (c->helper.joinable()) c->helper.join();
    Region* r = c->region;
    if (--r->users == 0) {
        if (r->mapped_file) {
            (void)hipHostUnregister(r->payload);
            (void)munmap(r->header, r->bytes);
        } else {
            (void)hipHostFree(r->header);
        }
        delete r;
    }
    delete c;
    return 0;
}

int ncclCommCount(void* comm, int* count) {
    if (!live(comm) || !count) return fail("bad arguments to ncclCommCount");
    *count = static_cast<Comm*>(comm)->world;
    return 0;
}

int ncclCommUserRank(void* comm, int* rank) {
    if (!live(comm) || !rank) return fail("bad arguments to ncclCommUserRank");
    *rank = static_cast<Comm*>(comm)->rank;
    return 0;
}

int ncclGroupStart() {
    ++g_depth;
    return 0;
}

int ncclGroupEnd() {
    if (g_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return 0;
    // the sends of a group first: a process that sends to itself (ncclCommInitAll) must have posted them before it
    // waits for them
    std::vector<Pending> todo;
    todo.swap(g_pending);
    int rc = 0;
    for (const Pending& p : todo)
        if (p.send && rc == 0) rc = execute(p);
    for (const Pending& p : todo)
        if (!p.send && rc == 0) rc = execute(p);
    return rc;
}

static size_t type_bytes(int type) {   // ncclDataType_t: 0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double
    static const size_t size[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};
    return type >= 0 && type <= 8 ? size[type] : 0;
}

// what a point-to-point call must look like before it is queued: a live communicator, a known type, a peer of the
// communicator, a buffer -- and an open group (see the header: libbgs never posts outside one)
static int post(bool send, void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream) {
    const char* who = send ? "ncclSend" : "ncclRecv";
    static thread_local char text[128];
    Comm* c = live(comm);
    if (!c) { snprintf(text, sizeof text, "%s: not a live communicator", who); return fail(text); }
    if (!type_bytes(type)) { snprintf(text, sizeof text, "%s: unknown data type %d", who, type); return fail(text); }
    if (peer < 0 || peer >= c->world) { snprintf(text, sizeof text, "%s: peer %d is outside the communicator (%d ranks)", who, peer, c->world); return fail(text); }
    if (!buf && count) { snprintf(text, sizeof text, "%s: NULL buffer with a non-zero count", who); return fail(text); }
    if (g_depth <= 0) { snprintf(text, sizeof text, "%s outside ncclGroupStart / ncclGroupEnd", who); return fail(text); }
    g_pending.push_back(Pending{send, buf, count * type_bytes(type), ((uint64_t)count << 8) | (uint64_t)type, peer, c, stream});
    return 0;
}

int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream) {
    return post(true, const_cast<void*>(buf), count, type, peer, comm, stream);
}

int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream) {
    return post(false, buf, count, type, peer, comm, stream);
}

}  // extern "C"
