// bgs_host.hip -- the asynchronous hand-over of results to HOST memory (include/bgs.h, "asynchronous hand-over").
//
// The reference returns `reward` as a host ndarray on every call (State::get_reward, src/simulator/game/connect.cpp:41,
// bounce.cpp:38, through the copying caster tensor.hpp:69-87).  For a batch that hand-over is part of the path and has
// to keep up with the rollout kernel (one batch of 2^20 games every ~50 us), so it is
//   device:  status bytes -> 2-bit outcome codes (k_pack_outcomes, 0.25 B per game)
//   PCIe:    the pack kernel stores the codes straight into a page-locked, device-mapped slot (256 KiB per 2^20
//            games, 16-byte stores; no copy call: a hipMemcpyAsync costs the launching thread more than the kernel)
//   host:    worker threads expand codes -> int8[n][2] reward pairs in the caller's array (AVX2 / table look-up)
// all of it enqueued behind the rollout on the batch's stream and overlapped with the next batches' kernels.
// No game rule lives here: a reward pair is a fixed function of the outcome code (bgs_common.h reward_pair).
#include <immintrin.h>
#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "bgs_capi_util.h"
#include "bgs_common.h"
#include "bgs_internal.h"

struct bgs_event {
    int device;
    hipEvent_t ev;
};

namespace {

using bgs::fail;

// code byte (4 games x 2 bits) -> 8 bytes (4 reward pairs), little endian
struct ExpandTable {
    uint64_t pairs[256];
    ExpandTable() {
        for (int byte = 0; byte < 256; ++byte) {
            uint64_t four = 0;
            for (int j = 0; j < 4; ++j) four |= (uint64_t)bgs::reward_pair((uint32_t)(byte >> (2 * j)) & 3u) << (16 * j);
            pairs[byte] = four;
        }
    }
};
const ExpandTable g_expand;

// 8 code bytes (32 games) -> 64 reward bytes per iteration with AVX2: every code byte is replicated to the four byte
// positions of its games, the 2-bit field of each position is isolated in place, and "field == 1" / "field == 2"
// compares (0 / -1 per byte) give r0 = is2 - is1 and r1 = is1 - is2, interleaved into (r0, r1) pairs.
template <bool STREAM>
__attribute__((target("avx2"))) int64_t expand_avx2(const uint8_t* src, int64_t code_bytes, int8_t* dst) {
    const __m256i spread = _mm256_setr_epi8(0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                            4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7);
    const __m256i field = _mm256_set1_epi32((int)0xC0300C03u);  // bytes 0x03, 0x0C, 0x30, 0xC0
    const __m256i one = _mm256_set1_epi32(0x40100401);           // code 1 in each position
    const __m256i two = _mm256_set1_epi32((int)0x80200802u);    // code 2 in each position
    int64_t i = 0;
    for (; i + 8 <= code_bytes; i += 8) {
        long long eight;
        memcpy(&eight, src + i, 8);
        const __m256i rep = _mm256_shuffle_epi8(_mm256_set1_epi64x(eight), spread);
        const __m256i f = _mm256_and_si256(rep, field);
        const __m256i is1 = _mm256_cmpeq_epi8(f, one), is2 = _mm256_cmpeq_epi8(f, two);
        const __m256i r0 = _mm256_sub_epi8(is2, is1), r1 = _mm256_sub_epi8(is1, is2);
        const __m256i lo = _mm256_unpacklo_epi8(r0, r1), hi = _mm256_unpackhi_epi8(r0, r1);
        const __m256i first = _mm256_permute2x128_si256(lo, hi, 0x20), second = _mm256_permute2x128_si256(lo, hi, 0x31);
        if (STREAM) {
            // non-temporal stores: the rewards are written once and read later by somebody else; skipping the
            // read-for-ownership of every destination line doubles what a memory-bound expansion can deliver
            _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 8 * i), first);
            _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 8 * i + 32), second);
        } else {
            _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + 8 * i), first);
            _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + 8 * i + 32), second);
        }
    }
    if (STREAM) _mm_sfence();
    return i;  // code bytes consumed
}

const bool g_have_avx2 = __builtin_cpu_supports("avx2");
const bool g_stream_stores = bgs::experiment("no_stream_stores") == nullptr;

void expand_range(const uint8_t* packed, int64_t first, int64_t count, int8_t* reward) {
    // whole code bytes: AVX2 where the CPU has it, the 256-entry table otherwise and for the last few bytes; a ragged
    // tail pair by pair
    const int64_t whole = count / 4;
    const uint8_t* src = packed + first / 4;
    int8_t* out = reward + 2 * first;
    int64_t i0 = 0;
    if (g_have_avx2) {
        // streaming stores need a 32-byte aligned destination and pay off on large shares only
        const bool stream = (reinterpret_cast<uintptr_t>(out) & 31u) == 0 && whole >= (1 << 14) && g_stream_stores;
        i0 = stream ? expand_avx2<true>(src, whole, out) : expand_avx2<false>(src, whole, out);
    }
    for (int64_t i = i0; i < whole; ++i) memcpy(out + 8 * i, &g_expand.pairs[src[i]], 8);
    for (int64_t g = first + whole * 4; g < first + count; ++g) {
        const uint16_t pair = bgs::reward_pair((uint32_t)(packed[g / 4] >> (2 * (g & 3))) & 3u);
        memcpy(reward + 2 * g, &pair, 2);
    }
}

// ---- grids: bit sets over the cells (cell = y * W + x) -> int8[n][H][W] -------------------------------------------
// The wire format is plane-major over the batch: word j of bit set p of game i at wire[(p * nwc + j) * n + i].  A cell's
// byte is offset + sum over the sets that contain it of the set's weight: Connect -1 + occupied + player 1's (k_connect_
// cell_planes), Bounce the four value bit-planes with weights 1, 2, 4, 8 (the boards as the batch stores them).  No game
// rule: a change of representation, like the copying tensor caster it stands in for (tensor.hpp:69-87).
struct CellFormat {
    int cells = 0;        // H * W
    int nwc = 0;          // 64-cell words per set
    int sets = 0;         // 2 (Connect) or 4 (Bounce); 0 = the wire IS the int8 grid (generic batches)
    int offset = 0;
    int weight[4] = {0, 0, 0, 0};
};

struct SpreadTable {  // bit b of a byte -> byte b of a uint64 (0 / 1)
    uint64_t v[256];
    SpreadTable() {
        for (int byte = 0; byte < 256; ++byte) {
            uint64_t w = 0;
            for (int b = 0; b < 8; ++b) w |= (uint64_t)((byte >> b) & 1) << (8 * b);
            v[byte] = w;
        }
    }
};
const SpreadTable g_spread;

__attribute__((target("avx512f,avx512bw"))) void expand_cells_avx512(const uint64_t* wire, int64_t n, const CellFormat& f,
                                                                      int64_t first, int64_t count, int8_t* out) {
    const __m512i base = _mm512_set1_epi8((char)f.offset);
    __m512i weight[4];
    for (int p = 0; p < f.sets; ++p) weight[p] = _mm512_set1_epi8((char)f.weight[p]);
    // Boards of up to 256 cells: 64 games at a time are expanded into a block on the stack (64 x cells bytes: whole cache
    // lines whatever the cell count) and the block goes out with non-temporal 64-byte stores -- the grids are written once
    // and read later by somebody else, and a 42-byte masked store per game would read every destination line first.
    alignas(64) int8_t block[64 * 256 + 64];
    const bool lines = f.cells <= 256 && g_stream_stores && (reinterpret_cast<uintptr_t>(out) & 63u) == 0;
#define BGS_ONE_GAME(i_)                                                                                              \
    do {                                                                                                                \
        int8_t* dst_ = out + (i_)*f.cells;                                                                              \
        for (int j = 0; j < f.nwc; ++j) {                                                                               \
            __m512i acc = base;                                                                                         \
            for (int p = 0; p < f.sets; ++p)                                                                            \
                acc = _mm512_mask_add_epi8(acc, (__mmask64)wire[((int64_t)p * f.nwc + j) * n + (i_)], acc, weight[p]);  \
            const int left = f.cells - 64 * j;                                                                          \
            const __mmask64 keep = left >= 64 ? ~0ull : ((1ull << left) - 1ull);                                        \
            _mm512_mask_storeu_epi8(dst_ + 64 * j, keep, acc);                                                          \
        }                                                                                                               \
    } while (0)
    const int64_t end = first + count;
    int64_t i = first;
    if (lines) {
        for (; i < end && (i & 63) != 0; ++i) BGS_ONE_GAME(i);   // up to the first whole block
        for (; i + 64 <= end; i += 64) {
            for (int64_t g = 0; g < 64; ++g) {
                int8_t* dst = block + g * f.cells;
                for (int j = 0; j < f.nwc; ++j) {
                    __m512i acc = base;
                    for (int p = 0; p < f.sets; ++p)
                        acc = _mm512_mask_add_epi8(acc, (__mmask64)wire[((int64_t)p * f.nwc + j) * n + i + g], acc, weight[p]);
                    _mm512_storeu_si512(dst + 64 * j, acc);  // (the tail of the last word is overwritten by the next game)
                }
            }
            int8_t* to = out + i * f.cells;
            for (int b = 0; b < f.cells; ++b)
                _mm512_stream_si512(reinterpret_cast<__m512i*>(to + 64 * b), _mm512_load_si512(block + 64 * b));
        }
        _mm_sfence();
    }
    for (; i < end; ++i) BGS_ONE_GAME(i);
#undef BGS_ONE_GAME
}

void expand_cells_portable(const uint64_t* wire, int64_t n, const CellFormat& f, int64_t first, int64_t count, int8_t* out) {
    const uint64_t base = 0x0101010101010101ull * (uint8_t)f.offset;
    for (int64_t i = first; i < first + count; ++i) {
        int8_t* dst = out + i * f.cells;
        for (int j = 0; j < f.nwc; ++j) {
            uint64_t word[4] = {0, 0, 0, 0};
            for (int p = 0; p < f.sets; ++p) word[p] = wire[((int64_t)p * f.nwc + j) * n + i];
            const int left = f.cells - 64 * j < 64 ? f.cells - 64 * j : 64;
            for (int c = 0; c < left; c += 8) {
                uint64_t sum = 0;  // per byte at most 15 x 4: no carry between bytes
                for (int p = 0; p < f.sets; ++p) sum += g_spread.v[(word[p] >> c) & 255u] * (uint64_t)f.weight[p];
                // + offset in every byte, modulo 256 and without carries into the neighbour (offset -1 is 0xFF)
                constexpr uint64_t low7 = 0x7F7F7F7F7F7F7F7Full;
                const uint64_t eight = ((sum & low7) + (base & low7)) ^ ((sum ^ base) & ~low7);
                memcpy(dst + 64 * j + c, &eight, left - c < 8 ? left - c : 8);
            }
        }
    }
}

const bool g_have_avx512bw = __builtin_cpu_supports("avx512bw") && bgs::experiment("no_avx512") == nullptr;

void expand_cells(const void* wire, int64_t n, const CellFormat& f, int64_t first, int64_t count, int8_t* out) {
    if (f.sets == 0) {  // generic batches keep the reference layout on the device: nothing to expand
        memcpy(out + first * f.cells, static_cast<const int8_t*>(wire) + first * f.cells, (size_t)count * f.cells);
        return;
    }
    if (g_have_avx512bw) expand_cells_avx512(static_cast<const uint64_t*>(wire), n, f, first, count, out);
    else expand_cells_portable(static_cast<const uint64_t*>(wire), n, f, first, count, out);
}

int enter_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return BGS_OK;
}

// ---- progress words: monotonic int64 counters in (possibly shared) host memory that sleepers wait on ---------------
// The futex is the low 32 bits of the word (little endian); a waiter re-reads the whole word after every wake-up.
long futex(volatile void* addr, int op, uint32_t val, const struct timespec* timeout) {
    return syscall(SYS_futex, addr, op, val, timeout, nullptr, 0);
}

void progress_store_max(volatile int64_t* word, int64_t value) {
    auto* a = reinterpret_cast<std::atomic<int64_t>*>(const_cast<int64_t*>(word));
    int64_t seen = a->load(std::memory_order_relaxed);
    while (seen < value && !a->compare_exchange_weak(seen, value, std::memory_order_release, std::memory_order_relaxed)) {
    }
    if (seen < value) futex(word, FUTEX_WAKE, INT_MAX, nullptr);  // (not FUTEX_PRIVATE: waiters may be other processes)
}

// CPUs of the NUMA node the device hangs off, intersected with what this process may use; empty = unknown / one node
bool device_node_cpus(int device, cpu_set_t* out) {
    CPU_ZERO(out);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return false;
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool have = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!have) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int any = 0;
    char* save = nullptr;
    for (char* tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
        int lo = 0, hi = 0;
        const int k = sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, out);
                ++any;
            }
    }
    return any > 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// reward sink: slots of pinned code buffers + worker threads
// ------------------------------------------------------------------------------------------------
namespace {
// experiment sink_trace=1: the sink's threads report when they saw a job's codes, expanded their share and completed it, in
// microseconds of CLOCK_MONOTONIC (what time.perf_counter() reads too) -- for tools/short_run_timeline.py
inline bool sink_trace_on() {
    static const bool on = bgs::experiment("sink_trace") != nullptr;
    return on;
}
inline double mono_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// is there a code 0 ("still running") among games [first, first + count) of the 2-bit codes?  (first is a multiple of 4)
static bool codes_hold_a_zero(const uint8_t* codes, int64_t first, int64_t count) {
    const uint8_t* p = codes + first / 4;
    const int64_t whole = count / 4;
    int64_t i = 0;
    uint64_t seen = 0;   // bit 2k set: the k-th code of some word was 0
    for (; i + 8 <= whole; i += 8) {
        uint64_t x;
        memcpy(&x, p + i, 8);
        seen |= ~(x | (x >> 1)) & 0x5555555555555555ull;
    }
    for (; i < whole; ++i) seen |= (uint64_t)(~(p[i] | (p[i] >> 1)) & 0x55u);
    for (int64_t k = 0; k < count - whole * 4; ++k) seen |= ((p[whole] >> (2 * k)) & 3u) == 0u ? 1u : 0u;
    return seen != 0;
}

struct bgs_reward_sink {
    int device = 0;
    int64_t max_games = 0;
    int slots = 0;
    int threads = 0;
    std::vector<uint8_t*> pinned;     // [slots] page-locked code buffers, (max_games + 3) / 4 bytes each
    std::vector<uint8_t*> mapped;     // [slots] the same buffers as the device sees them: the pack kernel stores its
                                      //         codes straight into host memory (one PCIe write per 16 B, no copy call)
    std::vector<hipEvent_t> landed;   // [slots] recorded behind the copy into pinned[slot]
    bool grids = false;               // a grid sink: the slots carry boards in the wire format `cells` describes
    CellFormat cells;
    size_t slot_bytes = 0;
    struct Job {
        int64_t n_games = 0;
        int8_t* host_reward = nullptr;  // (a grid sink: int8[n][H][W])
        int event_slot = 0;             // the slot whose `landed` event says this job's bytes have arrived: its own, or
                                        // the LAST slot of a group of jobs that were delivered behind one event (the
                                        // in-library gather: one record per group of steps, bgs_multi.hip)
        bool all_end = false;           // every game of this job must have ENDED (an uncapped rollout from the start): a code 0
                                        // ("still running") among them says that a rank's step failed -- its message was zeros
                                        // (bgs_multi.hip) -- and the job is reported as failed (round-5 advisor)
    };
    std::vector<Job> jobs;            // [slots]
    std::mutex mu;
    std::condition_variable cv_submit;   // a job was published / shutdown
    std::condition_variable cv_landed;   // a job's codes have arrived in its slot / shutdown
    std::condition_variable cv_done;     // a job completed
    int64_t claimed = 0;                 // tickets handed out (claim): their slots are reserved
    int64_t submitted = 0;               // jobs [0, submitted) are published, in ticket order
    volatile int64_t* progress = nullptr;  // optional progress word (bgs_sink_set_progress): receives `completed`
    int64_t landed_upto = 0;             // the codes of jobs [0, landed_upto) are in their slots
    int64_t completed = 0;               // jobs [0, completed) are in their host arrays
    std::vector<int> parts_done;         // [slots] workers that finished their share of the slot's job
    std::vector<char> slot_ok;           // [slots] the slot's current job arrived intact (worker 0 -> the expanders)
    bool stop = false;
    // Failures belong to TICKETS, not to the sink: a hand-over that could not be enqueued (or whose event failed) is
    // reported once, to the first bgs_sink_wait for that ticket or a later one, and the deliveries after it are as good
    // as any (round-3 advisor: a sticky flag made one argument error poison every later wait).  Under mu.
    std::vector<int64_t> failed_tickets;
    bool poll = false;                   // worker 0 polls the slot's event instead of sleeping on it (experiment sink_poll=1)
    std::vector<std::thread> workers;
    // Lock-free mirrors of the three counters: a waiter may spin on them for up to spin_us microseconds before it
    // sleeps on the condition variable (BGS_SINK_SPIN_US; default 0 = sleep at once).  Measured on the bench, spinning
    // buys nothing -- with 2 x depth slots the pipeline has enough slack to hide a sleeper's wake-up, and at the end of
    // a run the tail is the last kernels and hipDeviceSynchronize, not the wake-ups -- so the default leaves the cores
    // alone; the knob stays for callers with one batch in flight.  The counters themselves stay under mu.
    std::atomic<int64_t> a_submitted{0}, a_landed{0}, a_completed{0};
    std::atomic<bool> a_stop{false};
    std::atomic<int> urgent{0};          // > 0: somebody is waiting for the LAST deliveries (bgs_sink_wait): worker 0 polls the
                                         // arrival events instead of sleeping in the runtime (a wake-up out of
                                         // hipEventSynchronize costs tens of microseconds, nothing overlaps it at the end)
    int spin_us = 0;
    int wait_spin_us = 200;              // bgs_sink_wait spins this long before it sleeps (BGS_SINK_WAIT_SPIN_US)

    // true once counter > ticket (or stop); false when the spin budget ran out
    bool spin_for(const std::atomic<int64_t>& counter, int64_t ticket) const {
        if (counter.load(std::memory_order_acquire) > ticket) return true;
        if (spin_us <= 0) return false;
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (counter.load(std::memory_order_acquire) > ticket || a_stop.load(std::memory_order_relaxed)) return true;
                _mm_pause();
            }
            if (std::chrono::steady_clock::now() >= until) return false;
        }
    }

    // Worker 0 is the only thread that waits in the HIP runtime: it sleeps until a job is published, waits for the
    // slot's event, then releases the others.  (Every worker waiting on the event itself kept several cores spinning
    // in hipEventSynchronize beside the thread that launches the kernels.)
    void work(int t) {
        (void)hipSetDevice(device);
        for (int64_t ticket = 0;; ++ticket) {
            const int slot = (int)(ticket % slots);
            Job job;
            bool ok = true;
            if (t == 0) {
                spin_for(a_submitted, ticket);
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv_submit.wait(lock, [&] { return stop || submitted > ticket; });
                    if (submitted <= ticket) return;  // stop, nothing left
                    job = jobs[slot];
                }
                if (job.n_games == 0) {
                    ok = false;  // the enqueue failed: no event was recorded for this ticket
                } else if (poll || urgent.load(std::memory_order_relaxed) > 0) {
                    // busy-poll: the wake-up out of hipEventSynchronize costs tens of microseconds, which matters at
                    // the end of a short run (the last delivery is not overlapped with anything)
                    hipError_t e;
                    while ((e = hipEventQuery(landed[job.event_slot])) == hipErrorNotReady) _mm_pause();
                    ok = e == hipSuccess;
                } else {
                    ok = hipEventSynchronize(landed[job.event_slot]) == hipSuccess;
                }
                if (sink_trace_on()) fprintf(stderr, "sink-trace ticket %lld landed %.1f\n", (long long)ticket, mono_us());
                {
                    std::lock_guard<std::mutex> lock(mu);
                    if (!ok && job.n_games != 0) failed_tickets.push_back(ticket);  // (n_games == 0: recorded by publish)
                    slot_ok[slot] = ok;
                    landed_upto = ticket + 1;
                    a_landed.store(ticket + 1, std::memory_order_release);
                }
                cv_landed.notify_all();
            } else {
                if (urgent.load(std::memory_order_relaxed) > 0) {
                    // somebody waits for the LAST deliveries: a wake-up out of the condition variable would cost this
                    // thread what its share of the expansion costs, so it watches the counter instead (bounded: a
                    // waiter that never comes back must not leave the cores spinning)
                    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
                    while (a_landed.load(std::memory_order_acquire) <= ticket && !a_stop.load(std::memory_order_relaxed) &&
                           urgent.load(std::memory_order_relaxed) > 0 && std::chrono::steady_clock::now() < until)
                        for (int i = 0; i < 16; ++i) _mm_pause();
                } else {
                    spin_for(a_landed, ticket);
                }
                std::unique_lock<std::mutex> lock(mu);
                cv_landed.wait(lock, [&] { return stop || landed_upto > ticket; });
                if (landed_upto <= ticket) return;  // stop, nothing left
                job = jobs[slot];  // published before its event could complete, and not reused before this job is done
                ok = slot_ok[slot];
            }
            // With more than one thread, worker 0 only waits and releases: while the others expand job k it is already
            // in the runtime waiting for job k + 1, so the event latency of a job overlaps the expansion of the one
            // before it (and at the end of a run the last few jobs, whose kernels finish together, are released at once)
            const int expanders = threads > 1 ? threads - 1 : 1;
            const int share = threads > 1 ? t - 1 : 0;
            if (threads > 1 && t == 0) continue;
            const double trace_t0 = sink_trace_on() ? mono_us() : 0.0;
            if (ok && grids) {
                // (shares are whole 64-game blocks: a block is what the expansion streams out in full cache lines)
                const int64_t blocks = (job.n_games + 63) / 64;
                const int64_t g0 = blocks * share / expanders * 64;
                const int64_t g1 = share + 1 == expanders ? job.n_games : blocks * (share + 1) / expanders * 64;
                static const bool skip = bgs::experiment("grid_no_expand") != nullptr;  // (measurement knob: the copy alone)
                if (g1 > g0 && !skip) expand_cells(pinned[slot], job.n_games, cells, g0, g1 - g0, job.host_reward);
            } else if (ok) {
                // shares are multiples of 4 games (one code byte), so threads never touch the same output word
                const int64_t bytes = (job.n_games + 3) / 4;
                const int64_t b0 = bytes * share / expanders, b1 = bytes * (share + 1) / expanders;
                const int64_t first = b0 * 4;
                int64_t count = b1 * 4 - first;
                if (first + count > job.n_games) count = job.n_games - first;
                if (count > 0) expand_range(pinned[slot], first, count, job.host_reward);
                if (count > 0 && job.all_end && codes_hold_a_zero(pinned[slot], first, count)) {
                    std::lock_guard<std::mutex> lock(mu);
                    if (std::find(failed_tickets.begin(), failed_tickets.end(), ticket) == failed_tickets.end()) failed_tickets.push_back(ticket);
                }
            }
            if (sink_trace_on()) fprintf(stderr, "sink-trace ticket %lld worker %d expand %.1f .. %.1f\n", (long long)ticket, t, trace_t0, mono_us());
            {
                std::lock_guard<std::mutex> lock(mu);
                if (++parts_done[slot] == expanders) {
                    parts_done[slot] = 0;
                    ++completed;  // jobs complete in ticket order: every worker walks the tickets in order
                    a_completed.store(completed, std::memory_order_release);
                    cv_done.notify_all();
                    if (progress) progress_store_max(progress, completed);  // sleepers in this or another process
                }
            }
        }
    }
};

namespace {

// Reserve the next ticket and its slot, waiting while the ring is full.  The ticket is taken HERE, under the lock, so
// two threads submitting to one sink (say one per stream) never share a slot; what they enqueue for their tickets may
// interleave freely, publish() puts the jobs back in ticket order.
int64_t claim(bgs_reward_sink* s) {
    s->spin_for(s->a_completed, s->a_submitted.load(std::memory_order_relaxed) - s->slots);
    std::unique_lock<std::mutex> lock(s->mu);
    s->cv_done.wait(lock, [&] { return s->claimed - s->completed < s->slots; });
    return s->claimed++;
}

// `ok` false: the enqueue for this ticket failed; the job is published all the same (the ring must not stall) with
// nothing to expand, and the sink remembers the failure
void publish(bgs_reward_sink* s, int64_t ticket, int64_t n_games, int8_t* host_reward, bool ok = true, int64_t event_ticket = -1,
             bool all_end = false) {
    {
        std::unique_lock<std::mutex> lock(s->mu);
        s->cv_done.wait(lock, [&] { return s->submitted == ticket; });  // (tickets of other threads publish first)
        s->jobs[ticket % s->slots].n_games = ok ? n_games : 0;
        s->jobs[ticket % s->slots].all_end = all_end;
        s->jobs[ticket % s->slots].host_reward = host_reward;
        s->jobs[ticket % s->slots].event_slot = (int)((event_ticket >= 0 ? event_ticket : ticket) % s->slots);
        if (!ok) s->failed_tickets.push_back(ticket);
        s->submitted = ticket + 1;
        s->a_submitted.store(ticket + 1, std::memory_order_release);
    }
    s->cv_submit.notify_one();  // only worker 0 waits here
    s->cv_done.notify_all();    // (publishers waiting for their turn)
}

}  // namespace

// ---- the sink as the in-library gather uses it (bgs_multi.hip) ------------------------------------------------------
namespace bgs {
int64_t sink_claim(bgs_reward_sink* s) { return claim(s); }
uint8_t* sink_slot_device(bgs_reward_sink* s, int64_t ticket) { return s->mapped[ticket % s->slots]; }
uint8_t* sink_slot_host(bgs_reward_sink* s, int64_t ticket) { return s->pinned[ticket % s->slots]; }
hipEvent_t sink_slot_event(bgs_reward_sink* s, int64_t ticket) { return s->landed[ticket % s->slots]; }
void sink_publish(bgs_reward_sink* s, int64_t ticket, int64_t n_games, int8_t* host_reward, bool ok, int64_t event_ticket, bool all_end) {
    publish(s, ticket, n_games, host_reward, ok, event_ticket, all_end);
}
// urgent: the caller is at the END of a run (bgs_pipeline_drain): worker 0 polls the arrival events from here on and
// the caller spins a little before it sleeps -- the last deliveries are a few tens of microseconds away and overlap
// with nothing, while a wake-up out of hipEventSynchronize or a condition variable costs as much again.  In the steady
// state (waits that only throttle the launching thread) nobody spins.
int sink_wait(bgs_reward_sink* s, int64_t ticket, bool urgent) {
    NEED(s != nullptr, "sink is NULL");
    NEED(ticket >= 0 && ticket < s->a_submitted.load(std::memory_order_acquire), "unknown ticket %lld", (long long)ticket);
    if (s->a_completed.load(std::memory_order_acquire) <= ticket) {
        if (urgent) {
            s->urgent.fetch_add(1, std::memory_order_relaxed);
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(s->wait_spin_us);
            while (s->a_completed.load(std::memory_order_acquire) <= ticket && std::chrono::steady_clock::now() < until)
                for (int i = 0; i < 32; ++i) _mm_pause();
        } else {
            s->spin_for(s->a_completed, ticket);
        }
        {
            std::unique_lock<std::mutex> lock(s->mu);
            s->cv_done.wait(lock, [&] { return s->completed > ticket; });
        }
        if (urgent) s->urgent.fetch_sub(1, std::memory_order_relaxed);
    }
    // a failure is reported once, to the first waiter for that ticket or a later one
    int64_t bad = -1;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        auto& f = s->failed_tickets;
        for (size_t i = 0; i < f.size();) {
            if (f[i] <= ticket) {
                if (bad < 0 || f[i] < bad) bad = f[i];
                f.erase(f.begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    if (bad >= 0)
        return fail(BGS_ERR_RUNTIME, "reward hand-over %lld failed (its enqueue, its arrival event, or -- a gathered step whose games all "
                    "had to end -- a rank delivered games that are still running: its step failed there); later deliveries are unaffected",
                    (long long)bad);
    return BGS_OK;
}
}  // namespace bgs

extern "C" {

int bgs_host_alloc(size_t bytes, void** host_ptr) {
    NEED(host_ptr != nullptr && bytes > 0, "bad argument");
    *host_ptr = nullptr;
    HIP_TRY(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return BGS_OK;
}

int bgs_host_free(void* host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return BGS_OK;
}

int bgs_stream_create(int device, void** hip_stream) {
    NEED(hip_stream != nullptr, "hip_stream is NULL");
    *hip_stream = nullptr;
    int rc = enter_device(device);
    if (rc) return rc;
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *hip_stream = s;
    return BGS_OK;
}

int bgs_stream_destroy(int device, void* hip_stream) {
    if (!hip_stream) return BGS_OK;
    int rc = enter_device(device);
    if (rc) return rc;
    HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(hip_stream)));
    return BGS_OK;
}

int bgs_event_create(int device, bgs_event** out) {
    NEED(out != nullptr, "out is NULL");
    *out = nullptr;
    int rc = enter_device(device);
    if (rc) return rc;
    bgs_event* e = new (std::nothrow) bgs_event();
    NEED(e != nullptr, "out of host memory");
    e->device = device;
    hipError_t err = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (err != hipSuccess) {
        delete e;
        return fail(BGS_ERR_RUNTIME, "hipEventCreateWithFlags failed: %s", hipGetErrorString(err));
    }
    *out = e;
    return BGS_OK;
}

int bgs_event_destroy(bgs_event* e) {
    if (!e) return BGS_OK;
    (void)hipSetDevice(e->device);
    (void)hipEventDestroy(e->ev);
    delete e;
    return BGS_OK;
}

int bgs_event_synchronize(bgs_event* e) {
    NEED(e != nullptr, "event is NULL");
    HIP_TRY(hipEventSynchronize(e->ev));
    return BGS_OK;
}

int bgs_event_query(bgs_event* e, int* done) {
    NEED(e != nullptr && done != nullptr, "NULL argument");
    const hipError_t err = hipEventQuery(e->ev);
    if (err == hipSuccess) *done = 1;
    else if (err == hipErrorNotReady) *done = 0;
    else return fail(BGS_ERR_RUNTIME, "hipEventQuery failed: %s", hipGetErrorString(err));
    return BGS_OK;
}

int bgs_read_reward_async(bgs_batch* b, int8_t* host_dst, bgs_event* done) {
    NEED(b != nullptr && host_dst != nullptr, "NULL argument");
    int rc = enter_device(b->device);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(host_dst, b->d_reward, (size_t)b->n * 2, hipMemcpyDeviceToHost, b->stream));
    if (done) HIP_TRY(hipEventRecord(done->ev, b->stream));
    return BGS_OK;
}

int bgs_read_outcomes_async(bgs_batch* b, uint8_t* host_dst, bgs_event* done) {
    NEED(b != nullptr && host_dst != nullptr, "NULL argument");
    int rc = enter_device(b->device);
    if (rc) return rc;
    const size_t bytes = (size_t)(b->n + 3) / 4;
    NEED(bytes <= b->staging_bytes, "staging buffer too small");
    // the codes are packed into the head of the staging region: calls that unpack through it are ordered behind this
    // copy on the same stream
    uint8_t* d_packed = b->d_staging;
    bgs::pack_outcomes(b, d_packed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_dst, d_packed, bytes, hipMemcpyDeviceToHost, b->stream));
    if (done) HIP_TRY(hipEventRecord(done->ev, b->stream));
    return BGS_OK;
}

int bgs_expand_outcomes_host(const uint8_t* packed, int64_t first, int64_t count, int8_t* reward) {
    NEED(packed != nullptr && reward != nullptr, "NULL argument");
    NEED(first >= 0 && count >= 0 && (first & 3) == 0, "first must be a non-negative multiple of 4, count >= 0");
    expand_range(packed, first, count, reward);
    return BGS_OK;
}

int bgs_rollout_to_host(bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, void* host_dst, int codes,
                        bgs_event* done) {
    int rc = bgs_rollout(b, seed, max_plies, flags);
    if (rc) return rc;
    return codes ? bgs_read_outcomes_async(b, static_cast<uint8_t*>(host_dst), done)
                 : bgs_read_reward_async(b, static_cast<int8_t*>(host_dst), done);
}

static int make_sink(int device, int64_t max_games, int slots, int threads, size_t slot_bytes, const CellFormat* cells,
                     bgs_reward_sink** out) {
    NEED(out != nullptr, "out is NULL");
    *out = nullptr;
    NEED(max_games >= 1 && slots >= 1 && slots <= 256 && threads >= 1 && threads <= 256,
         "need max_games >= 1, 1 <= slots <= 256, 1 <= threads <= 256");
    int rc = enter_device(device);
    if (rc) return rc;
    bgs_reward_sink* s = new (std::nothrow) bgs_reward_sink();
    NEED(s != nullptr, "out of host memory");
    s->device = device;
    s->max_games = max_games;
    s->slot_bytes = slot_bytes;
    if (cells) {
        s->grids = true;
        s->cells = *cells;
    }
    s->slots = slots;
    s->threads = threads;
    s->jobs.resize(slots);
    if (const char* env = bgs::experiment("sink_poll")) s->poll = atoi(env) != 0;
    if (const char* env = bgs::experiment("sink_wait_spin_us")) {
        const int v = atoi(env);
        if (v >= 0 && v <= 1000000) s->wait_spin_us = v;
    }
    if (const char* env = bgs::experiment("sink_spin_us")) {
        const int v = atoi(env);
        if (v >= 0 && v <= 1000000) s->spin_us = v;
    }
    s->parts_done.assign(slots, 0);
    s->slot_ok.assign(slots, 1);
    const size_t bytes = slot_bytes;
    hipError_t err = hipSuccess;
    // The page-locked slots are written by the GPU and read by the workers, which sit on the GPU's NUMA node: allocate
    // (= first-touch) them from there too, whatever node the calling thread happens to run on (grid hand-over at 2^20
    // boards: 5.4-5.8 against 4.5-5.5 x 10^10 env-steps/s), then give the caller its affinity back.
    const char* aff0 = getenv("BGS_SINK_AFFINITY");
    const bool place = !(aff0 && atoi(aff0) == 0);
    cpu_set_t node_cpus, caller_cpus;
    const bool moved = place && device_node_cpus(device, &node_cpus) && sched_getaffinity(0, sizeof caller_cpus, &caller_cpus) == 0 &&
                       sched_setaffinity(0, sizeof node_cpus, &node_cpus) == 0;
    for (int k = 0; k < slots && err == hipSuccess; ++k) {
        void* host = nullptr;
        void* dev = nullptr;
        hipEvent_t ev = nullptr;
        err = hipHostMalloc(&host, bytes, hipHostMallocMapped);
        if (err == hipSuccess) s->pinned.push_back(static_cast<uint8_t*>(host));
        if (err == hipSuccess) err = hipHostGetDevicePointer(&dev, host, 0);
        if (err == hipSuccess) s->mapped.push_back(static_cast<uint8_t*>(dev));
        if (err == hipSuccess) err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (err == hipSuccess) s->landed.push_back(ev);
    }
    if (moved) (void)sched_setaffinity(0, sizeof caller_cpus, &caller_cpus);
    if (err != hipSuccess) {
        for (auto p : s->pinned) (void)hipHostFree(p);
        for (auto e : s->landed) (void)hipEventDestroy(e);
        delete s;
        return fail(BGS_ERR_RUNTIME, "reward sink allocation failed: %s", hipGetErrorString(err));
    }
    for (int t = 0; t < threads; ++t) s->workers.emplace_back([s, t] { s->work(t); });
    // the workers write the caller's array and read the slots the GPU fills: keep them on the NUMA node the device hangs
    // off (a no-op on one-node hosts and when the process is already confined; BGS_SINK_AFFINITY=0 leaves them alone)
    const char* aff = getenv("BGS_SINK_AFFINITY");
    cpu_set_t cpus;
    if (!(aff && atoi(aff) == 0) && device_node_cpus(device, &cpus))
        for (auto& w : s->workers) (void)pthread_setaffinity_np(w.native_handle(), sizeof cpus, &cpus);
    *out = s;
    return BGS_OK;
}

int bgs_sink_create(int device, int64_t max_games, int slots, int threads, bgs_reward_sink** out) {
    NEED(max_games >= 1, "max_games must be >= 1");
    // whole 16-byte units: kernels store codes dword- / uint4-wise
    return make_sink(device, max_games, slots, threads, (size_t)(max_games + 63) / 64 * 16, nullptr, out);
}

// bytes per board of the grid hand-over's wire format, and how the host expands it
static CellFormat cell_format(const bgs_batch* b) {
    CellFormat f;
    const int h = b->generic ? b->gen_h : (b->game == BGS_GAME_CONNECT ? b->cg.h : b->bg.h);
    const int w = b->generic ? b->gen_w : (b->game == BGS_GAME_CONNECT ? b->cg.w : b->bg.w);
    f.cells = h * w;
    if (b->generic) return f;  // sets = 0: the int8 grid itself crosses PCIe
    f.nwc = (f.cells + 63) / 64;
    if (b->game == BGS_GAME_CONNECT) {
        f.sets = 2;
        f.offset = -1;
        f.weight[0] = f.weight[1] = 1;
    } else {
        f.sets = 4;
        for (int p = 0; p < 4; ++p) f.weight[p] = 1 << p;
    }
    return f;
}

int bgs_grid_sink_create(const bgs_batch* like, int slots, int threads, bgs_reward_sink** out) {
    NEED(like != nullptr, "batch is NULL");
    const CellFormat f = cell_format(like);
    const size_t per_board = f.sets ? (size_t)f.sets * f.nwc * 8 : (size_t)f.cells;
    return make_sink(like->device, like->n, slots, threads, per_board * (size_t)like->n, &f, out);
}

// the boards of `b` in the wire format -> the ticket's slot (one asynchronous copy behind whatever the stream holds)
static hipError_t enqueue_grids(bgs_reward_sink* s, bgs_batch* b, int slot) {
    const void* src = b->d_planes;  // Bounce value planes and generic int8 grids cross as they are
    if (!b->generic && b->game == BGS_GAME_CONNECT) {
        // The conversion kernel stores straight into the page-locked slot: coalesced 512-byte runs over PCIe, 51 GB/s
        // measured at 2^20 boards (a copy engine behind a staging buffer, experiment grid_copy=1: 36 GB/s)
        static const bool direct = bgs::experiment("grid_copy") == nullptr;
        if (direct) {
            bgs::connect_cell_planes(b, reinterpret_cast<uint64_t*>(s->mapped[slot]));
            return hipGetLastError();
        }
        bgs::connect_cell_planes(b, reinterpret_cast<uint64_t*>(b->d_staging));  // (staging: ordered on the batch's stream)
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        src = b->d_staging;
    }
    return hipMemcpyAsync(s->pinned[slot], src, s->slot_bytes, hipMemcpyDeviceToHost, b->stream);
}

static int grid_sink_matches(const bgs_reward_sink* s, const bgs_batch* b) {
    const CellFormat f = cell_format(b);
    NEED(b->n == s->max_games && f.cells == s->cells.cells && f.sets == s->cells.sets,
         "the grid sink was made for batches of %lld boards of %d cells", (long long)s->max_games, s->cells.cells);
    NEED(f.sets == 0 || b->staging_bytes >= s->slot_bytes, "staging buffer too small");
    return BGS_OK;
}

int bgs_expand_grid_host(const void* wire, int64_t n, int cells, int sets, int offset, const int32_t* weights, int64_t first,
                         int64_t count, int8_t* grid, int portable) {
    NEED(wire != nullptr && grid != nullptr && n >= 1 && cells >= 1, "bad argument");
    NEED(sets == 0 || (sets >= 1 && sets <= 4 && weights != nullptr), "sets must be 0 (the wire is the grid) or 1..4 with weights");
    NEED(first >= 0 && count >= 0 && first + count <= n, "games [first, first + count) must lie inside the batch");
    CellFormat f;
    f.cells = cells;
    f.sets = sets;
    f.nwc = (cells + 63) / 64;
    f.offset = offset;
    for (int p = 0; p < sets; ++p) f.weight[p] = weights[p];
    if (sets && (portable || !g_have_avx512bw)) expand_cells_portable(static_cast<const uint64_t*>(wire), n, f, first, count, grid);
    else expand_cells(wire, n, f, first, count, grid);
    return BGS_OK;
}

int bgs_sink_set_progress(bgs_reward_sink* s, int64_t* word) {
    NEED(s != nullptr, "sink is NULL");
    NEED(word == nullptr || (reinterpret_cast<uintptr_t>(word) & 7u) == 0, "progress word must be 8-byte aligned");
    std::lock_guard<std::mutex> lock(s->mu);
    s->progress = word;
    if (word) progress_store_max(word, s->completed);
    return BGS_OK;
}

int bgs_sink_completed(bgs_reward_sink* s, int64_t* completed) {
    NEED(s != nullptr && completed != nullptr, "NULL argument");
    *completed = s->a_completed.load(std::memory_order_acquire);
    return BGS_OK;
}

int bgs_progress_store(int64_t* word, int64_t value) {
    NEED(word != nullptr && (reinterpret_cast<uintptr_t>(word) & 7u) == 0, "progress word must be 8-byte aligned");
    progress_store_max(word, value);
    return BGS_OK;
}

int bgs_progress_wait(const int64_t* words, int64_t count, int64_t stride_words, int64_t target, int64_t timeout_ms,
                      int64_t* laggard) {
    NEED(words != nullptr && count >= 1 && stride_words >= 1, "bad argument");
    NEED((reinterpret_cast<uintptr_t>(words) & 7u) == 0, "progress words must be 8-byte aligned");
    struct timespec start;
    clock_gettime(CLOCK_MONOTONIC, &start);
    for (int64_t i = 0; i < count; ++i) {
        auto* a = reinterpret_cast<const std::atomic<int64_t>*>(words + i * stride_words);
        for (;;) {
            const int64_t seen = a->load(std::memory_order_acquire);
            if (seen >= target) break;
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            const int64_t waited_ms = (now.tv_sec - start.tv_sec) * 1000 + (now.tv_nsec - start.tv_nsec) / 1000000;
            if (timeout_ms >= 0 && waited_ms >= timeout_ms) {
                if (laggard) *laggard = i;
                return bgs::fail(BGS_ERR_RUNTIME, "progress word %lld is at %lld, waiting for %lld: timed out after %lld ms",
                                 (long long)i, (long long)seen, (long long)target, (long long)waited_ms);
            }
            // sleep until the low half changes (or 50 ms pass: a store by a process that died is never announced)
            int64_t slice_ms = 50;
            if (timeout_ms >= 0 && timeout_ms - waited_ms < slice_ms) slice_ms = timeout_ms - waited_ms;
            if (slice_ms < 1) slice_ms = 1;
            const struct timespec ts = {(time_t)(slice_ms / 1000), (long)(slice_ms % 1000) * 1000000L};
            futex(const_cast<int64_t*>(words + i * stride_words), FUTEX_WAIT, (uint32_t)seen, &ts);
        }
    }
    return BGS_OK;
}

int bgs_progress_barrier(int64_t* words, int64_t count, int64_t stride_words, int64_t mine, int64_t epoch, int64_t spin_us,
                         int64_t timeout_ms) {
    NEED(words != nullptr && count >= 1 && stride_words >= 1 && mine >= 0 && mine < count, "bad argument");
    NEED((reinterpret_cast<uintptr_t>(words) & 7u) == 0, "progress words must be 8-byte aligned");
    progress_store_max(words + mine * stride_words, epoch);   // (wakes whoever already sleeps on this word)
    // the others are a few microseconds away when the ranks run in step: watch their words for a while before sleeping
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us > 0 ? spin_us : 0);
    for (int64_t i = 0; i < count; ++i) {
        auto* a = reinterpret_cast<const std::atomic<int64_t>*>(words + i * stride_words);
        while (a->load(std::memory_order_acquire) < epoch) {
            if (std::chrono::steady_clock::now() >= until)
                return bgs_progress_wait(words, count, stride_words, epoch, timeout_ms, nullptr);
            for (int k = 0; k < 16; ++k) _mm_pause();
        }
    }
    return BGS_OK;
}

int bgs_bind_host_thread(int device, int* cpus_out) {
    cpu_set_t cpus;
    int n = 0;
    if (device_node_cpus(device, &cpus)) {
        if (sched_setaffinity(0, sizeof cpus, &cpus) != 0) return fail(BGS_ERR_RUNTIME, "sched_setaffinity failed");
        n = CPU_COUNT(&cpus);
    }
    if (cpus_out) *cpus_out = n;
    return BGS_OK;
}

int bgs_sink_destroy(bgs_reward_sink* s) {
    if (!s) return BGS_OK;
    {
        std::unique_lock<std::mutex> lock(s->mu);
        s->cv_done.wait(lock, [&] { return s->completed == s->claimed; });  // let claimed jobs finish
        s->stop = true;
        s->a_stop.store(true, std::memory_order_release);
    }
    s->cv_submit.notify_all();
    s->cv_landed.notify_all();
    for (auto& w : s->workers) w.join();
    (void)hipSetDevice(s->device);
    for (auto p : s->pinned) (void)hipHostFree(p);
    for (auto e : s->landed) (void)hipEventDestroy(e);
    delete s;
    return BGS_OK;
}

int bgs_sink_submit(bgs_reward_sink* s, bgs_batch* b, int8_t* host_reward, int64_t* ticket) {
    NEED(s != nullptr && b != nullptr && host_reward != nullptr, "NULL argument");
    NEED(b->device == s->device, "batch lives on device %d, the sink on device %d", b->device, s->device);
    NEED(b->n <= s->max_games, "batch of %lld games exceeds the sink's %lld", (long long)b->n, (long long)s->max_games);
    int rc = enter_device(s->device);
    if (rc) return rc;
    if (s->grids && (rc = grid_sink_matches(s, b))) return rc;
    const int64_t t = claim(s);
    const int slot = (int)(t % s->slots);
    hipError_t err;
    if (s->grids) {
        err = enqueue_grids(s, b, slot);
    } else {
        bgs::pack_outcomes(b, s->mapped[slot]);  // the codes go straight into the page-locked slot
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipEventRecord(s->landed[slot], b->stream);
    publish(s, t, b->n, host_reward, err == hipSuccess);
    if (ticket) *ticket = t;
    if (err != hipSuccess) return fail(BGS_ERR_RUNTIME, "reward hand-over could not be enqueued: %s", hipGetErrorString(err));
    return BGS_OK;
}

int bgs_sink_rollout(bgs_reward_sink* s, bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags,
                     int8_t* host_reward, int64_t* ticket) {
    NEED(s != nullptr && b != nullptr && host_reward != nullptr, "NULL argument");
    NEED(b->device == s->device, "batch lives on device %d, the sink on device %d", b->device, s->device);
    NEED(b->n <= s->max_games, "batch of %lld games exceeds the sink's %lld", (long long)b->n, (long long)s->max_games);
    if (s->grids) {
        int rc0 = grid_sink_matches(s, b);
        if (rc0) return rc0;
    }
    // what bgs_rollout would refuse is refused HERE, before a ticket exists: an argument error is the caller's, not a
    // failed delivery
    NEED(max_plies >= 0, "max_plies must be >= 0");
    {
        int rc0 = enter_device(s->device);
        if (rc0) return rc0;
    }
    const int64_t t = claim(s);
    const int slot = (int)(t % s->slots);
    // the rollout kernel stores the outcome codes of the games it finishes straight into the page-locked slot (or the
    // pack kernel does, for kernels without that epilogue): when the event fires the codes are in host memory.  A grid
    // sink: the final boards follow the rollout in the wire format, one asynchronous copy
    int rc = s->grids ? bgs_rollout(b, seed, max_plies, flags) : bgs::rollout_with_codes(b, seed, max_plies, flags, s->mapped[slot]);
    hipError_t err = hipSuccess;
    if (rc == BGS_OK && s->grids) err = enqueue_grids(s, b, slot);
    if (rc == BGS_OK && err == hipSuccess) err = hipEventRecord(s->landed[slot], b->stream);
    publish(s, t, b->n, host_reward, rc == BGS_OK && err == hipSuccess);
    if (ticket) *ticket = t;
    if (rc) return rc;
    if (err != hipSuccess) return fail(BGS_ERR_RUNTIME, "reward hand-over could not be enqueued: %s", hipGetErrorString(err));
    return BGS_OK;
}

int bgs_sink_submit_packed(bgs_reward_sink* s, void* hip_stream, const void* device_packed, int64_t n_games,
                           int8_t* host_reward, int64_t* ticket) {
    NEED(s != nullptr && device_packed != nullptr && host_reward != nullptr, "NULL argument");
    NEED(!s->grids, "a grid sink takes boards (bgs_sink_submit / bgs_sink_rollout), not outcome codes");
    NEED(n_games >= 1 && n_games <= s->max_games, "n_games %lld outside 1..%lld", (long long)n_games,
         (long long)s->max_games);
    int rc = enter_device(s->device);
    if (rc) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int64_t t = claim(s);
    const int slot = (int)(t % s->slots);
    hipError_t err = hipMemcpyAsync(s->pinned[slot], device_packed, (size_t)(n_games + 3) / 4, hipMemcpyDeviceToHost, stream);
    if (err == hipSuccess) err = hipEventRecord(s->landed[slot], stream);
    publish(s, t, n_games, host_reward, err == hipSuccess);
    if (ticket) *ticket = t;
    if (err != hipSuccess) return fail(BGS_ERR_RUNTIME, "reward hand-over could not be enqueued: %s", hipGetErrorString(err));
    return BGS_OK;
}

int bgs_sink_wait(bgs_reward_sink* s, int64_t ticket) { return bgs::sink_wait(s, ticket, false); }

}  // extern "C"
