// bgs_pipeline.hip -- the rollout loop as a native executor (include/bgs.h, bgs_pipeline_*).
//
// The reference's loop is `while not state.has_ended: state = random.choice(state.actions).sample_next_state()`
// (README.md:45-72), one board at a time under the GIL.  The batched form of "many such games, one after the other" is
//   step s: batch (s mod depth) plays all its boards from the initial state to the end with seed seed0 + s, on its own
//           stream, and the step's rewards go to host array (j mod n_host), j = number of hand-overs so far,
// and at 2^20 Connect4 boards a step is ~34 us of GPU time: a Python loop that makes one ctypes call per step spends
// 27 us per step on the launching thread and is launch-bound on short runs.  This executor makes the whole loop ONE
// library call: it enqueues `count` steps back to back (~ a kernel launch and an event record each), blocks only when
// the host array a step is about to reuse is still being delivered, and brackets a sample of the launches with timing
// events for the roofline figures.  The hand-over is whatever the pipeline was given: a reward sink (one GPU, or N
// ranks delivering into a shared host array) or a reward gather (RCCL to rank 0).
#include <cstdlib>
#include <new>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <immintrin.h>
#include <string>
#include <vector>

#include "bgs_capi_util.h"
#include "bgs_common.h"
#include "bgs_internal.h"

using bgs::fail;

// A thread that synchronises ONE stream when a drain says so.  hipStreamSynchronize is a marker's round trip through the
// GPU even on a stream that has run dry (~15 us), and whichever synchronising call ends the caller's region pays it
// once per stream, one after the other (3 streams: 40-45 us at the end of a 0.7 ms region).  With one of these per
// stream the round trips run side by side and BEHIND the last kernels, while the sink still expands the last delivery:
// the draining thread only waits for the delivery and for their flags, and the caller's own hipDeviceSynchronize finds
// idle streams (5 us).  round 3, r3_drain.sh in the git history: 20-step regions 6.48 -> 6.83 x 10^11, last delivery expanded -> device
// synchronised 55 -> 22 us.  The threads sleep between drains.
struct StreamSyncer {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int cmd = 0;                // (under mu) 0 asleep, 1 go, 2 exit
    std::atomic<int> done{1};   // 0 while a synchronise is in flight
    std::atomic<int> err{0};
    int device = 0;
    hipStream_t stream = nullptr;
    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return cmd != 0; });
                if (cmd == 2) return;
                cmd = 0;
            }
            if (hipStreamSynchronize(stream) != hipSuccess) err.store(1, std::memory_order_relaxed);
            done.store(1, std::memory_order_release);
        }
    }
    void go(hipStream_t s) {
        stream = s;
        done.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lock(mu); cmd = 1; }
        cv.notify_one();
    }
    bool wait() {   // true: the stream is idle; false: the synchronise failed
        // (a few hundred microseconds of watching the flag -- the end of a run -- then the core is given back: a stream
        // that still holds milliseconds of kernels is waited for by its helper, inside the runtime)
        for (int spins = 0; !done.load(std::memory_order_acquire); ++spins) {
            if (spins < 20000) _mm_pause();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        return err.exchange(0) == 0;
    }
};

struct bgs_pipeline {
    int device = 0;
    std::vector<bgs_batch*> batches;   // [depth]
    bgs_reward_sink* sink = nullptr;
    bgs_gather* gather = nullptr;
    std::vector<int8_t*> host;         // [n_host] destinations (NULL entries: a gather rank other than 0)
    std::vector<int64_t> ticket;       // [n_host] the hand-over last delivered into that array, -1 = none pending
    std::vector<int64_t> held;         // [n_host] index of the hand-over whose rewards that array holds (or will hold), -1 = none
    uint64_t seed0 = 0;
    int32_t max_plies = 0x7FFFFFFF;
    uint32_t flags = BGS_ROLLOUT_FROM_INITIAL;
    int64_t step = 0;                  // steps enqueued so far (seed of a step = seed0 + its index)
    int64_t handed = 0;                // hand-overs enqueued so far (host array of a hand-over = its index mod n_host)
    // shared-array hand-shake (bgs_pipeline_set_ring)
    const int64_t* rank_words = nullptr;   // progress word of rank r at rank_words[r * word_stride]
    int64_t word_stride = 8;
    int world = 1;
    int64_t* consumed = nullptr;
    bool consumer = false;
    int lag = 0;
    int64_t timeout_ms = 60000;
    std::vector<std::unique_ptr<StreamSyncer>> syncers;   // [depth], started by the first drain of a pipeline deeper than 1
    // timing brackets
    std::vector<hipEvent_t> ev0, ev1;
    size_t brackets = 0;
    // The FEEDER (bgs_pipeline_feed / _release; round 5): a thread of the pipeline's own that enqueues the steps a caller has
    // fed -- one seed each -- as soon as the host array a step lands in has been RELEASED by the consumer, so that a consumer
    // loop in a slow language (simulator.pipeline.RolloutPipeline.run: a generator in Python) only waits for a hand-over,
    // reads it and releases it -- three cheap calls -- while the launches are made beside it.  `mu` guards step / handed /
    // ticket / held / the feeder's own fields whenever a feeder exists.
    std::mutex mu;
    std::condition_variable cv;
    std::thread feeder;
    std::deque<uint64_t> fed;     // seeds the feeder has not taken yet
    int64_t fed_end = 0;           // the hand-over index one past the last fed step: the steps [handed, fed_end) are fed and
                                   // not enqueued yet -- waiting in `fed`, or the ONE the feeder is enqueuing right now
    int64_t released = 0;          // hand-overs [0, released) are the consumer's no more: their arrays may be overwritten
    bool feeding = false;          // the feeder thread exists
    bool stop = false;
    int feed_rc = BGS_OK;          // the first failure of an enqueue made by the feeder ...
    std::string feed_error;        // ... and its message (reported by the consumer's next call)
};

extern "C" {

int bgs_pipeline_create(bgs_batch* const* batches, int depth, bgs_reward_sink* sink, bgs_gather* gather,
                        int8_t* const* host_rewards, int n_host, uint64_t seed0, int32_t max_plies, uint32_t flags,
                        bgs_pipeline** out) {
    NEED(out != nullptr && batches != nullptr, "NULL argument");
    *out = nullptr;
    NEED(depth >= 1 && depth <= 64, "depth must be in 1..64");
    NEED(!(sink && gather), "a pipeline hands over through a sink OR a gather");
    NEED(max_plies >= 0, "max_plies must be >= 0");
    NEED((sink || gather) ? (host_rewards != nullptr && n_host >= 1 && n_host <= 256) : n_host == 0,
         "a hand-over needs 1..256 host arrays, a pipeline without one takes none");
    for (int k = 0; k < depth; ++k) {
        NEED(batches[k] != nullptr, "batch %d is NULL", k);
        NEED(batches[k]->device == batches[0]->device && batches[k]->n == batches[0]->n, "the batches must be alike");
        for (int j = 0; j < k; ++j) NEED(batches[j] != batches[k], "batch %d is listed twice", k);
    }
    bgs_pipeline* p = new (std::nothrow) bgs_pipeline();
    NEED(p != nullptr, "out of host memory");
    p->device = batches[0]->device;
    p->batches.assign(batches, batches + depth);
    for (bgs_batch* b : p->batches) b->launches_in_flight = depth;  // (the launch shape follows it: bgs_set_launches_in_flight)
    p->sink = sink;
    p->gather = gather;
    for (int h = 0; h < n_host; ++h) p->host.push_back(host_rewards[h]);
    p->ticket.assign(n_host, -1);
    p->held.assign(n_host, -1);
    p->seed0 = seed0;
    p->max_plies = max_plies;
    p->flags = flags;
    *out = p;
    return BGS_OK;
}

int bgs_pipeline_set_ring(bgs_pipeline* p, const int64_t* rank_words, int64_t word_stride, int world, int64_t* consumed,
                          int is_consumer, int lag, int64_t timeout_ms) {
    NEED(p != nullptr, "pipeline is NULL");
    NEED(rank_words == nullptr || (world >= 1 && word_stride >= 1 && consumed != nullptr), "bad ring description");
    NEED(!is_consumer || (lag >= 1 && lag < (int)p->host.size()),
         "the consumer trails the launches by 1 .. (host arrays - 1) hand-overs");
    p->rank_words = rank_words;
    p->word_stride = word_stride;
    p->world = world;
    p->consumed = consumed;
    p->consumer = is_consumer != 0;
    p->lag = lag;
    p->timeout_ms = timeout_ms;
    return BGS_OK;
}

static int wait_ticket(bgs_pipeline* p, int64_t t, bool urgent = false) {
    return p->sink ? bgs::sink_wait(p->sink, t, urgent) : bgs::gather_wait(p->gather, t, urgent);
}

// the consumer's side of the shared-array hand-shake: every rank has delivered hand-over j -> release its slot
static int consume(bgs_pipeline* p, int64_t j) {
    int rc = bgs_progress_wait(p->rank_words, p->world, p->word_stride, j + 1, p->timeout_ms, nullptr);
    if (rc) return rc;
    return bgs_progress_store(p->consumed, j + 1);
}

static int enqueue_steps(bgs_pipeline* p, const uint64_t* seeds, int64_t count, int handover, int time_stride, bool from_feeder = false) {
    NEED(p != nullptr && count >= 0, "bad argument");
    NEED(!handover || p->sink || p->gather, "this pipeline has no hand-over");
    // (with a feeder the consumer's bgs_pipeline_wait reads ticket / held beside this loop: the bookkeeping of a step is
    // made under the pipeline's lock, the waits and the launches outside it)
    std::unique_lock<std::mutex> lock(p->mu);
    const bool fed_pending = p->feeding && p->handed < p->fed_end && p->feed_rc == BGS_OK;
    lock.unlock();
    NEED(from_feeder || !fed_pending, "steps are being fed to this pipeline (bgs_pipeline_feed): enqueue when they have all been consumed");
    HIP_TRY(hipSetDevice(p->device));
    const int depth = (int)p->batches.size();
    const int n_host = (int)p->host.size();
    for (int64_t i = 0; i < count; ++i) {
        bgs_batch* b = p->batches[p->step % depth];
        const uint64_t seed = seeds ? seeds[i] : p->seed0 + (uint64_t)p->step;
        size_t bracket = (size_t)-1;
        if (time_stride > 0 && i % time_stride == 0) {
            if (p->brackets == p->ev0.size()) {
                hipEvent_t a = nullptr, z = nullptr;
                HIP_TRY(hipEventCreate(&a));
                p->ev0.push_back(a);
                HIP_TRY(hipEventCreate(&z));
                p->ev1.push_back(z);
            }
            bracket = p->brackets;
            HIP_TRY(hipEventRecord(p->ev0[bracket], b->stream));
        }
        int rc;
        if (handover) {
            const int64_t j = p->handed;
            const int h = (int)(j % n_host);
            lock.lock();
            const int64_t before = p->ticket[h];
            lock.unlock();
            if (before >= 0) {   // the array is about to be overwritten: its previous delivery must be over
                if ((rc = wait_ticket(p, before))) return rc;
                lock.lock();
                if (p->ticket[h] == before) p->ticket[h] = -1;
                lock.unlock();
            }
            if (p->rank_words) {
                if (p->consumer && j - p->lag >= 0 && (rc = consume(p, j - p->lag))) return rc;
                // ... and the consumer must have released what this hand-over overwrites
                if (j >= n_host && (rc = bgs_progress_wait(p->consumed, 1, 1, j - n_host + 1, p->timeout_ms, nullptr))) return rc;
            }
            int64_t t = -1;
            rc = p->sink ? bgs_sink_rollout(p->sink, b, seed, p->max_plies, p->flags, p->host[h], &t)
                         : bgs_gather_rollout(p->gather, b, seed, p->max_plies, p->flags, p->host[h], &t);
            if (rc) return rc;
            // the ring's progress words count the sink's deliveries: hand-over j must be the sink's job j
            NEED(!p->rank_words || t == j, "a pipeline on a shared array needs a sink of its own (ticket %lld for hand-over %lld)",
                 (long long)t, (long long)j);
            lock.lock();
            p->ticket[h] = t;
            p->held[h] = j;
            ++p->handed;
            if (!from_feeder) p->released = p->handed;   // (a caller who enqueues himself answers for his arrays himself)
            lock.unlock();
            if (p->feeding) p->cv.notify_all();          // (a consumer may be waiting for this hand-over to exist)
        } else {
            if ((rc = bgs_rollout(b, seed, p->max_plies, p->flags))) return rc;
        }
        if (bracket != (size_t)-1) {
            // (the rollout kernels that deliver the outcome codes themselves leave nothing but an event record between
            // the brackets; for the others the bracket includes the pack kernel)
            HIP_TRY(hipEventRecord(p->ev1[bracket], b->stream));
            ++p->brackets;
        }
        lock.lock();
        ++p->step;
        lock.unlock();
    }
    return BGS_OK;
}

// the feeder thread: one fed seed at a time, as soon as its host array has been released.  The seed LEAVES the deque under
// the lock that finds it there (nobody else ever sees a seed that is being enqueued: a destroy or a drain beside this
// thread counts pending steps with fed_end, not with the deque's size)
static void feeder_loop(bgs_pipeline* p) {
    (void)hipSetDevice(p->device);
    const int64_t n_host = (int64_t)p->host.size();
    for (;;) {
        uint64_t seed;
        {
            std::unique_lock<std::mutex> lock(p->mu);
            p->cv.wait(lock, [&] { return p->stop || (!p->fed.empty() && p->feed_rc == BGS_OK && p->handed - p->released < n_host); });
            if (p->stop) return;
            seed = p->fed.front();
            p->fed.pop_front();
        }
        const int rc = enqueue_steps(p, &seed, 1, 1, 0, true);
        if (rc != BGS_OK) {
            std::lock_guard<std::mutex> lock(p->mu);
            if (p->feed_rc == BGS_OK) {
                p->feed_rc = rc;
                p->feed_error = bgs_last_error();
            }
        }
        p->cv.notify_all();
    }
}

int bgs_pipeline_enqueue(bgs_pipeline* p, int64_t count, int handover, int time_stride) {
    return enqueue_steps(p, nullptr, count, handover, time_stride);
}

int bgs_pipeline_enqueue_seeds(bgs_pipeline* p, const uint64_t* seeds, int64_t count, int handover) {
    NEED(seeds != nullptr || count == 0, "seeds is NULL");
    return enqueue_steps(p, seeds, count, handover, 0);
}

int bgs_pipeline_feed(bgs_pipeline* p, const uint64_t* seeds, int64_t count) {
    NEED(p != nullptr && count >= 0 && (seeds != nullptr || count == 0), "bad argument");
    NEED(p->sink || p->gather, "this pipeline has no hand-over");
    NEED(!p->rank_words, "a pipeline on a shared array is driven by bgs_pipeline_enqueue");
    {
        std::lock_guard<std::mutex> lock(p->mu);
        if (p->feed_rc != BGS_OK) return fail(p->feed_rc, "%s", p->feed_error.c_str());
        if (!p->feeding) {
            p->released = p->handed;   // (what was enqueued before is the caller's business)
            p->feeding = true;
            p->feeder = std::thread([p] { feeder_loop(p); });
        }
        if (p->fed_end < p->handed) p->fed_end = p->handed;   // (steps the caller enqueued himself since the last feed)
        p->fed_end += count;
        p->fed.insert(p->fed.end(), seeds, seeds + count);
    }
    p->cv.notify_all();
    return BGS_OK;
}

int bgs_pipeline_release(bgs_pipeline* p, int64_t handover_index) {
    NEED(p != nullptr, "pipeline is NULL");
    {
        std::lock_guard<std::mutex> lock(p->mu);
        NEED(handover_index >= 0 && handover_index < p->handed, "hand-over %lld has not been enqueued", (long long)handover_index);
        if (handover_index + 1 > p->released) p->released = handover_index + 1;
    }
    p->cv.notify_all();
    return BGS_OK;
}

int bgs_pipeline_wait(bgs_pipeline* p, int64_t handover_index) {
    NEED(p != nullptr, "pipeline is NULL");
    int64_t t;
    int h;
    {
        std::unique_lock<std::mutex> lock(p->mu);
        // a fed step exists once the feeder has enqueued it
        if (p->feeding && handover_index >= p->handed && handover_index < p->fed_end)
            p->cv.wait(lock, [&] { return handover_index < p->handed || p->feed_rc != BGS_OK || p->stop; });
        if (p->feed_rc != BGS_OK && handover_index >= p->handed) return fail(p->feed_rc, "%s", p->feed_error.c_str());
        NEED(handover_index >= 0 && handover_index < p->handed, "hand-over %lld has not been enqueued", (long long)handover_index);
        h = (int)(handover_index % (int64_t)p->host.size());
        NEED(p->held[h] == handover_index, "hand-over %lld is not in flight any more: its host array was reused by hand-over %lld",
             (long long)handover_index, (long long)p->held[h]);
        t = p->ticket[h];
    }
    if (t < 0) return BGS_OK;   // waited for before
    int rc = wait_ticket(p, t);
    if (rc == BGS_OK) {
        std::lock_guard<std::mutex> lock(p->mu);
        if (p->ticket[h] == t) p->ticket[h] = -1;
    }
    return rc;
}

int bgs_pipeline_drain(bgs_pipeline* p) {
    NEED(p != nullptr, "pipeline is NULL");
    HIP_TRY(hipSetDevice(p->device));
    int rc;
    if (p->feeding) {
        // what has been fed is enqueued first; every array is released (a drain ends the consumer's claim on them)
        std::unique_lock<std::mutex> lock(p->mu);
        if (p->fed_end > p->released) p->released = p->fed_end;
        p->cv.notify_all();
        p->cv.wait(lock, [&] { return p->handed >= p->fed_end || p->feed_rc != BGS_OK || p->stop; });
        p->released = p->handed;   // (never past what exists: the next fed step waits for ITS array's release again)
        if (p->feed_rc != BGS_OK) return fail(p->feed_rc, "%s", p->feed_error.c_str());
    }
    // the newest hand-over first: every waiter it turns urgent stays so until the last delivery is in
    int64_t newest = -1;
    for (int64_t t : p->ticket) newest = t > newest ? t : newest;
    // every stream is synchronised by its helper, starting NOW (see StreamSyncer); experiment drain_serial_sync=1: by this
    // thread, one after the other, once the deliveries are in (the A/B of round 3, r3_drain.sh in the git history)
    const size_t depth = p->batches.size();
    static const bool serial = bgs::experiment("drain_serial_sync") != nullptr;
    const bool helpers = depth > 1 && !serial;
    if (helpers) {
        while (p->syncers.size() < depth) {
            p->syncers.emplace_back(new StreamSyncer());
            StreamSyncer* h = p->syncers.back().get();
            h->device = p->device;
            h->th = std::thread([h] { h->run(); });
        }
        for (size_t k = 0; k < depth; ++k) p->syncers[k]->go(p->batches[k]->stream);
    }
    auto streams_idle = [&]() -> int {   // (on every path out: a helper must not be left inside a synchronise)
        bool ok = true;
        for (size_t k = 0; k < depth; ++k) {
            if (helpers) ok = p->syncers[k]->wait() && ok;
            else ok = hipStreamSynchronize(p->batches[k]->stream) == hipSuccess && ok;
        }
        return ok ? BGS_OK : fail(BGS_ERR_RUNTIME, "hipStreamSynchronize failed in bgs_pipeline_drain");
    };
    if (newest >= 0 && (rc = wait_ticket(p, newest, true))) {   // (deliveries complete in ticket order)
        (void)streams_idle();
        return rc;
    }
    for (size_t h = 0; h < p->ticket.size(); ++h) p->ticket[h] = -1;
    // the consumer sees every hand-over of every rank before it calls the region done
    if (p->rank_words && p->consumer && p->handed > 0 && (rc = consume(p, p->handed - 1))) {
        (void)streams_idle();
        return rc;
    }
    // steps without hand-over (and everything else the batches have enqueued): their streams run dry
    return streams_idle();
}

int bgs_pipeline_progress(const bgs_pipeline* p, int64_t* steps, int64_t* handovers) {
    NEED(p != nullptr, "pipeline is NULL");
    if (steps) *steps = p->step;
    if (handovers) *handovers = p->handed;
    return BGS_OK;
}

int bgs_pipeline_kernel_ms(bgs_pipeline* p, double* mean_ms, int* pairs) {
    NEED(p != nullptr && mean_ms != nullptr, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    double total = 0.0;
    for (size_t k = 0; k < p->brackets; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(p->ev1[k]));  // (recorded just behind a delivered hand-over: a formality)
        HIP_TRY(hipEventElapsedTime(&ms, p->ev0[k], p->ev1[k]));
        total += ms;
    }
    *mean_ms = p->brackets ? total / (double)p->brackets : 0.0;
    if (pairs) *pairs = (int)p->brackets;
    p->brackets = 0;  // the events are reused by the next timed region
    return BGS_OK;
}

int bgs_pipeline_timeline(bgs_pipeline* p, float* start_ms, float* end_ms, int capacity, int* pairs) {
    NEED(p != nullptr && start_ms != nullptr && end_ms != nullptr && capacity >= 0, "bad argument");
    HIP_TRY(hipSetDevice(p->device));
    const int count = (int)p->brackets < capacity ? (int)p->brackets : capacity;
    for (int k = 0; k < count; ++k) {
        HIP_TRY(hipEventSynchronize(p->ev1[k]));
        HIP_TRY(hipEventElapsedTime(&start_ms[k], p->ev0[0], p->ev0[k]));
        HIP_TRY(hipEventElapsedTime(&end_ms[k], p->ev0[0], p->ev1[k]));
    }
    if (pairs) *pairs = count;
    return BGS_OK;
}

int bgs_pipeline_destroy(bgs_pipeline* p) {
    if (!p) return BGS_OK;
    (void)hipSetDevice(p->device);
    if (p->feeding) {   // (steps fed and not yet enqueued are dropped: nobody is there to consume them)
        {
            std::lock_guard<std::mutex> lock(p->mu);
            p->stop = true;
        }
        p->cv.notify_all();
        if (p->feeder.joinable()) p->feeder.join();   // (a step it was enqueuing is enqueued; the drain below waits for it)
        p->fed.clear();
        p->fed_end = p->handed;
        p->feeding = false;
    }
    (void)bgs_pipeline_drain(p);
    for (auto& h : p->syncers) {
        { std::lock_guard<std::mutex> lock(h->mu); h->cmd = 2; }
        h->cv.notify_one();
        if (h->th.joinable()) h->th.join();
    }
    for (auto e : p->ev0) (void)hipEventDestroy(e);
    for (auto e : p->ev1) (void)hipEventDestroy(e);
    delete p;
    return BGS_OK;
}

}  // extern "C"
