/* A plain C99 host of libbgs.so: what a non-Python embedder of the C ABI (include/bgs.h) writes.  Plays one batch of
 * Connect4(6,7,4) games to the end, receives the rewards in page-locked host memory through the asynchronous hand-over,
 * runs the native rollout loop over two batches, and prints counts the Python test compares with the CPU oracle.  Built and run by tests/test_gpu_c_abi.py. */
#include <stdio.h>
#include <stdlib.h>

#include "bgs.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != BGS_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bgs_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const uint64_t seed = 0x0123456789ABCDEFull;
    bgs_batch* batch = NULL;
    bgs_event* done = NULL;
    bgs_reward_sink* sink = NULL;
    void* pinned = NULL;
    int8_t* by_sink = (int8_t*)malloc((size_t)n * 2);
    uint64_t steps = 0;
    int64_t ticket = -1, wins0 = 0, wins1 = 0, draws = 0, mismatches = 0;

    CHECK(bgs_connect_create(6, 7, 4, n, 0, NULL, 0, &batch));
    CHECK(bgs_set_first_game(batch, 1000));
    CHECK(bgs_host_alloc((size_t)n * 2, &pinned));
    CHECK(bgs_event_create(0, &done));
    /* 1: rollout + asynchronous copy of the int8 reward pairs */
    CHECK(bgs_rollout_to_host(batch, seed, 0x7FFFFFFF, BGS_ROLLOUT_FROM_INITIAL, pinned, 0, done));
    CHECK(bgs_event_synchronize(done));
    CHECK(bgs_steps(batch, &steps));
    /* 2: the same games again through a reward sink (outcome codes + host worker threads) */
    CHECK(bgs_sink_create(0, n, 2, 2, &sink));
    CHECK(bgs_sink_rollout(sink, batch, seed, 0x7FFFFFFF, BGS_ROLLOUT_FROM_INITIAL, by_sink, &ticket));
    CHECK(bgs_sink_wait(sink, ticket));
    const int8_t* r = (const int8_t*)pinned;
    for (int64_t i = 0; i < n; ++i) {
        if (r[2 * i] == 1 && r[2 * i + 1] == -1) ++wins0;
        else if (r[2 * i] == -1 && r[2 * i + 1] == 1) ++wins1;
        else if (r[2 * i] == 0 && r[2 * i + 1] == 0) ++draws;
        else ++mismatches;
        if (r[2 * i] != by_sink[2 * i] || r[2 * i + 1] != by_sink[2 * i + 1]) ++mismatches;
    }
    /* 3: the native rollout loop (bgs_pipeline_*): two batches on streams of their own, five steps with seeds seed - 4 ..
     * seed, every step's rewards handed over; the last host array must hold the rewards of `seed` again */
    {
        bgs_batch* two[2] = {NULL, NULL};
        void* streams[2] = {NULL, NULL};
        bgs_reward_sink* loop_sink = NULL;
        bgs_pipeline* loop = NULL;
        int8_t* hosts[3];
        int64_t enqueued = 0, handed = 0;
        float start_ms[8], end_ms[8];
        int pairs = 0;
        for (int k = 0; k < 3; ++k) hosts[k] = (int8_t*)malloc((size_t)n * 2);
        for (int k = 0; k < 2; ++k) {
            CHECK(bgs_stream_create(0, &streams[k]));
            CHECK(bgs_connect_create(6, 7, 4, n, 0, NULL, 0, &two[k]));
            CHECK(bgs_set_stream(two[k], streams[k]));
            CHECK(bgs_set_first_game(two[k], 1000));
            CHECK(bgs_set_launches_in_flight(two[k], 2));
        }
        CHECK(bgs_sink_create(0, n, 3, 2, &loop_sink));
        CHECK(bgs_pipeline_create(two, 2, loop_sink, NULL, hosts, 3, seed - 4, 0x7FFFFFFF, BGS_ROLLOUT_FROM_INITIAL, &loop));
        CHECK(bgs_pipeline_enqueue(loop, 5, 1, 1));
        CHECK(bgs_pipeline_drain(loop));
        CHECK(bgs_pipeline_progress(loop, &enqueued, &handed));
        CHECK(bgs_pipeline_timeline(loop, start_ms, end_ms, 8, &pairs));
        if (enqueued != 5 || handed != 5 || pairs != 5 || !(end_ms[4] > start_ms[4])) ++mismatches;
        for (int64_t i = 0; i < 2 * n; ++i)
            if (hosts[4 % 3][i] != by_sink[i]) { ++mismatches; break; }
        CHECK(bgs_pipeline_destroy(loop));
        CHECK(bgs_sink_destroy(loop_sink));
        for (int k = 0; k < 2; ++k) {
            CHECK(bgs_destroy(two[k]));
            CHECK(bgs_stream_destroy(0, streams[k]));
        }
        for (int k = 0; k < 3; ++k) free(hosts[k]);
    }
    printf("C_ABI n=%lld steps=%llu wins0=%lld wins1=%lld draws=%lld mismatches=%lld build=%s version=%d\n", (long long)n,
           (unsigned long long)steps, (long long)wins0, (long long)wins1, (long long)draws, (long long)mismatches,
           bgs_build_id(), bgs_version());
    CHECK(bgs_sink_destroy(sink));
    CHECK(bgs_event_destroy(done));
    CHECK(bgs_host_free(pinned));
    CHECK(bgs_destroy(batch));
    free(by_sink);
    return 0;
}
