// bgs_internal.h -- host-side batch object and kernel launcher prototypes (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

struct bgs_reward_sink;  // bgs_host.hip
struct bgs_gather;       // bgs_multi.hip

enum { BGS_GAME_CONNECT = 1, BGS_GAME_BOUNCE = 2 };

// (the limits of the packed representations live in bgs_common.h, the geometry records and the launch tuning of a game in
// connect_unit.h / bounce_unit.h -- the headers the kernel units' ids hash; this one is host-side plumbing and is not hashed)
#include "connect_unit.h"
#include "bounce_unit.h"

struct bgs_batch {
    int game;
    int generic;             // 1: the board is an int8 grid in the reference layout, played by generic_kernels.hip
    int gen_h, gen_w, gen_k;
    uint32_t gen_init_status;
    int device;
    int64_t n;
    hipStream_t stream;
    uint64_t first_game;
    ConnectGeom cg;
    BounceGeom bg;
    int planes;              // uint64 planes per board
    int num_cus;             // compute units of the device
    int rollout_wps;         // waves per SIMD the fused rollout is sized for
    int bounce_group;        // lanes per board of a single-launch Bounce rollout: 8 (small batches) or 1 (experiment bounce_group)
    int bounce_group_auto;   // 1: not set from the environment (bounce_rollout may still choose by the launches in flight)
    int bounce_flat;         // 1: one-lane-per-board Bounce rollouts use the flattened search (experiment bounce_flat=0: nested loops)
    int bounce_pieces;       // 1: from-initial flat rollouts run on the piece list (K3p; experiment bounce_pieces=0: K3f)
    int bounce_block;        // K3p: threads per workgroup, 256 / 512 / 1024 (experiment bounce_block): the waves of a workgroup share their drain
    int bounce_flat_wps;     // waves per SIMD of a flat Bounce rollout launch (experiment bounce_flat_wps)
    int bounce_flat_waves;   // > 0: that many waves per launch instead (experiment bounce_flat_waves)
    int bounce_pool;         // K3p: the last wave of a workgroup parks its boards for other workgroups (experiment bounce_pool=0: off)
    int bounce_pieces_park;  // K3p: a draining wave parks its boards at this many or fewer (0..63; experiment bounce_pieces_park, else experiment bounce_park, else -1 = bounce_shape().park)
    int bounce_park;         // flat rollout: a draining wave parks its boards for its workgroup at this many or fewer (0..32; experiment bounce_park)
    int bounce_flat_chunk;   // boards a wave draws from the work queue at a time (experiment bounce_chunk)
    int launches_in_flight;  // the caller's hint (bgs_set_launches_in_flight), 1 = one launch at a time: see bounce_shape()
    int bounce_memo_cold, bounce_memo_bypass;  // K3w: after this many look-ups in a row that missed, this many plies without the memo (experiment "bounce_memo_policy=cold:plies")
    int bounce_epoch_limit;  // K3w: the memo starts over after this many replacements (0 = the 16 bits a link has; experiment bounce_epoch_limit: the tests' way to reach the restart)
    int bounce_wave_pass;    // 1: the automatic plan ends with the one-board-per-wave pass (K3w); experiment bounce_wave_pass=0 switches it off
    int bounce_plan_auto;    // 1: the library chooses between one launch and bulk + tail passes (experiment bounce_plan unset or "auto")
    int bounce_passes;       // multi-pass Bounce rollout: number of passes, their ply caps and lanes per board
    uint32_t bounce_pass_cap[BGS_BOUNCE_MAX_PASSES];
    int bounce_pass_group[BGS_BOUNCE_MAX_PASSES];
    int bounce_wave_grid;    // > 0: one-wave workgroups of a K3w launch (experiment bounce_wave_grid; 0: 8192)
    int transition_wave;     // 1: the object API's one-board transition runs on one wave, a piece per lane (experiment transition_wave=0: thread per board)
    int bounce_static_geom;  // 1: the default board is played by the kernels instantiated on its compile-time geometry (experiment bounce_static_geom=0: the run-time record)
    int bounce_tail;         // 1: the games beyond K3p's ply cap are finished by a tail kernel that runs BESIDE it (experiment bounce_tail=0: a pass behind it); -1: by the launch shape
    int bounce_tail_handoff; // >= 0: a workgroup's last wave hands its boards to the tail queue at this many or fewer (experiment; -1: bounce_shape())
    int bounce_tail_prio;    // s_setprio of the tail kernel's waves (experiment bounce_tail_prio, 0..3)
    int bounce_tail_limit;   // > 0: tail waves that may wait for entries at a time (experiment; 0: bounce_shape())
    enum { kTailStages = 4 };
    mutable hipStream_t tail_stream[kTailStages];   // the streams of the staged tail launches beside the bulk kernel (created at the first such rollout)
    mutable hipEvent_t tail_fork, tail_bulk_done, tail_join[kTailStages];
    mutable uint32_t tail_serial;     // the "entry complete" word of the last fused launch (the region of d_worklist holds such words)
    mutable int tail_flags_dirty;     // 1: d_worklist holds something else (game indices of a compaction pass, nothing yet)
    int rollout_generic;     // 1: never take the block-aligned from-initial kernel (A/B timing, experiment rollout_generic)
    int rollout_chunk;       // games per wave of the fused rollout, 0 = derived from rollout_wps (experiment rollout_chunk)
    int rollout_opening;     // opening blocks of the from-initial one-word rollout: 0 = K2a, 1..4, default 3 (experiment rollout_opening)
    int rollout_no_lds;      // 1: large boards stay in registers (K2b) instead of the LDS-staged kernel (experiment rollout_no_lds)
    int rng_per_ply;         // Connect: 1 = the strict RNG contract, a philox word per ply (bgs_set_rng_contract); 0 = a word per four plies
    // device buffers (inside the arena)
    void* arena;
    size_t arena_bytes;
    bool owns_arena;
    uint64_t* d_planes;      // [planes][n]
    uint8_t* d_status;       // [n]
    uint16_t* d_plies;       // [n] (Bounce)
    int8_t* d_reward;        // [n][2]
    unsigned long long* d_steps;
    uint8_t* d_staging;
    size_t staging_bytes;
    uint64_t* d_gen_masks;   // generic Bounce: [6][16] cell masks (all, interior, x > 0, x < w - 1, top row, bottom row)
    int8_t* d_gen_cfg;       // generic Bounce: the configured start grid
    uint32_t* d_worklist;    // [n] board indices still to play (Bounce multi-pass rollout)
    uint32_t* d_work_count;  // [2 * BGS_BOUNCE_MAX_PASSES] list lengths, then work-queue heads; device-resident
    uint32_t* d_pool;        // packed Bounce: [BGS_BOUNCE_POOL_WORDS] the piece-list rollout's device-wide pool of parked boards
    // packed Bounce, large batches: the opening book of the start position (bounce_kernels.hip; shared by the batches of a
    // process that have the same start position on the same device, bgs::bounce_book_acquire / release)
    const uint32_t* book_links;
    const void* book_table;
    int book_depth;          // plies a new game skips (0: no book)
    uint32_t book_n0;        // actions at the start position
    void* book_owner;
    // pinned bounce buffers for large device -> host copies (allocated on first use)
    void* pinned[2];
    hipEvent_t pinned_done[2];
    hipEvent_t order_event;  // orders the batch's work across a change of stream (bgs_set_stream)
};

namespace bgs {

// The A/B switches and test hooks of the library -- kernels no automatic plan selects, fault injection, measurement knobs --
// live behind ONE environment variable: BGS_EXPERIMENT="name=value;name=value" (names as in tools/README.md; a value may
// hold commas and colons).  Returns the value of `name` (valid until the calling thread's next call) or NULL.  What a user
// of the library may want to set has a variable of its own and is listed in INTEGRATION.md section G.
// The PRODUCT library (libbgs.so) is built without them (round 6): experiment() is an inline nullptr there, the names of
// the switches do not exist in the binary, the fault-injection code is not compiled, and BGS_EXPERIMENT is never read.  The
// TEST build (libbgs_test.so: the host-side units compiled with -DBGS_TEST_HOOKS, linked with the SAME kernel objects -- same
// kernel unit ids) is what tests/ and tools/ load when they force a kernel family or inject a fault (tests/conftest.py,
// tests/knobs.py).  The kernel translation units never call experiment(): every switch reaches them as a field of the batch.
#ifdef BGS_TEST_HOOKS
const char* experiment(const char* name);
#else
inline const char* experiment(const char*) { return nullptr; }
#endif

// ---- Connect (connect_kernels.hip) ----
void connect_reset(const bgs_batch* b);
void connect_step_random(const bgs_batch* b, uint64_t seed, uint32_t count);  // count plies per board
void connect_step_actions(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out);
bool connect_step_observe(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out, uint8_t* d_legal, uint8_t* d_ended,
                          int8_t* d_reward_out, bool auto_reset);
void connect_reset_ended(const bgs_batch* b);   // boards that have ended -> the initial state (packed boards)
void bounce_reset_ended(const bgs_batch* b);
void status_to_ended(const bgs_batch* b, uint8_t* d_ended);  // uint8[n]: the board has ended (any game)
void connect_transition(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out, int8_t* d_grid, int8_t* d_player,
                        int8_t* d_winner, int32_t* d_plies, uint8_t* d_legal, int8_t* d_reward_out, uint32_t* d_done = nullptr,
                        uint32_t ticket = 0);  // d_done: host word the one-workgroup kernel sets to `ticket` behind its records
bool connect_rollout(const bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, uint32_t* codes_out);
void connect_unpack_grid(const bgs_batch* b, int8_t* d_grid);
void connect_cell_planes(const bgs_batch* b, uint64_t* d_dst);  // wire format of the grid hand-over (see the kernel)
void connect_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies);
void connect_legal(const bgs_batch* b, uint8_t* d_legal, int32_t* d_count);
void connect_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                  int32_t* d_status_out);

// ---- Bounce (bounce_kernels.hip) ----
void bounce_reset(const bgs_batch* b);
void bounce_step_random(const bgs_batch* b, uint64_t seed, uint32_t count);
void bounce_step_actions(const bgs_batch* b, const int32_t* d_moves, int32_t* d_status_out);
void bounce_transition(const bgs_batch* b, const int32_t* d_moves, int32_t* d_status_out, int8_t* d_grid, int8_t* d_player,
                       int8_t* d_winner, int32_t* d_plies, uint64_t* d_targets, int8_t* d_reward_out, uint32_t* d_done = nullptr,
                       uint32_t ticket = 0);
void bounce_rollout(const bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags);
// the opening book of the batch's start position (see bounce_kernels.hip): built on the batch's stream the first time a
// start position is seen on a device, shared afterwards; 0 or a HIP error code
int bounce_book_acquire(bgs_batch* b, int max_depth);
void bounce_book_release(bgs_batch* b);
void bounce_unpack_grid(const bgs_batch* b, int8_t* d_grid);
void bounce_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies);
void bounce_targets(const bgs_batch* b, uint64_t* d_targets, int32_t* d_count);
void bounce_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                 const int32_t* d_plies, int32_t* d_status_out);

// ---- any geometry (generic_kernels.hip): the batch's "planes" region holds int8[n][h][w] ----
size_t generic_bounce_legal_bytes(int h, int w);  // bytes per board of the wide legal-move record
void generic_reset(const bgs_batch* b);
void generic_play(const bgs_batch* b, uint64_t seed, uint32_t max_plies, uint32_t count, bool from_initial, bool per_ply = false);
void generic_step_actions(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out);
void generic_unpack_grid(const bgs_batch* b, int8_t* d_grid);
void generic_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies);
void generic_connect_legal(const bgs_batch* b, uint8_t* d_legal, int32_t* d_count);
void generic_bounce_targets(const bgs_batch* b, uint8_t* d_wide, int32_t* d_count);
void generic_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                  const int32_t* d_plies, int32_t* d_status_out);

// ---- shared by the C-ABI translation units (bgs_capi.hip, bgs_host.hip) ----
// status bytes -> 2-bit outcome codes, 4 boards per byte, enqueued on the batch's stream
void pack_outcomes(const bgs_batch* b, uint8_t* d_packed);
// bgs_rollout with the outcome codes delivered to `codes_out` (16-byte aligned, (n + 63) / 64 * 16 bytes; device or
// device-mapped host memory): by the rollout kernel itself where it can, by k_pack_outcomes behind it otherwise
int rollout_with_codes(bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, uint8_t* codes_out);


// ---- the reward sink as the in-library gather drives it (bgs_host.hip) ----
// claim reserves the next ticket (blocking while every slot is in use); the caller then fills the ticket's slot --
// sink_slot_device is the slot as the GPU sees it (device-mapped page-locked memory), sink_slot_host as the CPU does --
// records sink_slot_event behind whatever fills it, and publishes the job (ok = false: the enqueue failed)
int64_t sink_claim(bgs_reward_sink* s);
uint8_t* sink_slot_device(bgs_reward_sink* s, int64_t ticket);
uint8_t* sink_slot_host(bgs_reward_sink* s, int64_t ticket);
hipEvent_t sink_slot_event(bgs_reward_sink* s, int64_t ticket);
// event_ticket >= 0: the job's bytes have arrived when THAT ticket's slot event fires (a group of jobs behind one record)
// all_end: every game of the job must have ended (an uncapped rollout from the start): a code 0 among them fails the job
void sink_publish(bgs_reward_sink* s, int64_t ticket, int64_t n_games, int8_t* host_reward, bool ok, int64_t event_ticket = -1,
                  bool all_end = false);
int sink_wait(bgs_reward_sink* s, int64_t ticket, bool urgent);  // urgent: poll / spin (the end of a run)
int gather_wait(bgs_gather* g, int64_t ticket, bool urgent);     // bgs_multi.hip

}  // namespace bgs
