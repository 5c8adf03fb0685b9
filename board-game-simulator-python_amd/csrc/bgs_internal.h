// bgs_internal.h -- host-side batch object and kernel launcher prototypes (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

struct bgs_reward_sink;  // bgs_host.hip
struct bgs_gather;       // bgs_multi.hip

enum { BGS_GAME_CONNECT = 1, BGS_GAME_BOUNCE = 2 };

// limits of the packed representations
enum {
    BGS_CONNECT_MAX_WORDS = 3,   // width * (height + 1) <= 192 bits per plane
    BGS_CONNECT_MAX_H = 15,      // column heights are kept 4 bits per column
    BGS_CONNECT_MAX_W = 16,
    BGS_BOUNCE_MAX_CELLS = 64,   // height * width <= 64: one bit per cell in a uint64
    BGS_BOUNCE_MAX_VALUE = 15,   // 4 value bit-planes
    BGS_BOUNCE_MAX_PASSES = 8,   // passes of the multi-pass Bounce rollout
    BGS_BOUNCE_MAX_PIECES = 16,  // piece-list rollout kernel (K3p): pieces on the configured start position
    BGS_BOUNCE_POOL_GROUPS = 2048,  // K3p: workgroups of a launch that can park boards in the device-wide pool
    BGS_BOUNCE_POOL_WORDS = 4 + 2 * 2048 + 2048 * 64 * 6,  // dwords: counters, per-group count / head, 64 entries of 6 dwords a group
    // the generic (reference-layout) kernels take over beyond the packed limits
    BGS_GENERIC_CONNECT_MAX_DIM = 64,      // height, width <= 64 (the oracle's own limit: nothing larger can be checked)
    BGS_GENERIC_BOUNCE_MAX_CELLS = 1024,   // height * width <= 1024, piece values <= 127 (int8)
    BGS_GENERIC_BOUNCE_MAX_DIM = 64
};

struct ConnectGeom {
    int h, w, k, nw;   // nw = 64-bit words per plane
};

struct BounceGeom {
    int h, w;
    uint32_t inv_w;          // ceil(2^16 / w): y = (cell * inv_w) >> 16 for cell < 64
    uint64_t all;            // every cell
    uint64_t interior;       // rows 1 .. h-2
    uint64_t goal_top;       // row h-1 (player 0's goal)
    uint64_t goal_bottom;    // row 0   (player 1's goal)
    uint64_t not_col0;       // cells with x > 0
    uint64_t not_collast;    // cells with x < w-1
    uint64_t init[4];        // value bit-planes of the configured start position
    uint32_t init_status;    // 0, or the terminal code of a start position without legal moves
    // The start position as a piece list (Bounce never captures and never changes a piece's value, so a board IS the
    // cells of its pieces): piece k has value piece_value[k] and starts on cell piece_cell[k]; pieces are numbered by
    // ascending (value, cell).  piece_count = 0: more than BGS_BOUNCE_MAX_PIECES pieces (the piece-list rollout is off).
    uint32_t piece_count;
    uint8_t piece_value[16];
    uint8_t piece_cell[16];
    uint64_t piece_idx[4];   // index planes of the start position: bit c of plane p = bit p of the index of the piece on cell c
};

// Launch tuning of the fused rollouts.  These live here (a header the build id hashes) and not in bgs_capi.hip because
// they change what a launch executes: counters taken under one setting must not be quoted for another.
constexpr int kRolloutOpeningBlocks = 3;    // K2o: 4-ply blocks played in lock step before a board joins the refill loop
constexpr int kGamesPerLaneOneWord = 8;     // one-word Connect boards: games per lane a launch aims for (512 per wave at 2^20)
constexpr int kGamesPerLane = 4;            // every other rollout
constexpr int kBouncePiecesPark = 40;       // K3p with the device-wide pool: 32 / 40 / 48 / 56 / 63 read 11.8 / 12.2 / 11.9 / 11.8 / 3.0 x 10^9 with 20 in flight
constexpr int kBouncePark = 32;             // flat Bounce rollout: see ParkedBoards (0 = every wave drains alone)
// K3p, automatic plan: the shape of a launch follows the number of launches the caller keeps in flight on the device
// (bgs_set_launches_in_flight; the rollout executor passes its depth).  tail_cap: games longer than this are finished
// by the tail pass; boards_per_wave: boards a wave of the bulk pass plays.  Alone on the chip a launch is bound by its
// longest chain of dependent plies (17 us a ply on the piece-list kernel, 0.65 us on the 8-lanes-per-board kernel of
// the tail) and by how many SIMDs it reaches, so: short bulk, many waves.  With 16 launches sharing the chip what counts
// is instructions per ply, so: few long-lived waves that stay full, and a bulk pass long enough to keep the tail small.
// 2^18 boards, 10^9 env-steps/s (round 3, r3_bounce_solo.sh in the git history, r3_bounce_depth.sh, r3_bounce_depth2.sh):
//   in flight        1      4      8      16
//   {384, 512}     1.11   2.16   6.05   9.7      (round 3's only shape until then)
//   {64, 128}      1.92   3.10   6.53   7.7
//   {128, 256}     1.63   2.92   6.85   8.8
//   {160, 512}     1.12    --     --   10.2
// Round 4, with the one-board-per-wave tail pass (K3w, with its memo and links) behind the bulk pass
// (tools/k3w_depth_probe.sh, GPU_MAX_HW_QUEUES=24):
//   in flight        2      4      6      8      12     16     20
//   {64, 128}      4.77   7.54   8.02   8.21   8.36   8.43   8.50
//   {128, 256}     4.88   8.88  12.24  13.98  14.11  14.29  14.44
//   {160, 512}     3.65   6.89   9.55  12.17  15.72  15.79  15.86
struct BounceShape { int tail_cap; int boards_per_wave; };
inline BounceShape bounce_shape(int launches_in_flight) {
    if (launches_in_flight >= 12) return {160, 512};
    if (launches_in_flight >= 4) return {128, 256};
    return {80, 128};   // (one launch at a time, K3w on 8192 waves: caps of 64 / 80 / 96 / 112 / 128 read 2.78 / 2.86 / 2.80 / 2.69 / 2.62 x 10^9)
}

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
    int bounce_group;        // lanes per board of a single-launch Bounce rollout: 8 (small batches) or 1 (BGS_BOUNCE_GROUP)
    int bounce_group_auto;   // 1: not set from the environment (bounce_rollout may still choose by the launches in flight)
    int bounce_flat;         // 1: one-lane-per-board Bounce rollouts use the flattened search (BGS_BOUNCE_FLAT=0: nested loops)
    int bounce_pieces;       // 1: from-initial flat rollouts run on the piece list (K3p; BGS_BOUNCE_PIECES=0: K3f)
    int bounce_block;        // K3p: threads per workgroup, 256 / 512 / 1024 (BGS_BOUNCE_BLOCK): the waves of a workgroup share their drain
    int bounce_flat_wps;     // waves per SIMD of a flat Bounce rollout launch (BGS_BOUNCE_FLAT_WPS)
    int bounce_flat_waves;   // > 0: that many waves per launch instead (BGS_BOUNCE_FLAT_WAVES)
    int bounce_pool;         // K3p: the last wave of a workgroup parks its boards for other workgroups (BGS_BOUNCE_POOL=0: off)
    int bounce_pieces_park;  // K3p: a draining wave parks its boards at this many or fewer (0..63; BGS_BOUNCE_PIECES_PARK, default: BGS_BOUNCE_PARK)
    int bounce_park;         // flat rollout: a draining wave parks its boards for its workgroup at this many or fewer (0..32; BGS_BOUNCE_PARK)
    int bounce_flat_chunk;   // boards a wave draws from the work queue at a time (BGS_BOUNCE_CHUNK)
    int launches_in_flight;  // the caller's hint (bgs_set_launches_in_flight), 1 = one launch at a time: see bounce_shape()
    int bounce_wave_pass;    // 1: the automatic plan ends with the one-board-per-wave pass (K3w); BGS_BOUNCE_WAVE_PASS=0 switches it off
    int bounce_plan_auto;    // 1: the library chooses between one launch and bulk + tail passes (BGS_BOUNCE_PLAN unset or "auto")
    int bounce_passes;       // multi-pass Bounce rollout: number of passes, their ply caps and lanes per board
    uint32_t bounce_pass_cap[BGS_BOUNCE_MAX_PASSES];
    int bounce_pass_group[BGS_BOUNCE_MAX_PASSES];
    int rollout_generic;     // 1: never take the block-aligned from-initial kernel (A/B timing, BGS_ROLLOUT_GENERIC)
    int rollout_chunk;       // games per wave of the fused rollout, 0 = derived from rollout_wps (BGS_ROLLOUT_CHUNK)
    int rollout_opening;     // opening blocks of the from-initial one-word rollout: 0 = K2a, 1..4, default 3 (BGS_ROLLOUT_OPENING)
    int rollout_no_lds;      // 1: large boards stay in registers (K2b) instead of the LDS-staged kernel (BGS_ROLLOUT_NO_LDS)
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
    // pinned bounce buffers for large device -> host copies (allocated on first use)
    void* pinned[2];
    hipEvent_t pinned_done[2];
    hipEvent_t order_event;  // orders the batch's work across a change of stream (bgs_set_stream)
};

namespace bgs {

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
void bounce_unpack_grid(const bgs_batch* b, int8_t* d_grid);
void bounce_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies);
void bounce_targets(const bgs_batch* b, uint64_t* d_targets, int32_t* d_count);
void bounce_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                 const int32_t* d_plies, int32_t* d_status_out);

// ---- any geometry (generic_kernels.hip): the batch's "planes" region holds int8[n][h][w] ----
size_t generic_bounce_legal_bytes(int h, int w);  // bytes per board of the wide legal-move record
void generic_reset(const bgs_batch* b);
void generic_play(const bgs_batch* b, uint64_t seed, uint32_t max_plies, uint32_t count, bool from_initial);
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
void sink_publish(bgs_reward_sink* s, int64_t ticket, int64_t n_games, int8_t* host_reward, bool ok, int64_t event_ticket = -1);
int sink_wait(bgs_reward_sink* s, int64_t ticket, bool urgent);  // urgent: poll / spin (the end of a run)
int gather_wait(bgs_gather* g, int64_t ticket, bool urgent);     // bgs_multi.hip

}  // namespace bgs
