// bounce_unit.h -- what the Bounce kernels and their launchers are made of besides bounce_kernels.hip itself: the geometry
// record the kernels take by value, the layout of the device-wide pool and the launch tuning.  Hashed (with bgs_common.h
// and the kernel source) into the Bounce unit's id (bgs_kernel_unit_id(1)).
#pragma once

#include <stdint.h>

enum {
    BGS_BOUNCE_POOL_GROUPS = 2048,  // K3p: workgroups of a launch that can park boards in the device-wide pool
    BGS_BOUNCE_POOL_WORDS = 4 + 2 * 2048 + 2048 * 64 * 6,  // dwords: counters, per-group count / head, 64 entries of 6 dwords a group
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

constexpr int kBouncePiecesPark = 32;       // K3p with the device-wide pool, 20 in flight / one launch at a time, x 10^9: round 4 32 / 40 / 48 / 56 / 63 = 11.8 / 12.2 / 11.9 / 11.8 / 3.0;
                                            // round 5 (opening book): 17.57 / 17.47 / 17.48 / 17.30 / 3.85 pipelined, 3.17 / 3.00 / 2.69 / 0.30 / 0.09 alone
constexpr int kBouncePark = 32;             // flat Bounce rollout: see ParkedBoards (0 = every wave drains alone)
// defaults that change what a launch executes (here and not in bgs_capi.hip so that the unit's id moves with them)
constexpr int kBounceBlock = 256;           // threads a workgroup of the flat / piece-list rollouts
constexpr int kBounceFlatChunk = 32;        // flat rollout: boards a wave draws from the queue at a time
constexpr int kBounceFlatWps = 2;           // flat rollout: waves per SIMD
constexpr int kBounceMemoCold = 4;          // K3w: consecutive memo misses after which a game plays without the memo ...
constexpr int kBounceMemoBypass = 28;       // ... for this many plies, then looks again
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

