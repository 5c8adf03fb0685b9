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

// ---- the DEFAULT board as a compile-time geometry (round 6).  The reference's own Bounce game -- 9 x 6, rows 1 and 7 hold the
// pieces 1 2 3 3 2 1 (textual/bounce.py:66-78) -- is BASELINE config 4; what Geo<1, 6, 7, 4> is for Connect, this is for
// Bounce: every mask a literal (round 5's K3p reloaded the geometry's masks, pointers and piece values from spilled scalar
// registers 290 times an iteration: BounceGeom is 50 dwords of kernel argument), the piece count and every piece's value
// constants (the segment loops unroll to their own length, the "is there a piece k" tests vanish).  The constants are
// COMPUTED by the same rule as bgs_capi.hip's bounce_geom() builds the run-time record, and a batch is served by the static
// instantiation exactly when its record equals default_bounce_geom() member for member (bounce_is_default()).
constexpr BounceGeom make_bounce_geom(const int8_t (&cfg)[64], int h, int w) {
    BounceGeom g{};
    g.h = h;
    g.w = w;
    g.inv_w = (65536u + (uint32_t)w - 1u) / (uint32_t)w;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int c = y * w + x;
            const uint64_t bit = 1ull << c;
            g.all |= bit;
            if (y == 0) g.goal_bottom |= bit;
            else if (y == h - 1) g.goal_top |= bit;
            else g.interior |= bit;
            if (x > 0) g.not_col0 |= bit;
            if (x < w - 1) g.not_collast |= bit;
            for (int p = 0; p < 4; ++p)
                if ((cfg[c] >> p) & 1) g.init[p] |= bit;
        }
    int k = 0;
    for (int v = 1; v <= 15; ++v)
        for (int c = 0; c < h * w; ++c)
            if (cfg[c] == v && k < 16) {
                g.piece_value[k] = (uint8_t)v;
                g.piece_cell[k] = (uint8_t)c;
                for (int p = 0; p < 4; ++p)
                    if ((k >> p) & 1) g.piece_idx[p] |= 1ull << c;
                ++k;
            }
    g.piece_count = (uint32_t)k;
    return g;
}
constexpr BounceGeom default_bounce_geom() {
    int8_t cfg[64] = {};
    const int8_t row[6] = {1, 2, 3, 3, 2, 1};
    for (int x = 0; x < 6; ++x) cfg[1 * 6 + x] = cfg[7 * 6 + x] = row[x];
    return make_bounce_geom(cfg, 9, 6);
}
constexpr BounceGeom kDefaultBounce = default_bounce_geom();
static_assert(kDefaultBounce.piece_count == 12 && kDefaultBounce.piece_value[0] == 1 && kDefaultBounce.piece_value[11] == 3 &&
              kDefaultBounce.piece_cell[0] == 7 - 1 && kDefaultBounce.init_status == 0, "the default board: twelve pieces");

#define BGS_DB(m) kDefaultBounce.m
struct DefaultBounceGeom {
    static constexpr int h = BGS_DB(h), w = BGS_DB(w);
    static constexpr uint32_t inv_w = BGS_DB(inv_w);
    static constexpr uint64_t all = BGS_DB(all), interior = BGS_DB(interior), goal_top = BGS_DB(goal_top), goal_bottom = BGS_DB(goal_bottom);
    static constexpr uint64_t not_col0 = BGS_DB(not_col0), not_collast = BGS_DB(not_collast);
    static constexpr uint64_t init[4] = {BGS_DB(init[0]), BGS_DB(init[1]), BGS_DB(init[2]), BGS_DB(init[3])};
    static constexpr uint32_t init_status = 0;
    static constexpr uint32_t piece_count = BGS_DB(piece_count);
    static constexpr uint8_t piece_value[16] = {BGS_DB(piece_value[0]), BGS_DB(piece_value[1]), BGS_DB(piece_value[2]), BGS_DB(piece_value[3]),
                                                BGS_DB(piece_value[4]), BGS_DB(piece_value[5]), BGS_DB(piece_value[6]), BGS_DB(piece_value[7]),
                                                BGS_DB(piece_value[8]), BGS_DB(piece_value[9]), BGS_DB(piece_value[10]), BGS_DB(piece_value[11]),
                                                BGS_DB(piece_value[12]), BGS_DB(piece_value[13]), BGS_DB(piece_value[14]), BGS_DB(piece_value[15])};
    static constexpr uint8_t piece_cell[16] = {BGS_DB(piece_cell[0]), BGS_DB(piece_cell[1]), BGS_DB(piece_cell[2]), BGS_DB(piece_cell[3]),
                                               BGS_DB(piece_cell[4]), BGS_DB(piece_cell[5]), BGS_DB(piece_cell[6]), BGS_DB(piece_cell[7]),
                                               BGS_DB(piece_cell[8]), BGS_DB(piece_cell[9]), BGS_DB(piece_cell[10]), BGS_DB(piece_cell[11]),
                                               BGS_DB(piece_cell[12]), BGS_DB(piece_cell[13]), BGS_DB(piece_cell[14]), BGS_DB(piece_cell[15])};
    static constexpr uint64_t piece_idx[4] = {BGS_DB(piece_idx[0]), BGS_DB(piece_idx[1]), BGS_DB(piece_idx[2]), BGS_DB(piece_idx[3])};
};
#undef BGS_DB
// the batch's record IS the default board's (every member the kernels read)
inline bool bounce_is_default(const BounceGeom& g) {
    const BounceGeom& d = kDefaultBounce;
    bool same = g.h == d.h && g.w == d.w && g.inv_w == d.inv_w && g.all == d.all && g.interior == d.interior && g.goal_top == d.goal_top &&
                g.goal_bottom == d.goal_bottom && g.not_col0 == d.not_col0 && g.not_collast == d.not_collast &&
                g.init_status == d.init_status && g.piece_count == d.piece_count;
    for (int k = 0; k < 4; ++k) same = same && g.init[k] == d.init[k] && g.piece_idx[k] == d.piece_idx[k];
    for (int k = 0; k < 16; ++k) same = same && g.piece_value[k] == d.piece_value[k] && g.piece_cell[k] == d.piece_cell[k];
    return same;
}

// K3p's park threshold is part of the launch shape since round 6 (BounceShape::park): with the device-wide pool, 20 in flight /
// one launch at a time, x 10^9: round 4 32 / 40 / 48 / 56 / 63 = 11.8 / 12.2 / 11.9 / 11.8 / 3.0; round 5 (opening book): 17.57 /
// 17.47 / 17.48 / 17.30 / 3.85 pipelined, 3.17 / 3.00 / 2.69 / 0.30 / 0.09 alone; round 6 (compile-time geometry): 8 / 16 / 32 =
// 20.4 / 21.6 / 22.8 with 20 in flight, 16.9 / 18.7 / 20.3 with 8 (16 hardware queues), 5.10 / 5.07 / 4.84 alone (1024 waves)
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
// Round 6 (the tail inside the bulk launch): handoff_at = a workgroup's last wave hands its last boards to the tail queue at
// this many or fewer; tail_waves = waves that may wait for tail entries at a time.
// Round 6 (compile-time geometry, the bulk pass writes the tail's work list itself): alone, 2^18 boards, ms a launch --
//   waves 768 / 1024 / 1280 / 1536 / 2048 / 3072 at park 16: 1.60 / 1.48 / 1.54 / 1.54 / 1.62 / 1.90; caps 64 / 80 / 96 / 112 / 128 /
//   160 (1536 waves): 1.66 / 1.58 / 1.61 / 1.66 / 1.73 / 1.83; park 4 / 8 / 16 / 32 (1024 waves): 1.454 / 1.448 / 1.456 / 1.524.
struct BounceShape { int tail_cap; int boards_per_wave; int handoff_at; int tail_waves; int park; };
inline BounceShape bounce_shape(int launches_in_flight) {
    // (20 in flight, 32 hardware queues, compile-time geometry, x 10^10: caps 128 / 160 / 192 / 224 / 256 / 320 at 256 waves -- / -- /
    // 2.38 / 2.39 / 2.40 / 2.38; waves 256 / 320 / 384 / 512 / 768 / 1024 at cap 256: 2.40 / 2.33 / 2.37 / 2.34 / 2.27 / 2.21; round 5's
    // {160, 512}: 2.29)
    if (launches_in_flight >= 12) return {240, 1024, 4, 512, 32};
    if (launches_in_flight >= 4) return {128, 256, 8, 1024, 32};
    return {80, 256, 16, 2048, 8};   // (one launch at a time, K3w on 8192 waves: caps of 64 / 80 / 96 / 112 / 128 read 2.78 / 2.86 / 2.80 / 2.69 / 2.62 x 10^9)
}

