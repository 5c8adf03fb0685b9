// bounce_kernels.hip -- "Bounce" hot path on gfx950: active row -> reachable targets of every movable piece ->
// uniform (source, target) sample -> move -> goal / blocked / draw -> reward.
//
// Replaces, N boards per launch, what the reference reaches through
//   State::get_actions / get_actions_at / get_action_at  (src/simulator/game/bounce.cpp:40-42)
//   Action::sample_next_state                            (bounce.cpp:51)
//   State::has_ended / get_reward                        (bounce.cpp:36,38)
// Rules as pinned by reference tests/test_bounce.py:92-362 (plain restatement in
// oracle/bgs_oracle.c).
//
// Board packing.  height * width <= 64 cells, cell index c = y * width + x (y = 0 bottom row).  Piece values
// (1..15) are bit-sliced into four uint64 planes: plane j holds bit j of every cell's value; occupancy is the OR
// of the planes.  32 bytes per board, planes stored SoA over the batch.
//
// The move search is bit-parallel instead of a per-cell recursion: a segment of v steps is v rounds of three
// frontier masks (arrived by a forward / left / right step); a forward step shifts by +-width, a sideways step by
// +-1 under an edge mask, "no immediate left<->right reversal" is the rule that a left frontier only feeds forward
// and left.  Landing on a piece queues that cell; every queued cell is expanded once (its continuation does not
// depend on how it was reached), which is the closure the recursive search computes.
#include <type_traits>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "bgs_common.h"
#include "bgs_internal.h"

// identity of this translation unit as compiled: hash of this file, the kernel headers and the compile flags (csrc/Makefile)
#ifndef BGS_TU_ID
#define BGS_TU_ID "unknown"
#endif
extern "C" const char bgs_tu_id_bounce[] = BGS_TU_ID;

namespace bgs {
namespace {

struct Board {
    uint64_t v[4];
};

// plies are stored as uint16: a board that holds 65535 plies is not stepped any further (no wrap-around, no reuse
// of philox blocks); bgs_step_actions reports BGS_ERR_ILLEGAL for it
constexpr uint32_t kMaxPlies = 65535u;

__device__ __forceinline__ uint64_t occupancy(const Board& b) { return b.v[0] | b.v[1] | b.v[2] | b.v[3]; }

__device__ __forceinline__ uint32_t value_at(const Board& b, int c) {
    return (uint32_t)((b.v[0] >> c) & 1ull) | ((uint32_t)((b.v[1] >> c) & 1ull) << 1) |
           ((uint32_t)((b.v[2] >> c) & 1ull) << 2) | ((uint32_t)((b.v[3] >> c) & 1ull) << 3);
}

// position of the k-th set bit of m (k < popcount(m)) without a loop: a popcount-guided binary search.  The loop form
// (clear the lowest bit k times) costs a wave the largest k of its 64 lanes.
__device__ __forceinline__ uint32_t select_bit64(uint64_t m, uint32_t k) {
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const uint32_t below = (uint32_t)__popc(lo);
    const bool upper = k >= below;
    uint32_t word = upper ? hi : lo, pos = upper ? 32u : 0u;
    k -= upper ? below : 0u;
#pragma unroll
    for (uint32_t half = 16u; half >= 1u; half >>= 1) {
        const uint32_t cnt = (uint32_t)__popc(word & ((1u << half) - 1u));
        const bool up = k >= cnt;
        word = up ? word >> half : word;
        pos += up ? half : 0u;
        k -= up ? cnt : 0u;
    }
    return pos;
}

// every legal landing cell of the piece on cell `src` for `player` (SURVEY Appendix B rules 4-5).
// Walkers only ever stand on interior cells (the start piece, empty interior cells), so a forward step never leaves
// the board and needs no mask; forward is "<< w" for player 0 and ">> w" for player 1, written as two shifts one of
// which is by 0, so there is no per-lane select in the step.
__device__ __forceinline__ uint64_t reach(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player, int src) {
    const uint64_t empty_interior = ~occ & g.interior;
    const uint64_t landing = empty_interior | (player ? g.goal_bottom : g.goal_top);
    const uint64_t bounce_on = occ & g.interior;
    const uint32_t up = player ? 0u : (uint32_t)g.w, down = player ? (uint32_t)g.w : 0u;
    uint64_t pending = 1ull << src, done = 0, targets = 0;
    while (pending) {
        const int c = __ffsll((unsigned long long)pending) - 1;
        pending &= pending - 1;
        done |= 1ull << c;
        const uint32_t v = value_at(b, c);
        uint64_t a0 = 1ull << c, al = 0, ar = 0, land = 0;
        for (uint32_t s = 1; s <= v; ++s) {
            const uint64_t via_left = a0 | al, via_right = a0 | ar;  // who may go on left / right (no reversal)
            const uint64_t nf = ((via_left | ar) << up) >> down;
            const uint64_t nl = (via_left & g.not_col0) >> 1;
            const uint64_t nr = (via_right & g.not_collast) << 1;
            if (s < v) {
                a0 = nf & empty_interior;
                al = nl & empty_interior;
                ar = nr & empty_interior;
                if (!(a0 | al | ar)) break;
            } else {
                land = nf | nl | nr;
            }
        }
        targets |= land & landing;
        pending |= land & bounce_on & ~done;
    }
    return targets;
}

// pieces the side to move may pick: those in the occupied non-goal row nearest its own side (Appendix B rule 3)
template <class GEO>
__device__ __forceinline__ uint64_t movable(const GEO& g, uint64_t occ, uint32_t player) {
    const uint64_t oi = occ & g.interior;
    if (!oi) return 0;
    const int cell = player ? 63 - __clzll((long long)oi) : __ffsll((unsigned long long)oi) - 1;
    const int row = (int)(((uint32_t)cell * g.inv_w) >> 16);
    return oi & (((1ull << g.w) - 1ull) << (row * g.w));
}

__device__ __forceinline__ uint32_t count_actions(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player) {
    uint64_t src = movable(g, occ, player);
    uint32_t n = 0;
    while (src) {
        const int s = __ffsll((unsigned long long)src) - 1;
        src &= src - 1;
        n += (uint32_t)__popcll(reach(g, b, occ, player, s));
    }
    return n;
}

// the idx-th action of the canonical list: sources by ascending x, targets by ascending (y, x)
__device__ __forceinline__ void pick_action(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player,
                                            uint32_t idx, int& src_cell, int& dst_cell) {
    uint64_t src = movable(g, occ, player);
    src_cell = -1;
    dst_cell = -1;
    while (src) {
        const int s = __ffsll((unsigned long long)src) - 1;
        src &= src - 1;
        uint64_t t = reach(g, b, occ, player, s);
        const uint32_t cnt = (uint32_t)__popcll(t);
        if (idx < cnt) {
            src_cell = s;
            dst_cell = (int)select_bit64(t, idx);
            return;
        }
        idx -= cnt;
    }
}

__device__ __forceinline__ void move_piece(Board& b, int src_cell, int dst_cell) {
    const uint32_t v = value_at(b, src_cell);
    const uint64_t keep = ~(1ull << src_cell);
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[j] = (b.v[j] & keep) | ((uint64_t)((v >> j) & 1u) << dst_cell);
}

// terminal test after `mover` moved to dst_cell (Appendix B rule 7); n_next = action count of the next player
__device__ __forceinline__ uint32_t settle(const BounceGeom& g, const Board& b, uint32_t mover, int dst_cell,
                                           uint32_t& n_next) {
    n_next = 0;
    if ((1ull << dst_cell) & (g.goal_top | g.goal_bottom)) return mover + 1u;
    const uint64_t occ = occupancy(b);
    n_next = count_actions(g, b, occ, 1u - mover);
    if (n_next) return BGS_ST_RUNNING;
    return count_actions(g, b, occ, mover) ? mover + 1u : BGS_ST_DRAW;
}

// a board whose side to move has no action although nobody ended the game (start positions, loaded boards)
__device__ __forceinline__ uint32_t settle_blocked(const BounceGeom& g, const Board& b, uint32_t player) {
    return count_actions(g, b, occupancy(b), 1u - player) ? (1u - player) + 1u : BGS_ST_DRAW;
}

__device__ __forceinline__ Board load_board(const uint64_t* __restrict__ planes, int64_t n, int64_t i) {
    Board b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[j] = planes[(int64_t)j * n + i];
    return b;
}

__device__ __forceinline__ void store_board(uint64_t* __restrict__ planes, int64_t n, int64_t i, const Board& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) planes[(int64_t)j * n + i] = b.v[j];
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_reset(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies,
               uint16_t* __restrict__ reward, int64_t n, int only_ended) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (only_ended && status[i] == 0) return;   // (bgs_env_step with BGS_ENV_AUTO_RESET: finished boards start over)
#pragma unroll
    for (int j = 0; j < 4; ++j) planes[(int64_t)j * n + i] = g.init[j];
    status[i] = (uint8_t)g.init_status;
    plies[i] = 0;
    reward[i] = reward_pair(g.init_status);
}

// plies of one board, in registers, until it ends, reaches max_plies, or (SINGLE) one ply was played
template <bool SINGLE>
__device__ __forceinline__ uint32_t play(const BounceGeom& g, Board& b, uint32_t& st, uint32_t& plies, uint64_t seed,
                                         uint64_t game, uint32_t max_plies) {
    uint32_t stepped = 0;
    if (st != BGS_ST_RUNNING) return 0;
    uint32_t n_act = count_actions(g, b, occupancy(b), plies & 1u);
    if (n_act == 0) {
        st = settle_blocked(g, b, plies & 1u);
        return 0;
    }
    Philox4 blk = philox4x32_10(seed, game, plies >> 2);
    while (st == BGS_ST_RUNNING && plies < max_plies) {
        const uint32_t mover = plies & 1u;
        const uint32_t idx = sample_index(philox_word(blk, plies), n_act);
        int s, t;
        pick_action(g, b, occupancy(b), mover, idx, s, t);
        move_piece(b, s, t);
        ++plies;
        ++stepped;
        st = settle(g, b, mover, t, n_act);
        if (SINGLE) break;
        if ((plies & 3u) == 0u) blk = philox4x32_10(seed, game, plies >> 2);
    }
    return stepped;
}

template <bool SINGLE, bool FROM_INITIAL>
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_play(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
              uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
              unsigned long long* __restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        Board b;
        uint32_t st, plies;
        if (FROM_INITIAL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b.v[j] = g.init[j];
            st = g.init_status;
            plies = 0;
        } else {
            b = load_board(planes, n, i);
            st = status[i];
            plies = plies_buf[i];
        }
        const uint32_t st_in = st;
        stepped = play<SINGLE>(g, b, st, plies, seed, first_game + (uint64_t)i, max_plies);
        if (FROM_INITIAL || stepped || st != st_in) {
            store_board(planes, n, i, b);
            status[i] = (uint8_t)st;
            plies_buf[i] = (uint16_t)plies;
            reward[i] = reward_pair(st);
        }
    }
    add_steps(steps, stepped);
}

// ------------------------------------------------------------------------------------------------
// Fused rollout with lane refill (boards up to 8 columns wide).  Games last from 2 to several hundred plies, so a
// lane that finishes takes the next game of its wave's chunk at once.  The target masks of the side to move are
// kept in registers (one uint64 per column of the active row): the enumeration that decides whether the previous
// move blocked the opponent IS the next ply's action list, so every ply runs the move search once.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxTrackedColumns = 8;

struct Moves {
    uint64_t t[kMaxTrackedColumns];  // legal landing cells of the piece in column x of the active row (0 if none)
    uint32_t row_base;               // cell index of column 0 of the active row
    uint32_t n;                      // number of actions
};

__device__ __forceinline__ void enumerate(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player, Moves& m) {
    const uint64_t src = movable(g, occ, player);
    const int first = src ? __ffsll((unsigned long long)src) - 1 : 0;
    const int row = (int)(((uint32_t)first * g.inv_w) >> 16);
    m.row_base = (uint32_t)(row * g.w);
    m.n = 0;
#pragma unroll
    for (int x = 0; x < kMaxTrackedColumns; ++x) {
        const int c = (int)m.row_base + x;
        uint64_t t = 0;
        if (x < g.w && ((src >> (c & 63)) & 1ull)) t = reach(g, b, occ, player, c);
        m.t[x] = t;
        m.n += (uint32_t)__popcll(t);
    }
}

// The same list computed by the 8 lanes that share one board (lane-group mode): lane `sub` of the group searches the
// piece in column `sub` of the active row and KEEPS its mask; the group only exchanges counts (an 8-lane prefix sum)
// and, when a move is picked, the one (source, target) pair.  The move search of a board is a handful of independent
// walks; one lane runs them one after the other, eight lanes run them side by side -- a ply then costs one walk, which
// is what the latency-bound end of a batch needs.
struct GroupMoves {
    uint64_t mine;       // legal landing cells of the piece in this lane's column of the active row (0 if none)
    uint32_t before;     // actions of the columns left of this lane's
    uint32_t row_base;   // cell index of column 0 of the active row
    uint32_t n;          // number of actions of the board
};

__device__ __forceinline__ void enumerate_group(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player,
                                                GroupMoves& m) {
    const uint32_t sub = threadIdx.x & 7u;
    const uint64_t src = movable(g, occ, player);
    const int first = src ? __ffsll((unsigned long long)src) - 1 : 0;
    const int row = (int)(((uint32_t)first * g.inv_w) >> 16);
    m.row_base = (uint32_t)(row * g.w);
    const int c = (int)(m.row_base + sub) & 63;
    m.mine = 0;
    if (sub < (uint32_t)g.w && ((src >> c) & 1ull)) m.mine = reach(g, b, occ, player, c);
    const uint32_t cnt = (uint32_t)__popcll(m.mine);
    uint32_t incl = cnt;  // inclusive prefix sum over the group's 8 lanes
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 8);
        incl += sub >= (uint32_t)d ? up : 0u;
    }
    m.before = incl - cnt;
    m.n = (uint32_t)__shfl((int)incl, 7, 8);
}

// the idx-th action of the canonical list: the lane whose column holds it works it out, an OR over the group hands
// the pair to all eight lanes
__device__ __forceinline__ void pick_group(const GroupMoves& m, uint32_t idx, int& src_cell, int& dst_cell) {
    const uint32_t k = idx - m.before;
    const bool here = k < (uint32_t)__popcll(m.mine);  // (unsigned: idx < before wraps to a huge k)
    uint32_t pair = here ? ((m.row_base + (threadIdx.x & 7u)) | (select_bit64(m.mine, here ? k : 0u) << 8)) : 0u;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) pair |= (uint32_t)__shfl_xor((int)pair, d, 8);
    src_cell = (int)(pair & 255u);
    dst_cell = (int)(pair >> 8);
}

// the idx-th action of the canonical list (sources by ascending x, targets by ascending cell index)
__device__ __forceinline__ void pick_from(const Moves& m, uint32_t idx, int& src_cell, int& dst_cell) {
    uint64_t chosen = 0;
    uint32_t column = 0;
    bool found = false;
#pragma unroll
    for (int x = 0; x < kMaxTrackedColumns; ++x) {
        const uint32_t cnt = (uint32_t)__popcll(m.t[x]);
        const bool here = !found && idx < cnt;
        chosen = here ? m.t[x] : chosen;
        column = here ? (uint32_t)x : column;
        idx = (found || here) ? idx : idx - cnt;
        found = found || here;
    }
    src_cell = (int)(m.row_base + column);
    dst_cell = (int)select_bit64(chosen, idx);
}

// GL = lanes per board: 1 (a lane owns a board) or 8 (a lane group shares a board; all eight lanes hold the same
// state and take the same decisions, lane 0 of the group stores)
// worklist != nullptr: the launch plays the boards worklist[0 .. *work_count) (indices into the batch, any order)
// instead of boards 0 .. n-1 -- the later passes of the multi-pass rollout, see bounce_rollout().  Results do not
// depend on which wave or lane plays a board: RNG streams are keyed by the board's global game id.
template <bool FROM_INITIAL, int GL>
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_rollout(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                 uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                 unsigned long long* __restrict__ steps, uint32_t games_per_wave, const uint32_t* __restrict__ worklist,
                 const uint32_t* __restrict__ work_count, uint32_t* __restrict__ queue) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    // A work list is the tail of a multi-pass rollout: a handful of very long games, each a chain of thousands of
    // dependent plies, and the launch ends when the longest is through.  Their waves go first in the SIMD's issue
    // arbitration: next to 16 batches' bulk waves a ply of theirs otherwise takes 2-3x as long as on an idle chip.
    if (worklist) __builtin_amdgcn_s_setprio(3);
    const int64_t total = worklist ? (int64_t)*work_count : n;
    // The whole batch: a wave owns the boards [wave, wave + 1) x games_per_wave.  A work list: its length is only known on
    // the device, so the grid is a fixed, modest number of waves and every wave DRAWS chunks of the list from a queue
    // until it is dry (round 3 launched one wave per 8 boards of the whole batch, 32768 waves for 2^18 boards, of which
    // a hundred had work: their prologues were 1.9 x 10^7 of a step's 4 x 10^8 instructions, at 7 lanes).
    int64_t begin = worklist ? 0 : (int64_t)wave * games_per_wave;
    uint32_t avail = 0;
    bool dry = worklist == nullptr;   // nothing (more) to draw
    if (!worklist) {
        const int64_t end = begin + games_per_wave < total ? begin + games_per_wave : total;
        avail = begin < end ? (uint32_t)(end - begin) : 0u;
        if (avail == 0u) return;  // (whole wave; nothing to count)
    }
    uint32_t taken = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const bool stores = GL == 1 || (lane & (GL - 1)) == 0;           // the lane that owns the board in memory
    const uint64_t below_group = (1ull << (lane & ~(uint32_t)(GL - 1))) - 1ull;  // lanes below this lane's group

    Board b;
    b.v[0] = b.v[1] = b.v[2] = b.v[3] = 0;
    typename std::conditional<GL == 1, Moves, GroupMoves>::type mv;
    uint32_t st = 0, plies = 0, first_ply = 0, game = 0, stepped = 0;
    bool live = false, dirty = false;
    Philox4 blk;
    blk.v[0] = blk.v[1] = blk.v[2] = blk.v[3] = 0;
    bool have_block = false;

    for (;;) {
        // ---- refill
        const uint64_t need = __builtin_amdgcn_ballot_w64(!live && stores);  // one bit per idle board slot
        if (need && taken >= avail && !dry) {   // (a work list) the chunk is used up: the next one
            uint32_t next = 0;
            if (lane == 0u) next = atomicAdd(queue, games_per_wave);
            begin = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)next);
            taken = 0;
            avail = begin < total ? (uint32_t)(total - begin < (int64_t)games_per_wave ? total - begin : (int64_t)games_per_wave) : 0u;
            dry = avail == 0u;
        }
        if (need && taken < avail) {
            const uint32_t rank = (uint32_t)__popcll(need & below_group);  // the same in every lane of a group
            if (!live && taken + rank < avail) {
                game = worklist ? worklist[begin + taken + rank] : (uint32_t)(begin + taken + rank);  // board index
                const int64_t i = game;
                if (FROM_INITIAL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b.v[j] = g.init[j];
                    st = g.init_status;
                    plies = 0;
                } else {
                    b = load_board(planes, n, i);
                    st = status[i];
                    plies = plies_buf[i];
                }
                first_ply = plies;
                dirty = FROM_INITIAL;
                have_block = false;
                if (st == BGS_ST_RUNNING) {
                    if constexpr (GL == 1) enumerate(g, b, occupancy(b), plies & 1u, mv);
                    else enumerate_group(g, b, occupancy(b), plies & 1u, mv);
                    if (mv.n == 0) {  // a running board whose side to move is blocked: settle it now
                        st = settle_blocked(g, b, plies & 1u);
                        dirty = true;
                    }
                }
                live = st == BGS_ST_RUNNING && plies < max_plies;
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }

        // ---- one ply on every live lane
        if (live) {
            if (!have_block || (plies & 3u) == 0u) {
                blk = philox4x32_10(seed, first_game + (uint64_t)game, plies >> 2);
                have_block = true;
            }
            const uint32_t mover = plies & 1u;
            int s, t;
            if constexpr (GL == 1) pick_from(mv, sample_index(philox_word(blk, plies), mv.n), s, t);
            else pick_group(mv, sample_index(philox_word(blk, plies), mv.n), s, t);
            move_piece(b, s, t);
            ++plies;
            dirty = true;
            if ((1ull << t) & (g.goal_top | g.goal_bottom)) {
                st = mover + 1u;
            } else {
                const uint64_t occ = occupancy(b);
                if constexpr (GL == 1) enumerate(g, b, occ, 1u - mover, mv);
                else enumerate_group(g, b, occ, 1u - mover, mv);
                if (mv.n == 0) st = count_actions(g, b, occ, mover) ? mover + 1u : BGS_ST_DRAW;
            }
            live = st == BGS_ST_RUNNING && plies < max_plies;
        }

        // ---- boards that stopped go to memory
        if (!live && dirty) {
            if (stores) {
                const int64_t i = game;
                store_board(planes, n, i, b);
                status[i] = (uint8_t)st;
                plies_buf[i] = (uint16_t)plies;
                reward[i] = reward_pair(st);
                stepped += plies - first_ply;
            }
            dirty = false;
        }
        if (!__builtin_amdgcn_ballot_w64(live) && taken >= avail && dry) break;
    }
    add_steps(steps, stepped);
}

// ------------------------------------------------------------------------------------------------
// K3f: the fused rollout, flattened.  One lane per board as in GL = 1 -- but the move search of a ply is not run as
// nested loops (for every column: while cells are pending: for every step), whose trip counts differ from lane to
// lane so that a wave executes the SUM over columns of the per-column maxima.  It is ONE loop per wave in which every
// lane expands one cell of its own work queue per iteration -- the queue runs through the lane's sources one after
// the other and through each source's pending bounce cells -- so a wave executes the maximum over its lanes of the
// number of cells, and all lanes run the same instructions on different cells (measured before: 12.7 of 64 lanes
// active per VALU instruction in lane-group mode, 8.4 with one lane per board).
// Per-source target masks go to a per-lane dword column of LDS ([dword][lane]: the bank is the lane, dynamic indices
// never conflict); per-source counts are packed 8 bits each (a source has at most 64 targets: 0..64 needs 7 bits).
// ------------------------------------------------------------------------------------------------
struct FlatMoves {
    uint64_t counts;     // byte x = number of targets of the piece in column x of the active row
    uint32_t n;          // number of actions
    uint32_t row_base;   // cell index of column 0 of the active row
};

// the action list of `player` for the lanes with `want` set; the other lanes idle through the loop
__device__ __forceinline__ void enumerate_flat(const BounceGeom& g, const Board& b, uint64_t occ, uint32_t player, bool want,
                                               uint32_t* column, FlatMoves& m) {
    const uint64_t empty_interior = ~occ & g.interior;
    const uint64_t landing = empty_interior | (player ? g.goal_bottom : g.goal_top);
    const uint64_t bounce_on = occ & g.interior;
    const uint32_t up = player ? 0u : (uint32_t)g.w, down = player ? (uint32_t)g.w : 0u;
    uint64_t rem = want ? movable(g, occ, player) : 0ull;   // sources still to search
    const int first = rem ? __ffsll((unsigned long long)rem) - 1 : 0;
    m.row_base = (uint32_t)((int)(((uint32_t)first * g.inv_w) >> 16) * g.w);
    m.counts = 0;
    m.n = 0;
    uint64_t pending = 0, done = 0, targets = 0;
    uint32_t x = 0;
    bool open_source = false;  // a source is being searched and has not been booked yet
    while (__builtin_amdgcn_ballot_w64(rem != 0 || pending != 0 || open_source)) {
        if (pending == 0) {
            if (open_source) {  // the source's closure is complete: book it
                const uint32_t cnt = (uint32_t)__popcll(targets);
                m.counts |= (uint64_t)cnt << (8u * x);
                m.n += cnt;
                column[(2u * x) * BGS_BLOCK] = (uint32_t)targets;
                column[(2u * x + 1u) * BGS_BLOCK] = (uint32_t)(targets >> 32);
                open_source = false;
            }
            if (rem) {  // next source
                const int cell = __ffsll((unsigned long long)rem) - 1;
                rem &= rem - 1;
                x = (uint32_t)cell - m.row_base;
                pending = 1ull << cell;
                done = 0;
                targets = 0;
                open_source = true;
            }
        }
        if (pending) {  // expand one cell: a segment of value(cell) steps
            const int c = __ffsll((unsigned long long)pending) - 1;
            pending &= pending - 1;
            done |= 1ull << c;
            const uint32_t v = value_at(b, c);
            uint64_t a0 = 1ull << c, al = 0, ar = 0, land = 0;
            for (uint32_t s = 1; s <= v; ++s) {
                const uint64_t via_left = a0 | al, via_right = a0 | ar;
                const uint64_t nf = ((via_left | ar) << up) >> down;
                const uint64_t nl = (via_left & g.not_col0) >> 1;
                const uint64_t nr = (via_right & g.not_collast) << 1;
                if (s < v) {
                    a0 = nf & empty_interior;
                    al = nl & empty_interior;
                    ar = nr & empty_interior;
                    if (!(a0 | al | ar)) break;
                } else {
                    land = nf | nl | nr;
                }
            }
            targets |= land & landing;
            pending |= land & bounce_on & ~done;
        }
    }
}

// the idx-th action of the canonical list from the packed counts and the LDS column
__device__ __forceinline__ void pick_flat(const FlatMoves& m, const uint32_t* column, uint32_t idx, int& src_cell, int& dst_cell) {
    uint32_t col = 0;
    bool found = false;
#pragma unroll
    for (int x = 0; x < kMaxTrackedColumns; ++x) {
        const uint32_t cnt = (uint32_t)(m.counts >> (8 * x)) & 255u;
        const bool here = !found && idx < cnt;
        col = here ? (uint32_t)x : col;
        idx = (found || here) ? idx : idx - cnt;
        found = found || here;
    }
    const uint64_t chosen = ((uint64_t)column[(2u * col + 1u) * BGS_BLOCK] << 32) | column[(2u * col) * BGS_BLOCK];
    src_cell = (int)(m.row_base + col);
    dst_cell = (int)select_bit64(chosen, idx);
}

// Boards a draining wave has parked for the other waves of its workgroup (see the drain paragraph in the kernel).
struct ParkedBoards {
    static constexpr uint32_t WAVES = BGS_BLOCK / BGS_WAVE, CAP = 32;
    uint64_t v[4][WAVES][CAP];   // the board's value planes
    uint32_t game[WAVES][CAP];
    uint32_t plies[WAVES][CAP];
    uint32_t count[WAVES];       // entries of the segment; published once, after the entries
    uint32_t head[WAVES];        // entries claimed so far (may run past count)
    uint32_t active;             // waves still in their loop
};

template <bool FROM_INITIAL>
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_rollout_flat(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                      uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                      unsigned long long* __restrict__ steps, uint32_t chunk, const uint32_t* __restrict__ worklist,
                      const uint32_t* __restrict__ work_count, uint32_t* __restrict__ queue, uint32_t park_at) {
    extern __shared__ uint32_t target_tile[];             // [2 * kMaxTrackedColumns dwords][256 lanes]
    __shared__ ParkedBoards parked;
    uint32_t* const column = target_tile + threadIdx.x;   // this lane's dword column
    constexpr uint32_t WAVES = ParkedBoards::WAVES;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < WAVES) {
        parked.count[threadIdx.x] = 0u;
        parked.head[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) parked.active = WAVES;
    __syncthreads();
    // The drain.  Game lengths are geometric-tailed: once the queue is dry a wave's last boards keep it alive for ~120
    // more plies with ever fewer lanes busy, and with a few boards per lane per launch that is most of what a launch
    // issues.  So a wave that has nothing left to draw and at most park_at boards in flight PARKS them in LDS and
    // leaves; waves of the workgroup that are still running adopt parked boards into their idle lanes (they search the
    // board's action list again, one extra search in a game's remaining plies).  Nobody waits for anybody: `active`
    // counts the waves still in their loop, a wave that parks (or runs out of boards) leaves only if others remain,
    // and the wave that finds itself the last one takes back what it parked and sweeps up what the others left.
    // Fences are LDS-only ("local"): they never wait for the global stores of finished boards.
    bool last = false;
    auto bump = [&](uint32_t* word, uint32_t by) {  // wave-level atomic add (lane 0 issues it), old value to all lanes
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(word, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane(old);
    };
    auto leave = [&]() {  // this wave stops adopting; returns the number of waves that were still in their loop
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        const uint32_t before = bump(&parked.active, ~0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        return before;
    };
    const uint32_t total = worklist ? *work_count : (uint32_t)n;
    // Work is handed out dynamically: a wave draws chunks of `chunk` boards from a device-wide counter (zeroed before
    // the launch) whenever its lanes run out.  Game lengths are heavy-tailed, so with static chunks a wave lives as
    // long as its unluckiest lane while the others idle; with the shared queue every lane is refilled until the
    // whole batch is handed out, and a wave that holds a long game simply stops drawing.
    uint32_t begin = 0, avail = 0, taken = 0;   // the wave's current chunk: boards begin + [taken, avail)
    bool dry = false;                             // the queue has nothing left

    Board b;
    b.v[0] = b.v[1] = b.v[2] = b.v[3] = 0;
    FlatMoves mv;
    mv.counts = 0;
    mv.n = 0;
    mv.row_base = 0;
    uint32_t st = 0, plies = 0, first_ply = 0, game = 0, stepped = 0;
    bool has = false;      // this lane holds a board
    bool dirty = false;    // ... that differs from what memory holds
    bool search = false;   // ... whose side to move has no action list yet
    Philox4 blk;
    blk.v[0] = blk.v[1] = blk.v[2] = blk.v[3] = 0;
    bool have_block = false;

    for (;;) {
        // ---- refill: free lanes take the next boards of the wave's chunk; an empty chunk is replaced from the queue
        const uint64_t need = __builtin_amdgcn_ballot_w64(!has);
        if (need && taken >= avail && !dry) {
            uint32_t next = 0;
            if ((threadIdx.x & 63u) == 0u) next = atomicAdd(queue, chunk);
            begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
            taken = 0;
            avail = begin < total ? (total - begin < chunk ? total - begin : chunk) : 0u;
            dry = avail == 0u;
        }
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (!has && taken + rank < avail) {
                game = worklist ? worklist[begin + taken + rank] : begin + taken + rank;
                const int64_t i = game;
                if (FROM_INITIAL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b.v[j] = g.init[j];
                    st = g.init_status;
                    plies = 0;
                } else {
                    b = load_board(planes, n, i);
                    st = status[i];
                    plies = plies_buf[i];
                }
                first_ply = plies;
                has = true;
                dirty = FROM_INITIAL;
                have_block = false;
                search = st == BGS_ST_RUNNING;
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }
        const bool draining = dry && taken >= avail;  // (wave-uniform) nothing left to draw
        bool adopted_now = false;   // (wave-uniform) this iteration's adoption attempt found something
        if (draining && need) {
            // ---- idle lanes adopt parked boards.  count[] and head[] are adjacent: lanes 0..7 fetch them in one access
            const uint32_t wanted = (uint32_t)__popcll(need);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            uint32_t word = 0;
            if (lane < 2u * WAVES) word = __hip_atomic_load(&parked.count[0] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            uint32_t assigned = 0;
#pragma unroll
            for (uint32_t sgm = 0; sgm < WAVES; ++sgm) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(word, sgm);
                const uint32_t h = (uint32_t)__builtin_amdgcn_readlane(word, WAVES + sgm);
                if (assigned < wanted && h < c) {
                    const uint32_t old = bump(&parked.head[sgm], wanted - assigned);
                    const uint32_t got = old < c ? (c - old < wanted - assigned ? c - old : wanted - assigned) : 0u;
                    if (!has && rank >= assigned && rank < assigned + got) {
                        const uint32_t e = old + (rank - assigned);
#pragma unroll
                        for (int j = 0; j < 4; ++j) b.v[j] = parked.v[j][sgm][e];
                        game = parked.game[sgm][e];
                        plies = parked.plies[sgm][e];
                        first_ply = plies;  // (the wave that parked it has counted the plies up to here)
                        st = BGS_ST_RUNNING;
                        has = true;
                        dirty = true;
                        have_block = false;
                        search = true;
                    }
                    assigned += got;
                }
            }
            adopted_now = assigned != 0u;
        }

        // ---- the action lists of the boards that need one (new boards, boards that have just moved); a board whose
        // side to move has no action is settled here, also at the ply cap (the transition that blocked it counts)
        if (__builtin_amdgcn_ballot_w64(search)) {
            const uint64_t occ = occupancy(b);
            enumerate_flat(g, b, occ, plies & 1u, search, column, mv);
            const bool blocked = search && mv.n == 0u;
            if (__builtin_amdgcn_ballot_w64(blocked)) {
                // the other side wins if IT could move, else a draw (Appendix B rule 7; a loaded or start position
                // without moves is settled the same way)
                FlatMoves other;
                enumerate_flat(g, b, occ, 1u - (plies & 1u), blocked, column, other);
                if (blocked) {
                    st = other.n ? (1u - (plies & 1u)) + 1u : BGS_ST_DRAW;
                    dirty = true;
                }
            }
            search = false;
        }
        const bool run = has && st == BGS_ST_RUNNING && plies < max_plies;

        // ---- boards that stopped go to memory and free their lane
        if (has && !run) {
            if (dirty) {
                const int64_t i = game;
                store_board(planes, n, i, b);
                status[i] = (uint8_t)st;
                plies_buf[i] = (uint16_t)plies;
                reward[i] = reward_pair(st);
                stepped += plies - first_ply;
            }
            has = false;
            dirty = false;
        }
        if (!__builtin_amdgcn_ballot_w64(has)) {
            if (!draining) continue;    // (a chunk of boards that were not running: draw the next one)
            // The last wave leaves only after an adoption attempt with ALL its lanes idle has found nothing: if its
            // lanes were busy at the top of this iteration and their boards all stopped in it (a ply cap does that), the
            // boards other waves parked meanwhile would otherwise never be played.
            // (... NOTHING: boards adopted in this very iteration may all have stopped in it -- a ply cap does that -- with
            // more still parked; round 4, found on the Connect kernel that shares the protocol)
            if (last && (need != ~0ull || adopted_now)) continue;
            if (last) break;
            if (leave() > 1u) break;    // others are still running: whatever gets parked later is theirs
            last = true;                // everybody else has left: sweep up what they parked
            continue;
        }
        if (draining && !last) {
            const uint64_t still = __builtin_amdgcn_ballot_w64(has);  // (every board still held is running here)
            const uint32_t left = (uint32_t)__popcll(still);
            if (left <= park_at) {
                const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(still >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)still, 0u));
                if (has) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) parked.v[j][w][e] = b.v[j];
                    parked.game[w][e] = game;
                    parked.plies[w][e] = plies;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_store(&parked.count[w], left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (leave() > 1u) {  // parked; somebody is still there to adopt them
                    if (has) stepped += plies - first_ply;
                    break;
                }
                // nobody is: take back what has not been adopted (an adopter that has left has finished its boards)
                const uint32_t adopted = bump(&parked.head[w], left);
                if (has && e < adopted) {
                    stepped += plies - first_ply;
                    has = false;
                    dirty = false;
                }
                last = true;
            }
        }

        // ---- one ply on every running board
        if (run) {
            if (!have_block || (plies & 3u) == 0u) {
                blk = philox4x32_10(seed, first_game + (uint64_t)game, plies >> 2);
                have_block = true;
            }
            const uint32_t mover = plies & 1u;
            int s, t;
            pick_flat(mv, column, sample_index(philox_word(blk, plies), mv.n), s, t);
            move_piece(b, s, t);
            ++plies;
            dirty = true;
            if ((1ull << t) & (g.goal_top | g.goal_bottom)) st = mover + 1u;  // (stored and freed next iteration)
            else search = true;
        }
    }
    add_steps(steps, stepped);
}

// ------------------------------------------------------------------------------------------------
// K3p: the fused rollout on the PIECE LIST (from the start position).  Bounce never captures and never changes a
// piece's value, so a board is the cells of its P pieces, and every board of a batch has the same pieces: piece k's
// value is wave-uniform.  The landing cells of a segment depend only on the cell it starts from and on the board -- not
// on which source's walk got there -- so a ply's move search splits into
//   A  every lane runs the segment of EVERY piece of its board, piece by piece: the trip count of the step loop is the
//      piece's value, the same in all 64 lanes -- no divergence at all (K3f: a lane expands the cells of its own queue,
//      ~8.75 a ply with ~5.9 distinct, a wave pays the maximum over its lanes, and a segment of 1-3 steps runs as 3);
//      the P landing masks go to the lane's LDS column;
//   B  the closure per source -- K3f's loop "take a pending cell, add its landing cells" -- with the segment replaced by
//      a look-up: which piece stands on the cell (index planes), then its mask from LDS (~30 instructions a cell
//      instead of ~110).
// The per-source target masks are not kept: a source's count is enough to sample, and the sampled source's closure is
// run once more to find the target.  Everything around the search (work queue, parked boards, refill) is K3f's.
// ------------------------------------------------------------------------------------------------
template <int PMAX>
struct PieceBoard {
    uint32_t pos[PMAX / 4];   // piece k stands on cell (pos[k >> 2] >> 8 (k & 3)) & 255
    uint64_t idx[4];          // plane p: bit c = bit p of the index of the piece on cell c
    uint64_t occ;
};

template <int PMAX, class GEO>
__device__ __forceinline__ void pieces_from_start(const GEO& g, PieceBoard<PMAX>& b) {
#pragma unroll
    for (int j = 0; j < PMAX / 4; ++j)
        b.pos[j] = (uint32_t)g.piece_cell[4 * j] | ((uint32_t)g.piece_cell[4 * j + 1] << 8) |
                   ((uint32_t)g.piece_cell[4 * j + 2] << 16) | ((uint32_t)g.piece_cell[4 * j + 3] << 24);
#pragma unroll
    for (int p = 0; p < 4; ++p) b.idx[p] = g.piece_idx[p];
    b.occ = g.init[0] | g.init[1] | g.init[2] | g.init[3];
}

// index planes and occupancy from the positions (adopted boards)
template <int PMAX, class GEO>
__device__ __forceinline__ void pieces_rebuild(const GEO& g, PieceBoard<PMAX>& b) {
    b.idx[0] = b.idx[1] = b.idx[2] = b.idx[3] = 0;
    b.occ = 0;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < (int)g.piece_count) {
            const uint64_t bit = 1ull << ((b.pos[k >> 2] >> (8 * (k & 3))) & 63u);
            b.occ |= bit;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if ((k >> p) & 1) b.idx[p] |= bit;
        }
}

// value planes (the batch's memory format) from the positions
template <int PMAX, class GEO>
__device__ __forceinline__ Board pieces_to_planes(const GEO& g, const PieceBoard<PMAX>& b) {
    Board out;
    out.v[0] = out.v[1] = out.v[2] = out.v[3] = 0;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < (int)g.piece_count) {
            const uint64_t bit = 1ull << ((b.pos[k >> 2] >> (8 * (k & 3))) & 63u);
            const uint32_t v = g.piece_value[k];  // (wave-uniform)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if ((v >> p) & 1u) out.v[p] |= bit;
        }
    return out;
}

// What K3p writes when a board stops: the POSITIONS of its pieces (the packed bytes, as two 64-bit words into the board's
// slots of planes 0 and 1) -- ten instructions.  The value planes, the batch's memory format, are made from them for all
// boards at once by k_bounce_positions_to_planes behind the rollout kernel: full lanes, 2^18 boards in a few microseconds.
// (Round 3 built the planes where the board stopped: ~150 instructions with two or three lanes active, in nine
// iterations of ten -- 4 % of the kernel's instructions.)
template <int PMAX>
__device__ __forceinline__ void store_positions(uint64_t* __restrict__ planes, int64_t n, int64_t i, const PieceBoard<PMAX>& b) {
    planes[i] = (uint64_t)b.pos[0] | ((uint64_t)b.pos[1] << 32);
    if (PMAX > 8) planes[n + i] = (uint64_t)b.pos[2] | (PMAX > 12 ? (uint64_t)b.pos[PMAX > 12 ? 3 : 0] << 32 : 0ull);
}

__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_positions_to_planes(BounceGeom g, uint64_t* __restrict__ planes, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t lo = planes[i], hi = g.piece_count > 8 ? planes[n + i] : 0ull;
    Board out;
    out.v[0] = out.v[1] = out.v[2] = out.v[3] = 0;
#pragma unroll
    for (int k = 0; k < BGS_BOUNCE_MAX_PIECES; ++k)
        if (k < (int)g.piece_count) {
            const uint32_t cell = (uint32_t)((k < 8 ? lo : hi) >> (8 * (k & 7))) & 63u;
            const uint64_t bit = 1ull << cell;
            const uint32_t v = g.piece_value[k];  // (uniform)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if ((v >> p) & 1u) out.v[p] |= bit;
        }
    store_board(planes, n, i, out);
}

template <int PMAX>
__device__ __forceinline__ uint32_t piece_on(const PieceBoard<PMAX>& b, int c) {
    return (uint32_t)((b.idx[0] >> c) & 1ull) | ((uint32_t)((b.idx[1] >> c) & 1ull) << 1) |
           ((uint32_t)((b.idx[2] >> c) & 1ull) << 2) | ((uint32_t)((b.idx[3] >> c) & 1ull) << 3);
}

template <int PMAX>
__device__ __forceinline__ void move_piece_on(PieceBoard<PMAX>& b, int src_cell, int dst_cell) {
    const uint32_t k = piece_on(b, src_cell);
    const uint64_t keep = ~(1ull << src_cell), put = 1ull << dst_cell;
#pragma unroll
    for (int p = 0; p < 4; ++p) b.idx[p] = (b.idx[p] & keep) | (((k >> p) & 1u) ? put : 0ull);
    b.occ = (b.occ & keep) | put;
    const uint32_t sh = 8u * (k & 3u);
#pragma unroll
    for (int j = 0; j < PMAX / 4; ++j)
        b.pos[j] = (k >> 2) == (uint32_t)j ? (b.pos[j] & ~(255u << sh)) | ((uint32_t)dst_cell << sh) : b.pos[j];
}

// what phase A leaves in REGISTERS (the piece loops are unrolled, so every index below is static): the landing cells of
// every piece's segment and, per piece, the set of pieces standing on them -- a 16 x 16 bit matrix, two rows to a dword
template <int PMAX>
struct Lands {
    uint64_t v[PMAX];
    uint32_t adj[PMAX / 2];   // row k (16 bits): bit j = piece j stands on a landing cell of piece k
};

// phase A: the landing cells of every piece's segment for `player`, and who stands on them
template <int PMAX, class GEO>
__device__ __forceinline__ void land_all(const GEO& g, const PieceBoard<PMAX>& b, uint32_t player, Lands<PMAX>& L) {
    const uint64_t empty_interior = ~b.occ & g.interior;
    const uint32_t up = player ? 0u : (uint32_t)g.w, down = player ? (uint32_t)g.w : 0u;
    // every piece's cell, unpacked ONCE a ply: the "who stands there" test below reads each of them PMAX times (the
    // compiler re-extracted the byte at every use: 9 shifts per piece, ~100 instructions a ply)
    uint32_t cell[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        cell[k] = (b.pos[k >> 2] >> (8 * (k & 3))) & 63u;
        asm("" : "+v"(cell[k]));
    }
#pragma unroll
    for (int j = 0; j < PMAX / 2; ++j) L.adj[j] = 0;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        L.v[k] = 0;
        if (k < (int)g.piece_count) {
            const uint32_t v = g.piece_value[k];  // wave-uniform: the step loop below does not diverge
            uint64_t a0 = 1ull << cell[k], al = 0, ar = 0, land = 0;
            for (uint32_t s = 1; s <= v; ++s) {
                const uint64_t via_left = a0 | al, via_right = a0 | ar;  // who may go on left / right (no reversal)
                const uint64_t nf = ((via_left | ar) << up) >> down;
                const uint64_t nl = (via_left & g.not_col0) >> 1;
                const uint64_t nr = (via_right & g.not_collast) << 1;
                if (s < v) {
                    a0 = nf & empty_interior;
                    al = nl & empty_interior;
                    ar = nr & empty_interior;
                } else {
                    land = nf | nl | nr;
                }
            }
            L.v[k] = land;
            // the pieces it lands ON (a walk that lands on a piece goes on with that piece's segment).  A segment never
            // comes back to the cell it started from (no step backwards, no reversal), so the piece itself is left out
            uint32_t hits = 0;
#pragma unroll
            for (int j = 0; j < PMAX; ++j)
                if (j != k && j < (int)g.piece_count) hits |= ((uint32_t)(land >> cell[j]) & 1u) << j;
            L.adj[k >> 1] |= hits << (16 * (k & 1));
        }
    }
}

// phase B, part 1: the transitive closure of "lands on" -- Warshall on the packed rows: for every pivot k, every row
// that has bit k takes row k in.  A row lives in a 16-bit field, so (fields with bit k) x row_k is one multiply per dword.
template <int PMAX, class GEO>
__device__ __forceinline__ void close_over_bounces(const GEO& g, Lands<PMAX>& L) {
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < (int)g.piece_count) {
            const uint32_t row = (L.adj[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
#pragma unroll
            for (int d = 0; d < PMAX / 2; ++d) L.adj[d] |= ((L.adj[d] >> k) & 0x00010001u) * row;
        }
}

// what a lane knows about its board's action list: per column of the active row the number of targets (8 bits each) and
// the PIECES whose landing cells make them up (the source's closure as a 16-bit set of piece indices, two columns a word)
struct PieceMoves {
    uint64_t counts;
    uint32_t reach[kMaxTrackedColumns / 2];
    uint32_t n;
    uint32_t row_base;
};

// the union of the landing masks of a set of pieces.  Written with masks, not selects: bit k of `members` sign-extended
// (one v_bfe_i32) ANDs the piece's mask in, (mask & sel) | acc is one v_and_or_b32 per half -- three instructions a piece.
// (The select form -- `members & (1 << k) ? L.v[k] : 0` -- compiled to and + compare + two v_cndmask + or, and the
// compare's SGPR result costs the v_cndmask behind it an s_nop: 650 of the ~930 instructions of a ply's counting.)
template <int PMAX, class GEO>
__device__ __forceinline__ uint64_t landed_by(const GEO& g, const Lands<PMAX>& L, uint32_t members) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < (int)g.piece_count) {
            uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)members, k, 1);  // 0 or ~0
            asm("" : "+v"(sel));   // (keeps the mask a mask: the compiler would turn it back into a compare + selects)
            lo |= (uint32_t)L.v[k] & sel;
            hi |= (uint32_t)(L.v[k] >> 32) & sel;
        }
    return ((uint64_t)hi << 32) | lo;
}

// phase B, part 2: per column of the active row, the source's closure and the number of its targets -- for the lanes
// with `want` set (phases A and B1 must have run for this board and player).  No loop whose trip count depends on the
// board: every lane does the same work for every column.
template <int PMAX, class GEO>
__device__ __forceinline__ void count_from_lands(const GEO& g, const PieceBoard<PMAX>& b, uint32_t player, bool want,
                                                 const Lands<PMAX>& L, PieceMoves& m) {
    const uint64_t occ = b.occ;
    const uint64_t landing = (~occ & g.interior) | (player ? g.goal_bottom : g.goal_top);
    const uint64_t rem = want ? movable(g, occ, player) : 0ull;   // the sources
    const int first = rem ? __ffsll((unsigned long long)rem) - 1 : 0;
    m.row_base = (uint32_t)((int)(((uint32_t)first * g.inv_w) >> 16) * g.w);
    m.counts = 0;
    m.n = 0;
#pragma unroll
    for (int j = 0; j < kMaxTrackedColumns / 2; ++j) m.reach[j] = 0;
#pragma unroll
    for (int x = 0; x < kMaxTrackedColumns; ++x)
        if (x < g.w) {
            const int cell = (int)((m.row_base + (uint32_t)x) & 63u);
            const bool is_source = (rem >> cell) & 1ull;
            const uint32_t i = piece_on(b, cell);
            uint32_t word = 0;
#pragma unroll
            for (int d = 0; d < PMAX / 2; ++d) word = (i >> 1) == (uint32_t)d ? L.adj[d] : word;
            const uint32_t members = is_source ? (((word >> (16u * (i & 1u))) & 0xFFFFu) | (1u << i)) : 0u;
            const uint32_t cnt = (uint32_t)__popcll(landed_by(g, L, members) & landing);
            m.counts |= (uint64_t)cnt << (8 * x);
            m.n += cnt;
            m.reach[x >> 1] |= members << (16 * (x & 1));
        }
}

// the idx-th action of the canonical list: the column from the packed counts, its targets from the closure's landing masks
template <int PMAX, class GEO>
__device__ __forceinline__ void pick_from_lands(const GEO& g, const PieceBoard<PMAX>& b, uint32_t player,
                                                const PieceMoves& m, const Lands<PMAX>& L, uint32_t idx, int& src_cell,
                                                int& dst_cell) {
    uint32_t col = 0;
    bool found = false;
#pragma unroll
    for (int x = 0; x < kMaxTrackedColumns; ++x) {
        const uint32_t cnt = (uint32_t)(m.counts >> (8 * x)) & 255u;
        const bool here = !found && idx < cnt;
        col = here ? (uint32_t)x : col;
        idx = (found || here) ? idx : idx - cnt;
        found = found || here;
    }
    src_cell = (int)((m.row_base + col) & 63u);
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < kMaxTrackedColumns / 2; ++j) word = (col >> 1) == (uint32_t)j ? m.reach[j] : word;
    const uint32_t members = (word >> (16u * (col & 1u))) & 0xFFFFu;
    const uint64_t landing = (~b.occ & g.interior) | (player ? g.goal_bottom : g.goal_top);
    dst_cell = (int)select_bit64(landed_by(g, L, members) & landing, idx);
}

// ------------------------------------------------------------------------------------------------
// The OPENING BOOK of a start position (round 5).  Every board of a from-initial rollout starts from the same position:
// ply 0 is ONE position on all 2^18 lanes, plies 1 and 2 a few hundred, and a wave iteration of K3p costs the same
// ~2 400 instructions whatever its lanes search.  So the first plies are not searched at all: the positions reachable in
// `depth` plies (default board: 22 actions at the start, 496 two-ply paths, 12 684 three-ply paths, 338 590 four-ply
// paths) are enumerated ONCE per start position -- by these kernels, with K3p's own search, pick and move, so the
// canonical action order is the rollout's by construction -- and a lane that takes a new game walks them with the game's
// own draws (philox block 0 = the words of plies 0 .. 3):
//     at = idx(v0, n0);  e = L1[at]: at = off(e) + idx(v1, n(e));  e = L2[at]: ...;  entry = T[depth][at]
// and starts at ply `depth` on the entry's board (positions, index planes, occupancy: one 64-byte line).  The walk is
// keyed by (game, ply) draws exactly as the plies it replaces, so the boards are those of a search of every ply, bit for
// bit.  A path that ends early (a piece reaches the goal row, a side is blocked) is carried down the levels as an entry
// with one pseudo-action leading to itself, its final status and ITS ply count.
//   level d: hdr[d] paths of d plies, link word of path i = first child path << 10 | number of actions (children)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kBookCap1 = 64, kBookCap2 = 1024, kBookCap3 = 16384, kBookCap4 = 1u << 19;
constexpr uint32_t kBookLinkBits = 10, kBookLinkMax = (1u << kBookLinkBits) - 1u;
constexpr uint32_t kBookLinks = kBookCap1 + kBookCap2 + kBookCap3;   // L1 at 0, L2 at kBookCap1, L3 at kBookCap1 + kBookCap2
constexpr uint32_t kBookLdsLinks = kBookCap1 + kBookCap2;            // what the rollout keeps in LDS
struct BookEntry {
    uint32_t pos[4];     // PieceBoard::pos (16 pieces)
    uint64_t idx[4];     // PieceBoard::idx
    uint64_t occ;
    uint32_t meta;       // status | plies << 8 | actions << 16 (the side to move's; 1 for a path that has ended)
    uint32_t pad;
};
static_assert(sizeof(BookEntry) == 64, "one cache line, four 16-byte loads");

// hdr[0]: flags (bit d: level d holds a position with more actions than a link word can say), hdr[d]: paths of d plies
__global__ void __launch_bounds__(BGS_WAVE) k_bounce_book_root(BounceGeom g, uint32_t* __restrict__ hdr) {
    constexpr int PMAX = BGS_BOUNCE_MAX_PIECES;
    PieceBoard<PMAX> b;
    pieces_from_start(g, b);
    Lands<PMAX> L;
    PieceMoves mv;
    land_all<PMAX>(g, b, 0u, L);
    close_over_bounces<PMAX>(g, L);
    count_from_lands<PMAX>(g, b, 0u, true, L, mv);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        hdr[0] = mv.n > kBookLinkMax ? 1u : 0u;
        hdr[1] = g.init_status == BGS_ST_RUNNING ? mv.n : 0u;
    }
}

__global__ void __launch_bounds__(BGS_WAVE)
k_bounce_book_expand(BounceGeom g, uint32_t level, const BookEntry* __restrict__ parents, const uint32_t* __restrict__ parent_links,
                     BookEntry* __restrict__ out, uint32_t out_cap, uint32_t* __restrict__ hdr) {
    constexpr int PMAX = BGS_BOUNCE_MAX_PIECES;
    const uint32_t j = blockIdx.x * BGS_WAVE + threadIdx.x;
    const uint32_t total = hdr[level] < out_cap ? hdr[level] : out_cap;
    const bool active = j < total;
    PieceBoard<PMAX> b;
    pieces_from_start(g, b);   // (idle lanes search the start position: every lane runs every phase)
    uint32_t st = BGS_ST_RUNNING, plies = level - 1u, action = active ? j : 0u;
    if (active && level > 1u) {
        uint32_t lo = 0, hi = hdr[level - 1u];   // the parent: the last path whose first child is <= j
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((parent_links[mid] >> kBookLinkBits) <= j) lo = mid; else hi = mid;
        }
        action = j - (parent_links[lo] >> kBookLinkBits);
        const BookEntry& e = parents[lo];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b.pos[k] = e.pos[k];
            b.idx[k] = e.idx[k];
        }
        b.occ = e.occ;
        st = e.meta & 255u;
        plies = (e.meta >> 8) & 255u;
    }
    const uint32_t side = (level - 1u) & 1u, other = level & 1u;
    Lands<PMAX> L;
    PieceMoves mv;
    land_all<PMAX>(g, b, side, L);
    close_over_bounces<PMAX>(g, L);
    count_from_lands<PMAX>(g, b, side, true, L, mv);
    uint32_t n = 1u;
    const bool moves = active && st == BGS_ST_RUNNING;   // (a running parent has mv.n > action actions: that is what its link says)
    int s = 0, t = 0;
    pick_from_lands<PMAX>(g, b, side, mv, L, moves ? action : 0u, s, t);
    if (moves) {
        move_piece_on(b, s, t);
        plies = level;
        if ((1ull << t) & (g.goal_top | g.goal_bottom)) st = side + 1u;
    }
    // the new position's side to move; a side without an action loses to a side that has one (SURVEY App. B rule 7)
    land_all<PMAX>(g, b, other, L);
    close_over_bounces<PMAX>(g, L);
    count_from_lands<PMAX>(g, b, other, true, L, mv);
    const uint32_t n_other = mv.n;
    land_all<PMAX>(g, b, side, L);
    close_over_bounces<PMAX>(g, L);
    count_from_lands<PMAX>(g, b, side, true, L, mv);
    if (moves && st == BGS_ST_RUNNING) {
        if (n_other == 0u) st = mv.n ? side + 1u : BGS_ST_DRAW;
        else n = n_other;
    }
    if (active) {
        if (n > kBookLinkMax) atomicOr(hdr, 1u << level);
        BookEntry e;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e.pos[k] = b.pos[k];
            e.idx[k] = b.idx[k];
        }
        e.occ = b.occ;
        e.meta = st | (plies << 8) | ((n <= kBookLinkMax ? n : kBookLinkMax) << 16);
        e.pad = 0;
        out[j] = e;
    }
}

// the links of a level (first child path << 10 | actions) and the number of paths of the next one: one workgroup
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_book_links(const BookEntry* __restrict__ table, uint32_t level, uint32_t cap, uint32_t* __restrict__ links,
                    uint32_t* __restrict__ hdr) {
    __shared__ uint32_t sums[BGS_BLOCK];
    const uint32_t total = hdr[level] < cap ? hdr[level] : cap;
    const uint32_t per = (total + BGS_BLOCK - 1u) / BGS_BLOCK;
    const uint32_t first = threadIdx.x * per, last = first + per < total ? first + per : total;
    uint32_t sum = 0;
    for (uint32_t i = first; i < last; ++i) sum += (table[i].meta >> 16) & 0xFFFFu;
    sums[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < BGS_BLOCK; ++k) {
            const uint32_t v = sums[k];
            sums[k] = run;
            run += v;
        }
        hdr[level + 1u] = run;
        if (hdr[level] > cap) atomicOr(hdr, 1u << level);
    }
    __syncthreads();
    uint32_t off = sums[threadIdx.x];
    for (uint32_t i = first; i < last; ++i) {
        const uint32_t n = (table[i].meta >> 16) & 0xFFFFu;
        links[i] = (off << kBookLinkBits) | n;
        off += n;
    }
}

// ------------------------------------------------------------------------------------------------
// K3w: ONE BOARD PER WAVE, a piece per lane -- the last pass of the rollout, for the handful of games per batch that
// run for thousands of plies (one or two per 2^18 never end and stop at max_plies).  Such a game is a chain of dependent
// plies; the launch is over when it is, so what counts is the LATENCY of its ply, i.e. how many instructions the wave
// that holds it has to issue per ply.  The 8-lanes-per-board kernel walks a column's closure cell by cell (a lane runs
// `reach`: ~1000 instructions a ply on the longest column).  Here the lanes ARE the pieces (lane k = the k-th occupied
// cell, re-derived from the planes every ply, so the lanes are always in cell order):
//   A   every lane runs its own piece's segment (the same code as K3p's phase A; the step loop runs to the largest value
//       on the board, lanes drop out at their own);
//   hit "which pieces does my segment land on": the cells of the other pieces come over v_readlane, one bit test each;
//   B1  the closure of "lands on" -- Warshall with one ROW PER LANE: pivot p's row comes over v_readlane, every lane
//       that has bit p takes it in (three instructions a pivot instead of K3p's multiply over six packed dwords);
//   B2  a source's targets = the OR of its members' landing masks (masks over v_readlane, selected by sign-extended
//       member bits), counts; the sources of the active row are neighbouring lanes in column order, so "actions before
//       mine" is a prefix sum over 16 lanes (four DPP row shifts), the total is lane 15's;
//   pick the lane whose range holds the index is the source (a ballot); its k-th target is found by the CELLS (lane c counts
//       the targets below cell c with v_mbcnt; the target cell whose count is k is it): a dozen instructions.
// ~320 VALU and ~50 v_readlane a ply against ~1000.  The board itself (four value planes) is wave-uniform, and so is all
// the state of the wave's one game -- kept in scalar registers on purpose (see the ply loop: `same`, `word_at`).
// Measured per ply of a wave that has its SIMD to itself (cycle counter, round 5): a full search 3 200 cycles, with the
// memo's look-up, fill and link bookkeeping around it 4 400, a remembered position 1 300, a hop along a link 250.
// Results cannot differ from the other kernels': the same rules, the same canonical action order (sources by column,
// targets by cell index), the RNG keyed by game id and ply.
// ------------------------------------------------------------------------------------------------
#ifndef BGS_MEMO_BITS
#define BGS_MEMO_BITS 5
#endif
#ifndef BGS_WAVE_LINKS
#define BGS_WAVE_LINKS 32
#endif
constexpr uint32_t kWaveMemoBits = BGS_MEMO_BITS, kWaveMemoSlots = 1u << kWaveMemoBits;
constexpr uint32_t kWaveLinks = BGS_WAVE_LINKS;   // actions per remembered position whose successor is remembered too (K3w, see the ply loop)
// a link word is epoch << 16 | actions of the successor << 8 | the successor's slot, and lanes 0 .. kWaveLinks - 1 clear a row
static_assert(kWaveMemoSlots <= 256 && kWaveLinks <= 64, "a link holds the slot in 8 bits; one wave clears a slot's row of links");
constexpr uint32_t kWaveEpochLimit = 0xFFFFu;   // 16 bits of epoch in a link: at the limit the memo starts over empty

struct WaveMoves {
    uint64_t targets;    // this lane's piece, if it is a source: its legal landing cells
    uint32_t count;      // popcount(targets)
    uint32_t before;     // actions of the sources left of this lane's
    uint32_t cell;       // this lane's piece stands here
    uint32_t n;          // actions of the board (uniform)
};

__device__ __forceinline__ uint32_t row_shift_right(uint32_t x, int by) {   // lane i of a 16-lane row gets lane i - by's x, 0 below
    switch (by) {
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
        case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    }
}

template <int PMAX, class GEO>
__device__ __forceinline__ void enumerate_wave(const GEO& g, const Board& b, uint32_t player, WaveMoves& m) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t occ = occupancy(b);
    const uint32_t pieces = (uint32_t)__popcll(occ);          // (uniform; <= PMAX: the host only sends such boards)
    const bool alive = lane < pieces;
    // the cell of the lane-th piece.  The board is the same on every lane, so lane c knows whether cell c is occupied and how
    // many pieces stand below it (two v_mbcnt on the occupancy) -- the number of the lane that wants to know -- and tells it
    // with ONE forward permute; the popcount-guided search for the lane-th set bit is forty instructions (round 5)
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(occ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)occ, 0u));
    const bool occupied = ((occ >> lane) & 1ull) != 0ull;
    static_assert(PMAX < 63, "lane 63 takes what the empty cells send");
    const uint32_t told = (uint32_t)__builtin_amdgcn_ds_permute((int)((occupied ? below : 63u) << 2), (int)lane);
    const uint32_t cell = alive ? told : 0u;
    const uint32_t value = alive ? value_at(b, (int)cell) : 0u;
    m.cell = cell;
    const uint64_t empty_interior = ~occ & g.interior;
    const uint64_t landing = empty_interior | (player ? g.goal_bottom : g.goal_top);
    const uint32_t up = player ? 0u : (uint32_t)g.w, down = player ? (uint32_t)g.w : 0u;
    // A: the landing cells of this lane's piece
    uint64_t a0 = alive ? 1ull << cell : 0ull, al = 0, ar = 0, land = 0;
    for (uint32_t s = 1; __builtin_amdgcn_ballot_w64(s <= value) != 0ull; ++s) {
        if (s <= value) {
            const uint64_t via_left = a0 | al, via_right = a0 | ar;
            const uint64_t nf = ((via_left | ar) << up) >> down;
            const uint64_t nl = (via_left & g.not_col0) >> 1;
            const uint64_t nr = (via_right & g.not_collast) << 1;
            if (s < value) {
                a0 = nf & empty_interior;
                al = nl & empty_interior;
                ar = nr & empty_interior;
            } else {
                land = nf | nl | nr;
            }
        }
    }
    // who stands on them (bit j = the piece of lane j); a segment never returns to its own start cell
    uint32_t row = 0;
    // (no "is there a piece j" tests: a lane without a piece has no landing cells and an empty row, so whatever bit the
    // others compute FOR it selects nothing -- and a branch costs a lone wave more than the three instructions it skips)
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cell, j);
        row |= ((uint32_t)(land >> cj) & 1u) << j;
    }
    // B1: the closure over bounces, a row per lane
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        const uint32_t rp = (uint32_t)__builtin_amdgcn_readlane((int)row, p);
        row |= (uint32_t)__builtin_amdgcn_sbfe((int)row, p, 1) & rp;
    }
    // B2: a source's targets are the landing cells of its closure
    const uint64_t sources = movable(g, occ, player);
    const bool is_source = alive && ((sources >> cell) & 1ull);
    const uint32_t members = row | (1u << lane);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)land, j);
        const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(land >> 32), j);
        const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)members, j, 1);
        lo |= lj & sel;
        hi |= hj & sel;
    }
    const uint64_t targets = is_source ? ((((uint64_t)hi << 32) | lo) & landing) : 0ull;
    const uint32_t count = (uint32_t)__popcll(targets);
    uint32_t incl = count;   // the lanes are in cell order and the sources share a row: a prefix sum is "in column order"
    incl += row_shift_right(incl, 1);
    incl += row_shift_right(incl, 2);
    incl += row_shift_right(incl, 4);
    incl += row_shift_right(incl, 8);
    m.targets = targets;
    m.count = count;
    m.before = incl - count;
    m.n = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
}

// the memo of one K3w wave (see wave_play_game): kWaveMemoSlots positions in sets of 2 ways, ~10 KB with the links (PMAX = 12)
template <int PMAX>
struct WaveMemo {
    uint64_t key[kWaveMemoSlots][4];
    uint64_t targets[kWaveMemoSlots][PMAX];
    uint32_t lane[kWaveMemoSlots][PMAX];
    uint32_t n[kWaveMemoSlots];
    uint32_t tag[kWaveMemoSlots];
    uint32_t last[kWaveMemoSlots / 2];   // per set: the way used last
    // ... and where the remembered positions LEAD: link[slot][action] = epoch << 16 | actions of the successor << 8 | its slot.
    // A link holds as long as no remembered position has been replaced since it was written (`epoch` counts those; a slot's
    // row is cleared when the slot is filled).  A game that never ends hops along them -- one LDS look-up, a sample, no
    // board -- for as long as the sampled action has a link: 0.13 us a ply where the look-up of the position costs 0.55.
    // (a row has one more word than links, always 0: "no link" for an action beyond the row, read without a test)
    uint32_t link[kWaveMemoSlots][kWaveLinks + 1u];
};

// orders the LDS accesses of ONE wave's lanes (the memo is wave-private: no other wave ever touches it).  The stand-alone
// K3w kernel has one-wave workgroups; the tail role of the bulk kernel (round 6) runs this code on waves of a four-wave
// workgroup, where a workgroup barrier would wait for waves that are somewhere else entirely.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}

template <int PMAX>
__device__ __forceinline__ void wave_memo_reset(WaveMemo<PMAX>& memo, uint32_t lane) {
    for (uint32_t e = lane; e < kWaveMemoSlots; e += BGS_WAVE) {
        memo.tag[e] = 0xFFFFFFFFu;   // (no side is 0xFFFFFFFF)
        memo.last[e >> 1] = 1u;
        memo.link[e][kWaveLinks] = 0u;
    }
    wave_lds_sync();
}

// ONE game on ONE wave, from where it stands (b, st = running, plies) to its end or to max_plies: K3w's ply loop.  `b`, `st`,
// `plies` are wave-uniform on entry and on return.
template <int PMAX, class GEO>
__device__ __forceinline__ void wave_play_game(const GEO& g, WaveMemo<PMAX>& memo, uint32_t& epoch, Board& b, uint32_t& st,
                                               uint32_t& plies, uint64_t seed, uint64_t game_id, uint32_t max_plies,
                                               uint32_t epoch_limit, uint32_t cold_limit, uint32_t bypass_plies) {
    const uint32_t lane = threadIdx.x & 63u;
    WaveMoves mv;
    // The words of the plies to come.  The board is one, the lanes are 64: lane l keeps the word of ply `ahead_first` + l
    // (its own philox call, for the block that ply lies in), so one call's latency buys the words of 64 plies, and a
    // ply's word is ONE v_readlane away -- a scalar.  That matters beyond the philox calls it saves (one every four
    // plies on all lanes alike, a third of a hop's dependent chain): bgs_common.h's philox_word selects its word
    // through vector registers on purpose, which made the word -- and with it the sampled index, the link, the loop
    // exits and the ply count of this whole loop -- divergent in the compiler's eyes: every hop ran as masked vector
    // code.  With the word a scalar the hop is scalar arithmetic around one LDS look-up (round 5).
    uint32_t ahead = 0u;
    uint32_t ahead_first = 0x80000000u;   // (uniform; nothing computed yet: every ply -- at most 65535 -- is "64 or more past it")
    auto word_at = [&](uint32_t ply) -> uint32_t {
        if (ply - ahead_first >= (uint32_t)BGS_WAVE) {   // (unsigned: also a ply below the window)
            ahead_first = ply & ~3u;
            const Philox4 mine = philox4x32_10(seed, game_id, (ahead_first >> 2) + (lane >> 2));
            ahead = philox_word(mine, lane);
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)ahead, (int)(ply - ahead_first));
    };
    // One enumeration site.  `side` is whose action list is built next; after a move it is the other player's, and
    // an empty list there means the game is over: the mover wins if HE could still move, else it is a draw -- one more
    // enumeration, for the mover (`blocked`).  A board that arrives blocked is settled by the same rule with the
    // roles of a move that never happened (settle_blocked).
    uint32_t side = plies & 1u;
    bool blocked = false;
    uint32_t came_from = 0xFFFFFFFFu, came_by = 0;   // the slot and the action that led to the position about to be looked up
    // The memo pays for games that stay among a few positions; a long game that WANDERS misses on every ply and pays for
    // the look-up, the fill and the link bookkeeping all the same: a fifth of its ply (the longest such game, a few hundred
    // plies, is what most launches of this kernel wait for -- an endless game hops through its 4000 plies in less).  So
    // after `cold_limit` look-ups in a row that missed the wave plays `bypass_plies` plies without the memo, then looks
    // again: a game that has settled into a cycle is found out within that many plies.  (Twice the limit for a game's first
    // look-ups: the memo may know nothing of it yet, and a game that arrives cycling should not sit out a bypass first.)
    uint32_t cold = 0, bypass = 0, slot = 0, cold_now = 2u * cold_limit;
    auto same = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    for (;;) {
        // (the state of the ONE game this wave plays is the same on every lane; saying so keeps it in scalar registers
        // and its tests on the scalar unit -- the compiler cannot see it through the loop's memory and lane traffic)
        plies = same(plies);
        side = same(side);
        epoch = same(epoch);
        // The games this pass exists for do not wander: the one endless game of a 2^18-board batch of the default start
        // visits 27 positions in 4096 plies, four of them in its last 2000 (tools/bounce_endless.py).  The action list of a
        // position is a function of the position, so the wave keeps the lists it has built in LDS, keyed by the board and
        // the side (compared in full: a hit IS the list enumerate_wave would build, for any board of the batch), and a ply
        // on a known position costs a look-up instead of the search.  32 sets of two ways, the way not used last is
        // replaced: direct-mapped, two of a game's handful of hot positions shared a slot in one launch of ten and every
        // ply of that game missed (4.7 ms against 2.5).
        const bool with_memo = bypass == 0u;
        if (!with_memo) {
            --bypass;
            enumerate_wave<PMAX>(g, b, side, mv);
        } else {
        uint32_t set;
        {
            uint32_t h = (uint32_t)b.v[0] * 0x9E3779B1u ^ (uint32_t)(b.v[0] >> 32) * 0x85EBCA77u;
            h ^= ((uint32_t)b.v[1] * 0xC2B2AE3Du) ^ ((uint32_t)(b.v[1] >> 32) * 0x27D4EB2Fu);
            h ^= ((uint32_t)b.v[2] * 0x165667B1u) ^ ((uint32_t)(b.v[2] >> 32) * 0xD3A2646Cu);
            h ^= ((uint32_t)b.v[3] * 0xFD7046C5u) ^ ((uint32_t)(b.v[3] >> 32) * 0xB55A4F09u);
            set = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((h ^ (h >> 15)) * 0x2C1B3C6Du) >> (33 - kWaveMemoBits))) ^ side;
            set &= kWaveMemoSlots / 2u - 1u;
        }
        // (both ways' keys and tags are read at once and compared afterwards: one LDS round trip, then the records of
        // the way that hit.  Written with && the compiler reads tag, then the key word by word: six round trips of ~110
        // cycles each, 40 % of a remembered ply.)
        const uint32_t w0 = 2u * set, w1 = w0 + 1u;
        const uint32_t tag0 = memo.tag[w0], tag1 = memo.tag[w1], last = memo.last[set];
        uint32_t differ0 = tag0 ^ side, differ1 = tag1 ^ side;   // (32-bit xor / or chains: 17 instructions a way)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t ka = memo.key[w0][j], kc = memo.key[w1][j];
            differ0 |= ((uint32_t)ka ^ (uint32_t)b.v[j]) | ((uint32_t)(ka >> 32) ^ (uint32_t)(b.v[j] >> 32));
            differ1 |= ((uint32_t)kc ^ (uint32_t)b.v[j]) | ((uint32_t)(kc >> 32) ^ (uint32_t)(b.v[j] >> 32));
        }
        const uint32_t found = (uint32_t)__builtin_amdgcn_readfirstlane((int)((differ0 == 0u ? 1u : 0u) | (differ1 == 0u ? 2u : 0u)));
        const uint32_t way = found ? found >> 1 : ((uint32_t)__builtin_amdgcn_readfirstlane((int)last) ^ 1u) & 1u;   // hit, or the victim
        slot = w0 + way;
        if (found) {
            cold = 0;
            const uint32_t at = lane < (uint32_t)PMAX ? lane : 0u;
            const uint32_t packed = memo.lane[slot][at];
            mv.targets = lane < (uint32_t)PMAX ? memo.targets[slot][at] : 0ull;
            mv.cell = packed & 255u;
            mv.count = lane < (uint32_t)PMAX ? (packed >> 8) & 255u : 0u;
            mv.before = packed >> 16;
            mv.n = (uint32_t)__builtin_amdgcn_readfirstlane((int)memo.n[slot]);
            if (way != ((uint32_t)__builtin_amdgcn_readfirstlane((int)last) & 1u)) {
                if (lane == 0u) memo.last[set] = way;
                wave_lds_sync();
            }
        } else {
            enumerate_wave<PMAX>(g, b, side, mv);
            // the victim way: if it held a position, every link written so far may point at it -- a new epoch
            const uint32_t victim_tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)(way ? tag1 : tag0));
            if (victim_tag != 0xFFFFFFFFu && ++epoch >= epoch_limit) {   // (16 bits in a link: start over with an empty memo)
                for (uint32_t e = lane; e < kWaveMemoSlots; e += BGS_WAVE) memo.tag[e] = 0xFFFFFFFFu;
                epoch = 1;
                wave_lds_sync();
            }
            if (lane < (uint32_t)PMAX) {
                memo.targets[slot][lane] = mv.targets;
                memo.lane[slot][lane] = mv.cell | (mv.count << 8) | (mv.before << 16);
            }
            if (lane < kWaveLinks) memo.link[slot][lane] = 0u;
            if (lane == 0u) {
                memo.key[slot][0] = b.v[0];
                memo.key[slot][1] = b.v[1];
                memo.key[slot][2] = b.v[2];
                memo.key[slot][3] = b.v[3];
                memo.n[slot] = mv.n;
                memo.tag[slot] = side;
                memo.last[set] = way;
            }
            wave_lds_sync();   // (uniform branch, one wave: lane 0's stores before anybody's next look-up)
            if (++cold >= cold_now) {
                cold = 0;
                cold_now = cold_limit;
                bypass = bypass_plies;
            }
        }
        }
        if (came_from != 0xFFFFFFFFu) {
            // the move played last led HERE: remember it (unless this very look-up evicted the position it was played from)
            if (came_from != slot && came_by < kWaveLinks) {
                // (a successor without actions -- the game ends there -- or with more than a byte holds gets no link: 0)
                if (lane == 0u) memo.link[came_from][came_by] = mv.n - 1u < 255u ? (epoch << 16) | (mv.n << 8) | slot : 0u;
                wave_lds_sync();
            }
            came_from = 0xFFFFFFFFu;
        }
        if (blocked) {
            st = mv.n ? side + 1u : BGS_ST_DRAW;
            break;
        }
        if (mv.n == 0) {
            blocked = true;
            side = 1u - side;
            continue;
        }
        if (plies >= max_plies) break;
        const uint32_t idx = sample_index(word_at(plies), mv.n);
        if (with_memo) {
            // does the sampled action have a link?  Then hop: the successor's slot and the number of its actions are in
            // the link, its own links are one look-up away -- the board stays behind until a hop has no link (or the
            // ply cap is reached), and is then taken from the memo's key of the position the hops ended on.  A hop is a
            // chain of dependent steps on a wave that has the SIMD to itself, so it is written for few of them: one test
            // per link (a link is valid iff it carries the current epoch: 0, the empty word, never does), the row's extra
            // word instead of "is the action inside the row".
            uint32_t at = slot, next = idx;
            const uint32_t ply0 = plies;
            for (;;) {
                const uint32_t link = same(memo.link[at][next < kWaveLinks ? next : kWaveLinks]);
                if ((link >> 16) != epoch) break;
                at = link & 255u;
                ++plies;
                if (plies >= max_plies) break;
                next = sample_index(word_at(plies), (link >> 8) & 255u);
            }
            if (plies != ply0) {
                cold = 0;
                side ^= (plies - ply0) & 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t word = memo.key[at][j];
                    b.v[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(word >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)word);
                }
                continue;   // the look-up at the top finds this position, and the ply goes on from its lists
            }
        }
        const bool here = idx - mv.before < mv.count;   // (unsigned: idx < before wraps; count is 0 on every lane that is no source)
        const uint64_t owner = __builtin_amdgcn_ballot_w64(here);   // exactly one lane: the source
        const int from = (__ffsll((unsigned long long)owner) - 1) & 63;
        // its k-th target, found by the CELLS: lane c counts the targets below cell c; the one target cell whose count is k
        // is it (a dozen instructions, most of them scalar, against the forty of the search for the k-th set bit)
        const uint32_t k = idx - (uint32_t)__builtin_amdgcn_readlane((int)mv.before, from);
        const uint32_t t_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv.targets, from);
        const uint32_t t_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv.targets >> 32), from);
        const uint64_t kth = __builtin_amdgcn_ballot_w64(__builtin_amdgcn_mbcnt_hi(t_hi, __builtin_amdgcn_mbcnt_lo(t_lo, 0u)) == k) &
                             (((uint64_t)t_hi << 32) | t_lo);
        const int s = __builtin_amdgcn_readlane((int)mv.cell, from), t = __ffsll((unsigned long long)kth) - 1;
        move_piece(b, s, t);
        ++plies;
        if ((1ull << t) & (g.goal_top | g.goal_bottom)) {
            st = side + 1u;
            break;
        }
        side = 1u - side;
        came_from = with_memo ? slot : 0xFFFFFFFFu;
        came_by = idx;
    }
}


// ---- the TAIL QUEUE (round 6): the games that outlive the bulk pass, finished WHILE the bulk pass runs -- an experiment
// (bounce_tail=1, test build), measured slower than the pass behind the bulk kernel in each of its three forms; no automatic
// plan takes it.
// Until round 5 a rollout was K3p to a ply cap -> positions to planes -> a compaction of the boards still running -> K3w,
// one after the other on the batch's stream: a lone launch spends a third of its time in a nearly empty K3w, which can
// only start when K3p's last wave has left -- although the games it waits for (the one that never ends: 4 000 plies along the
// memo's links; the longest that wanders) mostly crossed the bulk cap in K3p's first third.  With the queue K3p hands a game
// that reaches the bulk cap over at once (and, `handoff_at`, the last boards of a workgroup's last wave): an entry of
// positions / game / plies, then its "complete" word (the launch's serial, release at agent scope).
//   Form 1: K3p's own waves turn into K3w waves when they leave the bulk loop.  No wave leaves before the queue is dry, i.e.
//     when the tail's games have long been waiting; the merged kernel needs 129 VGPRs and 50 KB LDS.  1.79-2.03 ms a lone
//     launch against 1.61 (compile-time geometry, before the shape was retuned).  Removed.
//   Form 2: a second kernel (K3w's code on 256-2048 one-wave workgroups) on a stream of the batch's own, its waves WAITING for
//     tickets' entries while the bulk kernel runs.  1.72 ms at its best against 1.44: the waiting waves' polls (agent-scope
//     loads, a million a millisecond) and their plies compete with the bulk waves.  Removed.
//   Form 3 (what is here): STAGED launches of that kernel, nobody waits on the device: stage k owns the entries [lo, hi) and
//     sits on a stream of its own behind a hipStreamWaitValue32 on the queue's progress word -- the command processor holds
//     the launch back until K3p has allocated `hi` entries (or has finished: its last wave stores the largest value) -- and the
//     last stage, behind K3p itself, owns the rest.  1.51-1.56 ms against 1.44 (256-2048 waves a stage, hand-over at 0 / 16 / 32,
//     wave priority 0 / 1 / 3 all within 5 %).  What the overlap saves -- the long games start at 0.35 ms instead of 0.9 -- the
//     four cross-stream joins, the bulk waves' lost issue slots and the last stage (a long game that crossed the cap late is
//     still the launch's end) take back.
//   counters  tq[0] entries allocated   tq[2] bulk waves that have left   tq[3] the progress word the streams wait on
//   ready[e]  = the launch's serial once entry e is complete
//   entry e   = 8 dwords: positions (4), game, plies, -, -
constexpr uint32_t kTailEntryWords = 8;

// value planes (wave-uniform, scalar registers) from the piece list of a queue entry
template <int PMAX, class GEO>
__device__ __forceinline__ Board planes_from_positions(const GEO& g, const uint32_t (&pos)[4]) {
    Board out;
    out.v[0] = out.v[1] = out.v[2] = out.v[3] = 0;
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        if (k < (int)g.piece_count) {
            const uint64_t bit = 1ull << ((pos[k >> 2] >> (8 * (k & 3))) & 63u);
            const uint32_t v = g.piece_value[k];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((v >> q) & 1u) out.v[q] |= bit;
        }
    return out;
}

// ... and back, where a tail wave's game has ended: the board goes to memory in K3p's format (k_bounce_positions_to_planes
// runs behind the kernel for every board).  Pieces are numbered by ascending (value, cell) at the start and keep their
// values, so ANY numbering that is ascending in the value reproduces the planes: rank = pieces of smaller value + pieces of
// the same value on lower cells.  Lane k holds the k-th occupied cell (as in enumerate_wave).
template <int PMAX>
__device__ __forceinline__ void wave_store_positions(const Board& b, uint64_t* __restrict__ planes, int64_t n, int64_t i) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t occ = occupancy(b);
    const uint32_t pieces = (uint32_t)__popcll(occ);
    const bool alive = lane < pieces;
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(occ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)occ, 0u));
    const bool occupied = ((occ >> lane) & 1ull) != 0ull;
    const uint32_t told = (uint32_t)__builtin_amdgcn_ds_permute((int)((occupied ? below : 63u) << 2), (int)lane);
    const uint32_t cell = alive ? told : 0u;
    const uint32_t value = alive ? value_at(b, (int)cell) : 0xFFu;
    uint32_t rank = 0;
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        const uint32_t vj = (uint32_t)__builtin_amdgcn_readlane((int)value, j);
        rank += (vj < value || (vj == value && (uint32_t)j < lane)) ? 1u : 0u;
    }
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)((alive ? rank : 63u) << 2), (int)cell);
    const uint32_t part = alive ? (got & 63u) << (8u * (lane & 3u)) : 0u;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < PMAX; ++k) w[k >> 2] |= (uint32_t)__builtin_amdgcn_readlane((int)part, k);
    if (lane == 0u) {
        planes[i] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        if (PMAX > 8) planes[n + i] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    }
}

// the device-wide pool of parked boards (see the kernel): boards a workgroup parks, dwords per entry (positions, game, plies)
constexpr uint32_t kPoolCap = 64, kPoolWords = 6;

template <int PMAX, int BLOCK>
struct ParkedPieces {
    static constexpr uint32_t WAVES = BLOCK / BGS_WAVE, CAP = 64;   // (a wave parks at most 63 boards: park_at <= 63)
    uint32_t pos[PMAX / 4][WAVES][CAP];
    uint32_t game[WAVES][CAP];
    uint32_t plies[WAVES][CAP];
    uint32_t count[WAVES];       // entries of the segment; published once, after the entries
    uint32_t head[WAVES];        // entries claimed so far (may run past count)
    uint32_t active;             // waves still in their loop
};

// BLOCK threads per workgroup (256, 512 or 1024): the waves of a workgroup share their drain through LDS, so a larger
// workgroup ends with fewer half-empty waves (one per workgroup carries the workgroup's longest games to their end)
// Waves per SIMD: with the opening book's loads the kernel needs 98 VGPRs -- two more than five waves a SIMD allow.  Held to
// 96 (amdgpu_waves_per_eu) the 8- and 12-piece instantiations fit without a spill (the 16-piece one spills 44 bytes), and a
// fifth wave a SIMD hides more of a ply's 17 us of dependent instructions: tools/k3p_waves_ab.sh has the A/B.
#ifndef BGS_K3P_WAVES
#define BGS_K3P_WAVES 0
#endif
#if BGS_K3P_WAVES > 0
#define BGS_K3P_OCCUPANCY __attribute__((amdgpu_waves_per_eu(BGS_K3P_WAVES, BGS_K3P_WAVES)))
#else
#define BGS_K3P_OCCUPANCY
#endif
// what the tail role of a launch needs (TAIL; see "the TAIL QUEUE" above)
struct TailArgs {
    uint32_t* counters;       // tq[0..3]
    uint32_t* ready;          // [capacity]
    uint32_t* entries;        // [capacity][kTailEntryWords]
    uint32_t capacity;
    uint32_t serial;          // this launch's "entry complete" word
    uint32_t final_cap;       // the rollout's ply cap (the bulk loop stops at max_plies)
    uint32_t handoff_at;      // a workgroup's last wave hands its boards over at this many or fewer (0: plays them to the bulk cap)
    uint32_t limit;           // tail waves that may wait for entries at a time (the last bulk wave always stays)
    uint32_t epoch_limit, cold_limit, bypass_plies;   // K3w's memo policy
    uint32_t prio;            // s_setprio of the tail kernel's waves
};

template <int PMAX, int BLOCK, bool TAIL, class GEO>
__global__ void __launch_bounds__(BLOCK) BGS_K3P_OCCUPANCY
k_bounce_rollout_pieces(GEO g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                        uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                        unsigned long long* __restrict__ steps, uint32_t chunk, uint32_t* __restrict__ queue, uint32_t park_at,
                        uint32_t* gpool, const uint32_t* __restrict__ book_links, const BookEntry* __restrict__ book_table,
                        uint32_t book_depth, uint32_t book_n0, TailArgs tail) {
    __shared__ ParkedPieces<PMAX, BLOCK> parked;
    __shared__ uint32_t book_lds[kBookLdsLinks];   // the opening book's links of levels 1 and 2 (4.25 KB)
    // ... and the boards of the wave's current chunk as the book hands them out: a chunk (at most 64 games) is walked through
    // the book by ALL lanes at once when it is drawn from the queue -- one philox call and one walk per lane, every lane busy
    // -- and parked here, a 64-byte line a game; a lane that takes a game reads its line.  (Walked where the lane takes the
    // game, the walk ran in almost every iteration -- some lane of 64 always finishes -- for two or three lanes: ~100 VALU
    // an iteration, 4 % of a ply.)
    __shared__ uint4 opened_lds[BLOCK / BGS_WAVE][4][BGS_WAVE];
    if (book_depth >= 2u)
        for (uint32_t i = threadIdx.x; i < kBookLdsLinks; i += BLOCK) book_lds[i] = book_links[i];
    Lands<PMAX> lands;   // (registers; valid from a ply's search to its move)
    constexpr uint32_t WAVES = ParkedPieces<PMAX, BLOCK>::WAVES;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < WAVES) {
        parked.count[threadIdx.x] = 0u;
        parked.head[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) parked.active = WAVES;
    __syncthreads();
    // (the drain -- parked boards, `active`, the last wave sweeping up -- is k_bounce_rollout_flat's, see there)
    bool last = false;
    auto bump = [&](uint32_t* word, uint32_t by) {
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(word, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane(old);
    };
    auto leave = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        const uint32_t before = bump(&parked.active, ~0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        return before;
    };
    const uint32_t total = (uint32_t)n;
    uint32_t begin = 0, avail = 0, taken = 0;   // the wave's current chunk: boards begin + [taken, avail)
    bool dry = false;                             // the queue has nothing left
    // ---- the DEVICE-WIDE pool of parked boards (round 4).  Inside a workgroup the drain is shared through LDS; what is
    // left is ONE wave per workgroup that carries the workgroup's last boards to their end at a few lanes (2^18 boards
    // on 512 waves: 268 wave iterations where 224 full ones would do).  That wave now parks them in global memory --
    // segment = its workgroup -- and leaves; the last waves of other workgroups, draining themselves, adopt them into
    // their idle lanes, so the launch's stragglers collect in ever fewer, fuller waves.  Same rules as in LDS, one level
    // up: entries are written, then the segment's count is published (release, agent scope); `left` counts the
    // workgroups that are gone, and the wave that finds itself the last of the LAUNCH takes back what it parked and
    // sweeps up every segment (its acquire of `left` has seen every other workgroup's release).  Nobody ever waits.
    //   gpool[0] workgroups that have left   gpool[1] boards parked so far   gpool[2] boards claimed so far
    //   gpool[4 + s] count of segment s      gpool[4 + G + s] head of segment s      entries behind the counters
    const uint32_t n_groups = gridDim.x;
    uint32_t* const g_count = gpool ? gpool + 4 : nullptr;
    uint32_t* const g_head = gpool ? gpool + 4 + n_groups : nullptr;
    uint32_t* const g_entries = gpool ? gpool + 4 + 2 * n_groups : nullptr;   // [segment][kPoolCap][kPoolWords]
    bool glast = false;   // this wave is the last of the launch
    auto gbump = [&](uint32_t* word, uint32_t by, bool acq_rel) {
        uint32_t old = 0;
        if (lane == 0)
            old = acq_rel ? __hip_atomic_fetch_add(word, by, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                          : __hip_atomic_fetch_add(word, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (uint32_t)__builtin_amdgcn_readfirstlane(old);
    };

    // the lanes with `mine` set hand their boards (positions, game, plies) to the tail queue: one allocation a wave
    auto push_tail = [&](bool mine, const PieceBoard<PMAX>& board, uint32_t the_game, uint32_t the_plies) {
        const uint64_t who = __builtin_amdgcn_ballot_w64(mine);
        if (!who) return;
        const uint32_t base = gbump(tail.counters, (uint32_t)__popcll(who), false);
        if (mine) {
            const uint32_t e = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(who >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)who, 0u));
            uint32_t* slot = tail.entries + (size_t)e * kTailEntryWords;   // (e < capacity: a game is handed over at most once)
            *reinterpret_cast<uint4*>(slot) = make_uint4(board.pos[0], PMAX > 4 ? board.pos[PMAX > 4 ? 1 : 0] : 0u,
                                                         PMAX > 8 ? board.pos[PMAX > 8 ? 2 : 0] : 0u, PMAX > 12 ? board.pos[PMAX > 12 ? 3 : 0] : 0u);
            *reinterpret_cast<uint2*>(slot + 4) = make_uint2(the_game, the_plies);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(tail.ready + e, tail.serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the word the STREAMS of the staged tail launches wait on (hipStreamWaitValue32: the command processor watches it, no
        // wave does): entries allocated so far
        if (lane == 0u) (void)__hip_atomic_fetch_max(tail.counters + 3, base + (uint32_t)__popcll(who), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };

    PieceBoard<PMAX> b;
    pieces_from_start(g, b);   // (every lane always holds valid positions: phase A runs on all 64 lanes)
    PieceMoves mv;
    mv.counts = 0;
    mv.n = 0;
    mv.row_base = 0;
#pragma unroll
    for (int j = 0; j < kMaxTrackedColumns / 2; ++j) mv.reach[j] = 0;
    uint32_t st = 0, plies = 0, first_ply = 0, game = 0, stepped = 0;
    bool has = false;      // this lane holds a board
    bool search = false;   // ... whose side to move has no action list yet
    bool pending = false;  // ... whose side to move is blocked: the NEXT search counts the other side's actions and settles it
    Philox4 blk;
    blk.v[0] = blk.v[1] = blk.v[2] = blk.v[3] = 0;
    bool have_block = false;

#ifdef BGS_BOUNCE_STATS
    uint32_t stat_iters = 0, stat_search = 0, stat_drain_iters = 0;
#endif
    for (;;) {
#ifdef BGS_BOUNCE_STATS
        ++stat_iters;
#endif
        // ---- refill: free lanes take the next boards of the wave's chunk; an empty chunk is replaced from the queue
        const uint64_t need = __builtin_amdgcn_ballot_w64(!has);
        if (need && taken >= avail && !dry) {
            uint32_t next = 0;
            if (lane == 0u) next = atomicAdd(queue, chunk);
            begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
            taken = 0;
            avail = begin < total ? (total - begin < chunk ? total - begin : chunk) : 0u;
            dry = avail == 0u;
            if (book_depth && avail) {
                // the opening book (see above), for the whole chunk in lock step: game begin + lane's first plies are a walk
                // along the book's links with the game's own draws -- the words of philox block 0 -- ending on a 64-byte line
                const Philox4 d = philox4x32_10(seed, first_game + (uint64_t)(begin + (lane < avail ? lane : 0u)), 0u);
                uint32_t at = sample_index(d.v[0], book_n0);
                if (book_depth >= 2u) {
                    const uint32_t e = book_lds[at];
                    at = (e >> kBookLinkBits) + sample_index(d.v[1], e & kBookLinkMax);
                }
                if (book_depth >= 3u) {
                    const uint32_t e = book_lds[kBookCap1 + at];
                    at = (e >> kBookLinkBits) + sample_index(d.v[2], e & kBookLinkMax);
                }
                if (book_depth >= 4u) {
                    const uint32_t e = book_links[kBookCap1 + kBookCap2 + at];
                    at = (e >> kBookLinkBits) + sample_index(d.v[3], e & kBookLinkMax);
                }
                const uint4* line = reinterpret_cast<const uint4*>(book_table + at);
                const uint4 q0 = line[0], q1 = line[1], q2 = line[2], q3 = line[3];
                opened_lds[w][0][lane] = q0;
                opened_lds[w][1][lane] = q1;
                opened_lds[w][2][lane] = q2;
                opened_lds[w][3][lane] = q3;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (!has && taken + rank < avail) {
                game = begin + taken + rank;
                if (book_depth) {
                    const uint32_t slot = taken + rank;   // (a chunk is at most 64 games: bounce_rollout)
                    have_block = false;
                    const uint4 q0 = opened_lds[w][0][slot], q1 = opened_lds[w][1][slot], q2 = opened_lds[w][2][slot],
                                q3 = opened_lds[w][3][slot];
                    const uint32_t where[4] = {q0.x, q0.y, q0.z, q0.w};
#pragma unroll
                    for (int j = 0; j < PMAX / 4; ++j) b.pos[j] = where[j];
                    b.idx[0] = ((uint64_t)q1.y << 32) | q1.x;
                    b.idx[1] = ((uint64_t)q1.w << 32) | q1.z;
                    b.idx[2] = ((uint64_t)q2.y << 32) | q2.x;
                    b.idx[3] = ((uint64_t)q2.w << 32) | q2.z;
                    b.occ = ((uint64_t)q3.y << 32) | q3.x;
                    st = q3.z & 255u;
                    plies = (q3.z >> 8) & 255u;
                } else {
                    pieces_from_start(g, b);
                    st = g.init_status;
                    plies = 0;
                    have_block = false;
                }
                first_ply = 0;
                has = true;
                search = st == BGS_ST_RUNNING;
                pending = false;
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }
        const bool draining = dry && taken >= avail;  // (wave-uniform) nothing left to draw
        bool adopted_now = false;     // (wave-uniform) this iteration's adoption attempt found something
        bool lds_exhausted = false;   // this iteration's adoption attempt took everything the workgroup's LDS pool held
#ifdef BGS_BOUNCE_STATS
        if (draining) ++stat_drain_iters;
#endif
#ifdef BGS_DRAIN_PRIO
        if (draining) __builtin_amdgcn_s_setprio(BGS_DRAIN_PRIO);
#endif
        if (draining && need) {
            // ---- idle lanes adopt parked boards.  count[] and head[] are adjacent: lanes 0 .. 2 WAVES - 1 fetch them in one access
            const uint32_t wanted = (uint32_t)__popcll(need);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            uint32_t word = 0;
            if (lane < 2u * WAVES) word = __hip_atomic_load(&parked.count[0] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            uint32_t assigned = 0;
            bool adopted_one = false;
            for (uint32_t sgm = 0; sgm < WAVES; ++sgm) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(word, sgm);
                const uint32_t h = (uint32_t)__builtin_amdgcn_readlane(word, WAVES + sgm);
                if (assigned < wanted && h < c) {
                    const uint32_t old = bump(&parked.head[sgm], wanted - assigned);
                    const uint32_t got = old < c ? (c - old < wanted - assigned ? c - old : wanted - assigned) : 0u;
                    if (!has && rank >= assigned && rank < assigned + got) {
                        const uint32_t e = old + (rank - assigned);
#pragma unroll
                        for (int j = 0; j < PMAX / 4; ++j) b.pos[j] = parked.pos[j][sgm][e];
                        game = parked.game[sgm][e];
                        plies = parked.plies[sgm][e];
                        first_ply = plies;  // (the wave that parked it has counted the plies up to here)
                        st = BGS_ST_RUNNING;
                        has = true;
                        have_block = false;
                        search = true;
                        pending = false;
                        adopted_one = true;
                    }
                    assigned += got;
                }
            }
            lds_exhausted = assigned < wanted;   // (wave-uniform) idle lanes are left over: the LDS pool is empty
            // ... and the last wave of a workgroup, when its own workgroup has nothing more for it, from the device-wide pool:
            // one segment per iteration (the scan stops at the first segment that still holds boards)
            const uint64_t need2 = __builtin_amdgcn_ballot_w64(!has);
            if (gpool && last && need2 && lds_exhausted) {
                uint32_t parked_total = 0, claimed_total = 0;
                if (lane == 0) {
                    parked_total = __hip_atomic_load(gpool + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    claimed_total = __hip_atomic_load(gpool + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                parked_total = (uint32_t)__builtin_amdgcn_readfirstlane(parked_total);
                claimed_total = (uint32_t)__builtin_amdgcn_readfirstlane(claimed_total);
                if (parked_total > claimed_total) {
                    const uint32_t wanted2 = (uint32_t)__popcll(need2);
                    const uint32_t rank2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(need2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need2, 0u));
                    uint32_t seg = ~0u, seg_count = 0;
                    for (uint32_t base = 0; base < n_groups && seg == ~0u; base += 64u) {
                        const uint32_t sgm = base + lane;
                        uint32_t c = 0, h = 0;
                        if (sgm < n_groups && sgm != blockIdx.x) {
                            c = __hip_atomic_load(g_count + sgm, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                            h = __hip_atomic_load(g_head + sgm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        const uint64_t open = __builtin_amdgcn_ballot_w64(h < c);
                        if (open) {
                            const int src = __ffsll((unsigned long long)open) - 1;
                            seg = base + (uint32_t)src;
                            seg_count = (uint32_t)__builtin_amdgcn_readlane(c, src);
                        }
                    }
                    if (seg != ~0u) {
                        const uint32_t old = gbump(g_head + seg, wanted2, false);
                        const uint32_t got = old < seg_count ? (seg_count - old < wanted2 ? seg_count - old : wanted2) : 0u;
                        if (got) (void)gbump(gpool + 2, got, false);
                        if (!has && rank2 < got) {
                            const uint32_t* e = g_entries + ((size_t)seg * kPoolCap + (old + rank2)) * kPoolWords;
#pragma unroll
                            for (int j = 0; j < PMAX / 4; ++j) b.pos[j] = e[j];
                            game = e[4];
                            plies = e[5];
                            first_ply = plies;
                            st = BGS_ST_RUNNING;
                            has = true;
                            have_block = false;
                            search = true;
                            pending = false;
                            adopted_one = true;
                        }
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(adopted_one)) {
                adopted_now = true;
                PieceBoard<PMAX> fresh = b;
                pieces_rebuild(g, fresh);
                if (adopted_one) b = fresh;
            }
        }

        // ---- the action counts of the boards that need them (new boards, boards that have just moved); a board whose
        // side to move has no action is settled here, also at the ply cap (the transition that blocked it counts)
        if (__builtin_amdgcn_ballot_w64(search)) {
#ifdef BGS_BOUNCE_STATS
            ++stat_search;
#endif
            // A board whose side to move turns out to have no action ("blocked") is settled by the OTHER side's count
            // (Appendix B rule 7: the other side wins if it could move, else a draw).  That count is not computed here and
            // now -- it would cost every lane of the wave a second search and a third one to restore its own masks (round
            // 3; 1.2 % of the games end this way, i.e. one wave iteration in 40 paid three searches) -- but by the next
            // iteration's search, which runs anyway: the lane is `pending`, its side for that search is the other one.
            const uint32_t side = pending ? 1u - (plies & 1u) : (plies & 1u);
            land_all<PMAX>(g, b, side, lands);
            close_over_bounces<PMAX>(g, lands);
            count_from_lands<PMAX>(g, b, side, search, lands, mv);
            if (search && pending) {
                st = mv.n ? (1u - (plies & 1u)) + 1u : BGS_ST_DRAW;
                pending = false;
            } else if (search && mv.n == 0u) {
                pending = true;
            }
            search = pending;
        }
        // (a pending board is neither running nor stopped: it sits out this iteration's move and store)
        const bool run = has && !pending && st == BGS_ST_RUNNING && plies < max_plies;

        // ---- boards that stopped go to memory and free their lane (from the start position: every board is written)
        if (TAIL) {
            // a game that goes on beyond the bulk cap: to the tail queue (its action list, if one was just counted, is left behind)
            const bool over = has && !run && !pending && st == BGS_ST_RUNNING && plies < tail.final_cap;
            push_tail(over, b, game, plies);
            if (over) {
                stepped += plies - first_ply;
                has = false;
                search = false;
            }
        }
        if (!TAIL && tail.entries) {
            // the games the NEXT pass has to finish -- still running at this pass's cap, below the rollout's -- go on its work
            // list here (round 6; until then a compaction kernel read every board's status behind this one: 44 us for a lone launch)
            const bool more = has && !run && !pending && st == BGS_ST_RUNNING && plies < tail.final_cap;
            const uint64_t who = __builtin_amdgcn_ballot_w64(more);
            if (who) {
                const uint32_t base = gbump(tail.counters, (uint32_t)__popcll(who), false);
                if (more) tail.entries[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(who >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)who, 0u))] = game;
            }
        }
        if (has && !run && !pending) {
            const int64_t i = game;
            store_positions<PMAX>(planes, n, i, b);
            status[i] = (uint8_t)st;
            plies_buf[i] = (uint16_t)plies;
            reward[i] = reward_pair(st);
            stepped += plies - first_ply;
            has = false;
        }
        if (!__builtin_amdgcn_ballot_w64(has)) {
            if (!draining) continue;    // (a chunk of boards that were not running: draw the next one)
            // The last wave leaves only after an adoption attempt with ALL its lanes idle has found nothing: if its
            // lanes were busy at the top of this iteration and their boards all stopped in it (a ply cap does that), the
            // boards other waves parked meanwhile would otherwise never be played.
            if (last && (need != ~0ull || adopted_now)) continue;   // (boards adopted in this very iteration may all have stopped in it)
            if (last) {
                // the workgroup is done.  The launch: whoever is not its last wave simply goes; the last one sweeps the
                // device-wide pool (an attempt with all lanes idle has just found nothing: it is empty)
                if (!gpool || glast) break;
                if (gbump(gpool, 1u, true) + 1u < n_groups) break;
                glast = true;
                continue;
            }
            if (leave() > 1u) break;    // others are still running: whatever gets parked later is theirs
            last = true;                // everybody else has left: sweep up what they parked
            continue;
        }
        if (TAIL && draining && last && tail.handoff_at && lds_exhausted) {
            // ---- the workgroup's last wave hands its last boards to the tail queue and follows them (same condition as the
            // device-wide pool's below: only in an iteration whose adoption attempt has emptied the workgroup's LDS pool)
            const uint32_t left = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(has));
            if (left <= tail.handoff_at) {
                push_tail(has, b, game, plies);
                if (has) stepped += plies - first_ply;
                has = false;
                break;
            }
        }
        if (draining && last && !glast && gpool && park_at && lds_exhausted) {
            // ---- the workgroup's last wave parks its last boards for the other workgroups' last waves -- but only in an
            // iteration whose adoption attempt has emptied the workgroup's LDS pool: what sits there is visible to this wave
            // alone, and a wave whose lanes were all busy at the top of the iteration has not looked (the round-3 bug, one
            // level up: boards parked in LDS at ply 0 were lost when the last wave parked ITS boards and left)
            const uint64_t still = __builtin_amdgcn_ballot_w64(has);
            const uint32_t left = (uint32_t)__popcll(still);
            if (left <= park_at && left <= kPoolCap) {
                const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(still >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)still, 0u));
                if (has) {
                    uint32_t* slot = g_entries + ((size_t)blockIdx.x * kPoolCap + e) * kPoolWords;
#pragma unroll
                    for (int j = 0; j < PMAX / 4; ++j) slot[j] = b.pos[j];
                    slot[4] = game;
                    slot[5] = plies;
                }
                // every lane's entry is in memory before the count says so (release at agent scope: the adopters sit on
                // other CUs, behind other L2s)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) {
                    __hip_atomic_store(g_count + blockIdx.x, left, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    (void)__hip_atomic_fetch_add(gpool + 1, left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (gbump(gpool, 1u, true) + 1u < n_groups) {  // parked; another workgroup is still there to adopt them
                    if (has) stepped += plies - first_ply;
                    break;
                }
                // nobody is: take back what has not been adopted (an adopter that has left has finished its boards)
                const uint32_t adopted = gbump(g_head + blockIdx.x, left, false);
                (void)gbump(gpool + 2, left > adopted ? left - adopted : 0u, false);
                if (has && e < adopted) {
                    stepped += plies - first_ply;
                    has = false;
                }
                glast = true;
            }
        }
        if (draining && !last) {
            const uint64_t still = __builtin_amdgcn_ballot_w64(has);  // (every board still held is running here)
            const uint32_t left = (uint32_t)__popcll(still);
            if (left <= park_at) {
                const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(still >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)still, 0u));
                if (has) {
#pragma unroll
                    for (int j = 0; j < PMAX / 4; ++j) parked.pos[j][w][e] = b.pos[j];
                    parked.game[w][e] = game;
                    parked.plies[w][e] = plies;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_store(&parked.count[w], left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (leave() > 1u) {  // parked; somebody is still there to adopt them
                    if (has) stepped += plies - first_ply;
                    break;
                }
                // nobody is: take back what has not been adopted (an adopter that has left has finished its boards)
                const uint32_t adopted = bump(&parked.head[w], left);
                if (has && e < adopted) {
                    stepped += plies - first_ply;
                    has = false;
                }
                last = true;
            }
        }

        // ---- one ply on every running board
        if (run) {
            if (!have_block || (plies & 3u) == 0u) {
                blk = philox4x32_10(seed, first_game + (uint64_t)game, plies >> 2);
                have_block = true;
            }
            const uint32_t mover = plies & 1u;
            int s, t;
            pick_from_lands<PMAX>(g, b, mover, mv, lands, sample_index(philox_word(blk, plies), mv.n), s, t);
            move_piece_on(b, s, t);
            ++plies;
            if ((1ull << t) & (g.goal_top | g.goal_bottom)) {
                // a piece in the goal row ends the game (98.8 % of the games end this way): to memory at once, so that
                // the lane takes its next board at the top of the next iteration instead of idling through it
                st = mover + 1u;
                const int64_t i = game;
                store_positions<PMAX>(planes, n, i, b);
                status[i] = (uint8_t)st;
                plies_buf[i] = (uint16_t)plies;
                reward[i] = reward_pair(st);
                stepped += plies - first_ply;
                has = false;
            } else {
                search = true;
            }
        }
    }
    if (TAIL) {
        // this wave's part is over: its entries, then its departure (the tail kernel's end signal is the last one's)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t gone = gbump(tail.counters + 2, 1u, true) + 1u;
        // the launch's last bulk wave releases every staged tail launch that is still waiting for its threshold
        if (gone == gridDim.x * (BLOCK / BGS_WAVE) && lane == 0u)
            __hip_atomic_store(tail.counters + 3, 0x7FFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef BGS_BOUNCE_STATS
    if (lane == 0) {  // words 1..4 of shard 0's cache line are free
        atomicAdd(steps + 1, (unsigned long long)stat_iters);
        atomicAdd(steps + 2, (unsigned long long)stat_search);
        atomicAdd(steps + 3, (unsigned long long)stat_drain_iters);
    }
#endif
    add_steps(steps, stepped);
}

// the tail kernel (see "the TAIL QUEUE"): K3w's code on one-wave workgroups.  Nobody waits for K3p: the launches of this
// kernel are STAGED -- stage k owns the queue's entries [lo, hi) and sits on a stream of its own behind a
// hipStreamWaitValue32 on the queue's progress word, i.e. the command processor holds the launch back until K3p has allocated
// `hi` entries (or has finished: its last wave stores the largest value); the last stage, behind K3p itself, owns everything
// from its `lo` on.  So when a stage runs, every entry of its range exists (or never will): its waves draw tickets from the
// stage's own counter and leave when the range is used up -- no wave ever holds a ticket for an entry that is still to come,
// and no two waves meet on a compare-and-swap (a first version claimed the queue's head that way: 2048 waves, 5000 entries,
// 200 ms a launch).
template <int PMAX, class GEO>
__global__ void __launch_bounds__(BGS_WAVE)
k_bounce_tail(GEO g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
              uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, unsigned long long* __restrict__ steps,
              TailArgs tail, uint32_t lo, uint32_t hi, uint32_t* __restrict__ stage_counter) {
    // (K3w raises its waves' priority: they are its launch's critical path.  Beside the bulk kernel that is a choice: tail.prio)
    if (tail.prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (tail.prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (tail.prio == 1u) __builtin_amdgcn_s_setprio(1);
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ WaveMemo<PMAX> memo;
    uint32_t epoch = 1, stepped = 0;
    bool memo_ready = false;
    // (the stage was released because `hi` entries exist, or because the bulk kernel is over: either way what is allocated now
    // is all this stage will ever own)
    uint32_t allocated = 0;
    if (lane == 0u) allocated = __hip_atomic_load(tail.counters, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    allocated = (uint32_t)__builtin_amdgcn_readfirstlane((int)allocated);
    const uint32_t end = hi < allocated ? hi : allocated;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0u) t = __hip_atomic_fetch_add(stage_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= end) break;
        if (!memo_ready) {
            wave_memo_reset(memo, lane);
            memo_ready = true;
        }
        for (;;) {   // (the entry's last word is written a few instructions after its allocation)
            uint32_t flag = 0;
            if (lane == 0u) flag = __hip_atomic_load(tail.ready + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)flag) == tail.serial) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t* e = tail.entries + (size_t)t * kTailEntryWords;
        uint32_t where[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) where[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[j]);
        const uint32_t game = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[4]);
        uint32_t plies = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[5]);
        const uint32_t came_with = plies;
        Board b = planes_from_positions<PMAX>(g, where);
        uint32_t st = BGS_ST_RUNNING;
        wave_play_game<PMAX>(g, memo, epoch, b, st, plies, seed, first_game + (uint64_t)game, tail.final_cap, tail.epoch_limit,
                             tail.cold_limit, tail.bypass_plies);
        wave_store_positions<PMAX>(b, planes, n, (int64_t)game);
        if (lane == 0u) {
            status[game] = (uint8_t)st;
            plies_buf[game] = (uint16_t)plies;
            reward[game] = reward_pair(st);
            stepped += plies - came_with;
        }
    }
    add_steps(steps, stepped);
}

__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_step_actions(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status,
                      uint16_t* __restrict__ plies_buf, uint16_t* __restrict__ reward, int64_t n,
                      const int32_t* __restrict__ moves, int32_t* __restrict__ result,
                      unsigned long long* __restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        const int sx = moves[4 * i], sy = moves[4 * i + 1], tx = moves[4 * i + 2], ty = moves[4 * i + 3];
        int32_t rc = 0;
        if (sx >= 0) {
            rc = -2;  // BGS_ERR_ILLEGAL
            const bool inside = sx < g.w && sy >= 0 && sy < g.h && tx >= 0 && tx < g.w && ty >= 0 && ty < g.h;
            if (inside && status[i] == BGS_ST_RUNNING && plies_buf[i] < kMaxPlies) {
                Board b = load_board(planes, n, i);
                uint32_t plies = plies_buf[i];
                const uint32_t mover = plies & 1u;
                const uint64_t occ = occupancy(b);
                const int s = sy * g.w + sx, t = ty * g.w + tx;
                if (((movable(g, occ, mover) >> s) & 1ull) && ((reach(g, b, occ, mover, s) >> t) & 1ull)) {
                    move_piece(b, s, t);
                    ++plies;
                    uint32_t n_next;
                    const uint32_t st = settle(g, b, mover, t, n_next);
                    store_board(planes, n, i, b);
                    plies_buf[i] = (uint16_t)plies;
                    if (st != BGS_ST_RUNNING) {
                        status[i] = (uint8_t)st;
                        reward[i] = reward_pair(st);
                    }
                    stepped = 1;
                    rc = 0;
                }
            }
        }
        if (result) result[i] = rc;
    }
    add_steps(steps, stepped);
}

// The object API's round trip on a small batch (see k_connect_transition): the chosen move, if any, then grid, player,
// winner, plies, the targets record and the reward pair of every board in one launch.  One board, by one thread:
__device__ __forceinline__ uint32_t bounce_transition_one(const BounceGeom& g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status,
                                                          uint16_t* __restrict__ plies_buf, uint16_t* __restrict__ reward, int64_t n, int64_t i,
                                                          const int32_t* __restrict__ moves, int32_t* __restrict__ result,
                                                          int8_t* __restrict__ grid, int8_t* __restrict__ player, int8_t* __restrict__ winner,
                                                          int32_t* __restrict__ plies_out, uint64_t* __restrict__ targets,
                                                          uint16_t* __restrict__ reward_out) {
    uint32_t stepped = 0;
    Board b = load_board(planes, n, i);
    uint32_t st = status[i];
    uint32_t plies = plies_buf[i];
    uint16_t pair = reward[i];
    if (moves) {
        const int sx = moves[4 * i], sy = moves[4 * i + 1], tx = moves[4 * i + 2], ty = moves[4 * i + 3];
        int32_t rc = 0;
        if (sx >= 0) {
            rc = -2;  // BGS_ERR_ILLEGAL
            const bool inside = sx < g.w && sy >= 0 && sy < g.h && tx >= 0 && tx < g.w && ty >= 0 && ty < g.h;
            if (inside && st == BGS_ST_RUNNING && plies < kMaxPlies) {
                const uint32_t mover = plies & 1u;
                const uint64_t occ = occupancy(b);
                const int s = sy * g.w + sx, t = ty * g.w + tx;
                if (((movable(g, occ, mover) >> s) & 1ull) && ((reach(g, b, occ, mover, s) >> t) & 1ull)) {
                    move_piece(b, s, t);
                    ++plies;
                    uint32_t n_next;
                    const uint32_t after = settle(g, b, mover, t, n_next);
                    store_board(planes, n, i, b);
                    plies_buf[i] = (uint16_t)plies;
                    if (after != BGS_ST_RUNNING) {
                        st = after;
                        pair = reward_pair(st);
                        status[i] = (uint8_t)st;
                        reward[i] = pair;
                    }
                    stepped = 1;
                    rc = 0;
                }
            }
        }
        result[i] = rc;
    }
    const int hw = g.h * g.w;
    for (int c = 0; c < hw; ++c) grid[i * hw + c] = (int8_t)value_at(b, c);
    player[i] = (int8_t)(plies & 1u);
    winner[i] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
    plies_out[i] = (int32_t)plies;
    const uint64_t occ = occupancy(b);
    const uint32_t mover = plies & 1u;
    const uint64_t src = st == BGS_ST_RUNNING ? movable(g, occ, mover) : 0ull;
    const int row = src ? (int)(((uint32_t)(__ffsll((unsigned long long)src) - 1) * g.inv_w) >> 16) : 0;
    for (int x = 0; x < g.w; ++x) {
        const int c = row * g.w + x;
        targets[i * (g.w + 1) + x] = ((src >> c) & 1ull) ? reach(g, b, occ, mover, c) : 0ull;
    }
    targets[i * (g.w + 1) + g.w] = src ? (uint64_t)row : ~0ull;
    reward_out[i] = pair;
    return stepped;
}

__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_transition(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                    uint16_t* __restrict__ reward, int64_t n, const int32_t* __restrict__ moves, int32_t* __restrict__ result,
                    unsigned long long* __restrict__ steps, int8_t* __restrict__ grid, int8_t* __restrict__ player,
                    int8_t* __restrict__ winner, int32_t* __restrict__ plies_out, uint64_t* __restrict__ targets,
                    uint16_t* __restrict__ reward_out, uint32_t* __restrict__ done, uint32_t ticket) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n)
        stepped = bounce_transition_one(g, planes, status, plies_buf, reward, n, i, moves, result, grid, player, winner, plies_out, targets,
                                        reward_out);
    add_steps(steps, stepped);
    if (done) {  // (see publish_ticket in connect_kernels.hip: one workgroup, records in host memory)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// packed planes -> reference layout int8[n][H][W]: one lane expands one board into the workgroup's LDS tile, the
// workgroup streams the tile out with 16-byte stores
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_unpack(BounceGeom g, const uint64_t* __restrict__ planes, int64_t n, int8_t* __restrict__ grid) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    const int hw = g.h * g.w;
    const int64_t base = (int64_t)blockIdx.x * BGS_BLOCK;
    const int64_t i = base + threadIdx.x;
    if (i < n) {
        const Board b = load_board(planes, n, i);
        uint8_t* mine = tile + threadIdx.x * hw;
        for (int c = 0; c < hw; ++c) mine[c] = (uint8_t)value_at(b, c);
    }
    __syncthreads();
    const int64_t boards = n - base < BGS_BLOCK ? n - base : BGS_BLOCK;
    tile_to_global(tile, reinterpret_cast<uint8_t*>(grid) + base * hw, (uint32_t)(boards * hw));
}

__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_meta(const uint8_t* __restrict__ status, const uint16_t* __restrict__ plies_buf, int64_t n,
              int8_t* __restrict__ player, uint8_t* __restrict__ ended, int8_t* __restrict__ winner,
              int32_t* __restrict__ plies) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t st = status[i];
    const uint32_t p = plies_buf[i];
    if (player) player[i] = (int8_t)(p & 1u);
    if (ended) ended[i] = st != 0;
    if (winner) winner[i] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
    if (plies) plies[i] = (int32_t)p;
}

// targets[n][W + 1]: legal landing cells of the piece in column x of the active row, then the active row itself
// (all ones when the board has ended or nothing can move); count[n] = number of actions
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_targets(BounceGeom g, const uint64_t* __restrict__ planes, const uint8_t* __restrict__ status,
                 const uint16_t* __restrict__ plies_buf, int64_t n, uint64_t* __restrict__ targets,
                 int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const Board b = load_board(planes, n, i);
    const uint64_t occ = occupancy(b);
    const uint32_t player = plies_buf[i] & 1u;
    const uint64_t src = status[i] == BGS_ST_RUNNING ? movable(g, occ, player) : 0ull;
    const int row = src ? (int)(((uint32_t)(__ffsll((unsigned long long)src) - 1) * g.inv_w) >> 16) : 0;
    int32_t total = 0;
    for (int x = 0; x < g.w; ++x) {
        const int c = row * g.w + x;
        const uint64_t t = ((src >> c) & 1ull) ? reach(g, b, occ, player, c) : 0ull;
        if (targets) targets[i * (g.w + 1) + x] = t;
        total += __popcll(t);
    }
    if (targets) targets[i * (g.w + 1) + g.w] = src ? (uint64_t)row : ~0ull;
    if (count) count[i] = total;
}

// reference layout -> packed planes.  Running boards must keep the goal rows empty; a running board whose side to
// move is blocked is settled here, exactly as a transition would have settled it.
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_pack(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
              uint16_t* __restrict__ reward, int64_t n, const int8_t* __restrict__ grid, const int8_t* __restrict__ player,
              const int8_t* __restrict__ winner, const int32_t* __restrict__ plies, int32_t* __restrict__ result) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int hw = g.h * g.w;
    Board b;
    b.v[0] = b.v[1] = b.v[2] = b.v[3] = 0;
    bool ok = true;
    for (int c = 0; c < hw; ++c) {
        const int v = grid[i * hw + c];
        if (v < 0 || v > BGS_BOUNCE_MAX_VALUE) ok = false;
#pragma unroll
        for (int p = 0; p < 4; ++p) b.v[p] |= (uint64_t)((v >> p) & 1) << c;
    }
    const int wv = winner ? winner[i] : -1;
    if (wv < -1 || wv > 2) ok = false;
    uint32_t st = wv == -1 ? 0u : (wv == 2 ? BGS_ST_DRAW : (uint32_t)(wv + 1));
    if (st == BGS_ST_RUNNING && (occupancy(b) & (g.goal_top | g.goal_bottom))) ok = false;
    const int pl = player ? player[i] : 0;
    if (pl != 0 && pl != 1) ok = false;
    int32_t np = plies ? plies[i] : pl;
    if (np < 0 || np > 65535 || (np & 1) != pl) ok = false;
    if (ok) {
        if (st == BGS_ST_RUNNING && count_actions(g, b, occupancy(b), (uint32_t)pl) == 0)
            st = settle_blocked(g, b, (uint32_t)pl);
        store_board(planes, n, i, b);
        status[i] = (uint8_t)st;
        plies_buf[i] = (uint16_t)np;
        reward[i] = reward_pair(st);
    }
    if (result) result[i] = ok ? 0 : -1;
}

// boards that are still running below `cap` plies -> worklist (order irrelevant), one atomic per wave
__global__ void __launch_bounds__(BGS_BLOCK)
k_bounce_compact(const uint8_t* __restrict__ status, const uint16_t* __restrict__ plies_buf, int64_t n, uint32_t cap,
                 uint32_t* __restrict__ worklist, uint32_t* __restrict__ work_count) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    const bool running = i < n && status[i] == BGS_ST_RUNNING && plies_buf[i] < cap;
    const uint64_t mask = __builtin_amdgcn_ballot_w64(running);
    if (!mask) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(work_count, (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (running)
        worklist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint32_t)i;
}

// (K3w's enumeration, memo and ply loop -- enumerate_wave, WaveMemo, wave_play_game -- are defined in front of K3p, whose waves
// run them too when they have left the bulk loop: see there)
template <int PMAX, class GEO>
__global__ void __launch_bounds__(BGS_WAVE)
k_bounce_rollout_wave(GEO g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                      uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                      unsigned long long* __restrict__ steps, const uint32_t* __restrict__ worklist,
                      const uint32_t* __restrict__ work_count, uint32_t epoch_limit, uint32_t cold_limit, uint32_t bypass_plies) {
    static_assert(PMAX <= 16, "the prefix sum runs over one 16-lane row");
    __builtin_amdgcn_s_setprio(3);   // (see k_bounce_rollout: these waves are the launch's critical path)
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ WaveMemo<PMAX> memo;   // the wave's memo of action lists (see wave_play_game)
    uint32_t epoch = 1;
    wave_memo_reset(memo, lane);
    const uint32_t total = *work_count;
    uint32_t stepped = 0;
    for (uint32_t entry = blockIdx.x; entry < total; entry += gridDim.x) {
        const uint32_t game = (uint32_t)__builtin_amdgcn_readfirstlane((int)worklist[entry]);
        const int64_t i = game;
        Board b = load_board(planes, n, i);
#pragma unroll
        for (int j = 0; j < 4; ++j)   // (wave-uniform: keep the planes in scalar registers)
            b.v[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b.v[j] >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b.v[j]);
        uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)status[i]);
        uint32_t plies = (uint32_t)__builtin_amdgcn_readfirstlane((int)plies_buf[i]);
        if (st != BGS_ST_RUNNING || plies >= max_plies) continue;
        if ((uint32_t)__popcll(occupancy(b)) > (uint32_t)PMAX) {
            // more pieces than lanes provided for (a batch loaded from memory may hold anything): lane 0 plays the board with
            // the thread-per-board code -- correct, slow, and rare (the host does not send batches CONFIGURED with that many;
            // the compile-time geometry is only used for boards that descend from its own start position: never)
            if constexpr (std::is_same<GEO, BounceGeom>::value)
            if (lane == 0u) {
                const uint32_t before = plies;
                (void)play<false>(g, b, st, plies, seed, first_game + (uint64_t)game, max_plies);
                store_board(planes, n, i, b);
                status[i] = (uint8_t)st;
                plies_buf[i] = (uint16_t)plies;
                reward[i] = reward_pair(st);
                stepped += plies - before;
            }
            continue;
        }
        const uint32_t first_ply = plies;
        wave_play_game<PMAX>(g, memo, epoch, b, st, plies, seed, first_game + (uint64_t)game, max_plies, epoch_limit, cold_limit, bypass_plies);
        if (lane == 0u) {
            store_board(planes, n, i, b);
            status[i] = (uint8_t)st;
            plies_buf[i] = (uint16_t)plies;
            reward[i] = reward_pair(st);
            stepped += plies - first_ply;
        }
    }
    add_steps(steps, stepped);
}

// The object API's round trip for ONE board (the engines of simulator.game.bounce hold one-board batches) on one wave, a piece
// per lane: K3w's enumerate_wave in place of the thread-per-board kernel's walks.  A transition is a legality test, the
// terminal test and the next action list -- up to thirteen closures walked one after the other by one thread (~4 us of a
// 14 us call); here at most three enumerations of ~0.4 us.  A board of more than PMAX pieces (a State loaded from JSON may
// hold anything) is served by lane 0 with the thread-per-board code.  Same records, same ticket.
template <int PMAX>
__global__ void __launch_bounds__(BGS_WAVE)
k_bounce_transition_wave(BounceGeom g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                         uint16_t* __restrict__ reward, const int32_t* __restrict__ moves, int32_t* __restrict__ result,
                         unsigned long long* __restrict__ steps, int8_t* __restrict__ grid, int8_t* __restrict__ player,
                         int8_t* __restrict__ winner, int32_t* __restrict__ plies_out, uint64_t* __restrict__ targets,
                         uint16_t* __restrict__ reward_out, uint32_t* __restrict__ done, uint32_t ticket) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t stepped = 0;
    Board b = load_board(planes, 1, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        b.v[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b.v[j] >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b.v[j]);
    if ((uint32_t)__popcll(occupancy(b)) > (uint32_t)PMAX) {
        if (lane == 0u)
            stepped = bounce_transition_one(g, planes, status, plies_buf, reward, 1, 0, moves, result, grid, player, winner, plies_out,
                                            targets, reward_out);
    } else {
        uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)status[0]);
        uint32_t plies = (uint32_t)__builtin_amdgcn_readfirstlane((int)plies_buf[0]);
        uint32_t pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)reward[0]);
        WaveMoves mv;
        bool listed = false;   // mv is the action list of (b, plies & 1)
        if (moves) {
            const int sx = __builtin_amdgcn_readfirstlane(moves[0]), sy = __builtin_amdgcn_readfirstlane(moves[1]);
            const int tx = __builtin_amdgcn_readfirstlane(moves[2]), ty = __builtin_amdgcn_readfirstlane(moves[3]);
            int32_t rc = 0;
            if (sx >= 0) {
                rc = -2;  // BGS_ERR_ILLEGAL
                const bool inside = sx < g.w && sy >= 0 && sy < g.h && tx >= 0 && tx < g.w && ty >= 0 && ty < g.h;
                if (inside && st == BGS_ST_RUNNING && plies < kMaxPlies) {
                    const uint32_t mover = plies & 1u;
                    const int s = sy * g.w + sx, t = ty * g.w + tx;
                    enumerate_wave<PMAX>(g, b, mover, mv);
                    listed = true;
                    const bool mine = mv.count != 0u && mv.cell == (uint32_t)s && ((mv.targets >> t) & 1ull);
                    if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
                        move_piece(b, s, t);
                        ++plies;
                        uint32_t after = BGS_ST_RUNNING;
                        if ((1ull << t) & (g.goal_top | g.goal_bottom)) {
                            after = mover + 1u;
                            listed = false;
                        } else {
                            enumerate_wave<PMAX>(g, b, 1u - mover, mv);   // the next player's list: the terminal test AND the observation
                            if (mv.n == 0) {
                                enumerate_wave<PMAX>(g, b, mover, mv);
                                after = mv.n ? mover + 1u : BGS_ST_DRAW;
                                listed = false;
                            }
                        }
                        if (lane == 0u) {
                            store_board(planes, 1, 0, b);
                            plies_buf[0] = (uint16_t)plies;
                        }
                        if (after != BGS_ST_RUNNING) {
                            st = after;
                            pair = reward_pair(st);
                            if (lane == 0u) {
                                status[0] = (uint8_t)st;
                                reward[0] = (uint16_t)pair;
                            }
                        }
                        stepped = lane == 0u ? 1u : 0u;
                        rc = 0;
                    }
                }
            }
            if (lane == 0u) result[0] = rc;
        }
        // the records
        const int hw = g.h * g.w;
        if ((int)lane < hw) grid[lane] = (int8_t)value_at(b, (int)lane);   // a cell per lane: one store instruction
        const uint64_t occ = occupancy(b);
        const uint32_t mover = plies & 1u;
        const uint64_t src = st == BGS_ST_RUNNING ? movable(g, occ, mover) : 0ull;
        const int row = src ? (int)(((uint32_t)(__ffsll((unsigned long long)src) - 1) * g.inv_w) >> 16) : 0;
        if (src && !listed) enumerate_wave<PMAX>(g, b, mover, mv);
        for (int x = 0; x < g.w; ++x) {   // (uniform: column x's mask from the lane that holds the column's piece, if it can move)
            const uint64_t holder = src ? __builtin_amdgcn_ballot_w64(mv.count != 0u && mv.cell == (uint32_t)(row * g.w + x)) : 0ull;
            uint64_t mask = 0;
            if (holder) {
                const int from = (__ffsll((unsigned long long)holder) - 1) & 63;
                mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mv.targets >> 32), from) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv.targets, from);
            }
            if (lane == 0u) targets[x] = mask;
        }
        if (lane == 0u) {
            targets[g.w] = src ? (uint64_t)row : ~0ull;
            player[0] = (int8_t)(plies & 1u);
            winner[0] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
            plies_out[0] = (int32_t)plies;
            reward_out[0] = (uint16_t)pair;
        }
    }
    add_steps(steps, stepped);
    if (done) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + BGS_BLOCK - 1) / BGS_BLOCK); }

}  // namespace

void bounce_reset(const bgs_batch* b) {
    hipLaunchKernelGGL(k_bounce_reset, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes, b->d_status,
                       b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, 0);
}

void bounce_reset_ended(const bgs_batch* b) {
    hipLaunchKernelGGL(k_bounce_reset, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes, b->d_status,
                       b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, 1);
}

void bounce_step_random(const bgs_batch* b, uint64_t seed, uint32_t count) {
    for (uint32_t q = 0; q < count; ++q)
    hipLaunchKernelGGL((k_bounce_play<true, false>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes,
                       b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game,
                       kMaxPlies, b->d_steps);
}

// One launch of the fused rollout: over the whole batch (worklist == nullptr) or over a work list.
// group = lanes per board (1 or 8), wps = waves per SIMD the grid is sized for.
// final_cap > cap: the bulk pass of a plan whose tail is K3w's code INSIDE the same launch (K3p only; see "the TAIL QUEUE")
// next_list / next_count (K3p without the tail inside): where the bulk pass leaves the work list of the pass behind it
static void launch_rollout(const bgs_batch* b, uint64_t seed, uint32_t cap, bool from_initial, int group, int wps,
                           const uint32_t* worklist, const uint32_t* work_count, uint32_t* queue, uint32_t final_cap = 0,
                           uint32_t* next_list = nullptr, uint32_t* next_count = nullptr) {
    if (group == 64) {   // K3w: a work list of a few very long games, one wave each (a wave strides over the list)
        auto launch_wave = [&](auto pmax_tag) {
            constexpr int PMAX = decltype(pmax_tag)::value;
            // one-wave workgroups; a wave without a list entry leaves at once, so the grid is sized for the longest list the
            // automatic plan produces (5 % of 2^18 boards after a short bulk pass: one or two entries a wave; 2048 / 4096 /
            // 8192 / 16384 waves read 2.39 / 2.61 / 2.78 / 2.83 x 10^9 one launch at a time, and the same with 4 and 20 in flight)
            const unsigned wave_grid = b->bounce_wave_grid > 0 ? (unsigned)b->bounce_wave_grid : 8192u;
            auto go = [&](auto geo) {
                hipLaunchKernelGGL((k_bounce_rollout_wave<PMAX, decltype(geo)>), dim3(wave_grid), dim3(BGS_WAVE), 0,
                                   b->stream, geo, b->d_planes, b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n,
                                   seed, b->first_game, cap, b->d_steps, worklist, work_count,
                                   b->bounce_epoch_limit >= 2 && b->bounce_epoch_limit < (int)kWaveEpochLimit ? (uint32_t)b->bounce_epoch_limit : kWaveEpochLimit,
                                   (uint32_t)b->bounce_memo_cold, (uint32_t)b->bounce_memo_bypass);
            };
            // (the default board from its own start position: the compile-time geometry, see bounce_unit.h)
            if (PMAX == 12 && from_initial && b->bounce_static_geom && bounce_is_default(b->bg)) go(DefaultBounceGeom{});
            else go(b->bg);
        };
        // (from_initial here: every board of the rollout descends from the configured start position, so none holds more
        // pieces than it; otherwise 16 lanes, and a board with more than that is played by lane 0)
        if (from_initial && b->bg.piece_count >= 1 && b->bg.piece_count <= 8) launch_wave(std::integral_constant<int, 8>{});
        else if (from_initial && b->bg.piece_count >= 1 && b->bg.piece_count <= 12) launch_wave(std::integral_constant<int, 12>{});
        else launch_wave(std::integral_constant<int, 16>{});
        return;
    }
    const int64_t slots_per_wave = BGS_WAVE / group;
    int64_t per_wave, waves;
    if (worklist) {
        // the list's length is only known on the device: a modest fixed grid whose waves draw chunks of one wave-load
        // from the pass's queue until it is dry (never more waves than the whole batch could need)
        per_wave = slots_per_wave;
        waves = (b->n + per_wave - 1) / per_wave;
        if (waves > 1024) waves = 1024;
    } else {
        const int64_t resident = (int64_t)b->num_cus * 4 * wps;
        per_wave = (b->n + resident - 1) / resident;
        if (per_wave < slots_per_wave) per_wave = slots_per_wave;
        waves = (b->n + per_wave - 1) / per_wave;
    }
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    auto launch = [&](auto initial_tag, auto group_tag) {
        constexpr bool INITIAL = decltype(initial_tag)::value;
        constexpr int GL = decltype(group_tag)::value;
        hipLaunchKernelGGL((k_bounce_rollout<INITIAL, GL>), dim3(blocks), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes,
                           b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game, cap,
                           b->d_steps, (uint32_t)per_wave, worklist, work_count, queue);
    };
    auto launch_flat = [&](auto initial_tag) {
        constexpr bool INITIAL = decltype(initial_tag)::value;
        const size_t tile = (size_t)2 * kMaxTrackedColumns * BGS_BLOCK * sizeof(uint32_t);
        // a grid that fills the chip (flat_wps waves per SIMD, never more waves than 64-board chunks); every wave
        // draws its boards from the shared queue
        const uint32_t chunk = (uint32_t)b->bounce_flat_chunk;
        int64_t flat_waves = b->bounce_flat_waves > 0 ? b->bounce_flat_waves : (int64_t)b->num_cus * 4 * b->bounce_flat_wps;
        const int64_t most = (b->n + 63) / 64;    // (never more waves than 64-board loads)
        if (flat_waves > most) flat_waves = most;
        hipLaunchKernelGGL((k_bounce_rollout_flat<INITIAL>), dim3((unsigned)((flat_waves + 3) / 4)), dim3(BGS_BLOCK), tile,
                           b->stream, b->bg, b->d_planes, b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n,
                           seed, b->first_game, cap, b->d_steps, chunk, worklist, work_count, queue, (uint32_t)b->bounce_park);
    };
    auto launch_pieces = [&](auto pmax_tag, auto block_tag, auto tail_tag) {
        constexpr int PMAX = decltype(pmax_tag)::value;
        constexpr int BLOCK = decltype(block_tag)::value;
        constexpr bool TAIL = decltype(tail_tag)::value;
        const size_t tile = 0;  // (the landing masks live in registers; LDS only holds the parked boards)
        const uint32_t chunk0 = (uint32_t)b->bounce_flat_chunk;
        // Every ply costs a wave the same whatever the number of its lanes that still hold a game, so what counts is how
        // full the waves stay: few, long-lived waves (boards_per_wave boards each, drawn from the queue) spend most
        // of their life refilling and little of it draining -- when many launches share the chip.  A launch that is
        // alone wants more, shorter-lived waves: see bounce_shape().
        const int64_t per_wave = bounce_shape(b->launches_in_flight).boards_per_wave;
        int64_t flat_waves = (int64_t)b->num_cus * 4 * b->bounce_flat_wps;
        if (b->bounce_flat_waves > 0) flat_waves = b->bounce_flat_waves;
        else if (b->n / per_wave < flat_waves) flat_waves = b->n / per_wave > 256 ? b->n / per_wave : 256;
        const int64_t most = (b->n + 63) / 64;    // (never more waves than 64-board loads)
        if (flat_waves > most) flat_waves = most;
        constexpr int per_block = BLOCK / BGS_WAVE;
        if (tile > 48 * 1024) {  // beyond the default dynamic-LDS limit (gfx950 has 160 KB per CU)
            static bool raised = false;  // (per instantiation; the attribute belongs to the function, not the launch)
            if (!raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bounce_rollout_pieces<PMAX, BLOCK, TAIL, BounceGeom>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile);
                raised = true;
            }
        }
        const unsigned groups = (unsigned)((flat_waves + per_block - 1) / per_block);
        // the opening book of the start position (bounce_book_acquire): games start `depth` plies in -- unless the cap is shorter
        const uint32_t book_depth = b->book_table && b->book_depth > 0 && cap >= (uint32_t)b->book_depth ? (uint32_t)b->book_depth : 0u;
        // (with the book a chunk is walked by the 64 lanes of the wave that draws it: a whole wave's worth, never more)
        const uint32_t chunk = book_depth ? 64u : chunk0;
        // the device-wide pool of parked boards: counters zeroed per launch (one small fill on the stream)
        const int park_at = b->bounce_pieces_park >= 0 ? b->bounce_pieces_park : bounce_shape(b->launches_in_flight).park;
        uint32_t* pool = b->bounce_pool && park_at > 0 && groups >= 2 && groups <= (unsigned)BGS_BOUNCE_POOL_GROUPS ? b->d_pool : nullptr;
        if (pool) (void)hipMemsetAsync(pool, 0, sizeof(uint32_t) * (4 + 2 * (size_t)groups), b->stream);
        TailArgs tail{};
        if (TAIL) {
            // the tail queue: counters among the rollout's list counters (cleared at its start: words 4 .. 7, the lengths of passes
            // no automatic plan has), "entry complete" words in the work list's region (one per game; they hold launch serials, so
            // they are cleared only when something else has written there), the entries in the staging region (32 bytes a game;
            // nothing else of this batch runs beside its rollout)
            const BounceShape shape = bounce_shape(b->launches_in_flight);
            if (b->tail_flags_dirty || b->tail_serial == 0xFFFFFFFFu) {
                (void)hipMemsetAsync(b->d_worklist, 0, sizeof(uint32_t) * (size_t)b->n, b->stream);
                b->tail_flags_dirty = 0;
                b->tail_serial = 0;
            }
            tail.counters = b->d_work_count + 4;
            tail.ready = b->d_worklist;
            tail.entries = reinterpret_cast<uint32_t*>(b->d_staging);
            tail.capacity = (uint32_t)b->n;
            tail.serial = ++b->tail_serial;
            tail.final_cap = final_cap;
            tail.handoff_at = b->bounce_tail_handoff >= 0 ? (uint32_t)b->bounce_tail_handoff : (uint32_t)shape.handoff_at;
            tail.limit = b->bounce_tail_limit > 0 ? (uint32_t)b->bounce_tail_limit : (uint32_t)shape.tail_waves;
            tail.epoch_limit = b->bounce_epoch_limit >= 2 && b->bounce_epoch_limit < (int)kWaveEpochLimit ? (uint32_t)b->bounce_epoch_limit : kWaveEpochLimit;
            tail.cold_limit = (uint32_t)b->bounce_memo_cold;
            tail.bypass_plies = (uint32_t)b->bounce_memo_bypass;
            tail.prio = (uint32_t)b->bounce_tail_prio;
        } else if (next_list && next_count && final_cap > cap) {
            tail.entries = next_list;
            tail.counters = next_count;
            tail.final_cap = final_cap;
        }
        // (Tried, round 4: a kernel specialised on "exactly PMAX pieces" -- every "is there a piece k" test decided at compile
        // time.  18 % fewer static instructions, one basic block a phase, and 174 VGPRs; held to 128 it spills 43 and reads
        // 1.14 against 1.26 x 10^10 with 20 launches in flight.)
        // TAIL: the staged launches of the tail kernel, each on a stream of the batch's own behind (1) everything the batch's
        // stream holds so far (the fork event) and (2) a wait for the queue's progress word to reach the stage's threshold; the
        // last one behind the bulk kernel itself.  The batch's stream goes on behind them all (the join events).  The bulk kernel
        // is enqueued FIRST: on whatever hardware queue a stage's wait ends up, the kernel that satisfies it is ahead of it.
        constexpr int kStages = bgs_batch::kTailStages;   // (the last stage is the one behind the bulk kernel)
        bool forked = false;
        if (TAIL) {
            if (!b->tail_stream[0]) {
                bool ok = true;
                for (int k = 0; k < kStages && ok; ++k) {
                    ok = hipStreamCreateWithFlags(&b->tail_stream[k], hipStreamNonBlocking) == hipSuccess &&
                         hipEventCreateWithFlags(&b->tail_join[k], hipEventDisableTiming) == hipSuccess;
                }
                ok = ok && hipEventCreateWithFlags(&b->tail_fork, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&b->tail_bulk_done, hipEventDisableTiming) == hipSuccess;
                if (!ok) b->tail_stream[0] = nullptr;   // (no staged launches: one tail launch behind the bulk kernel, on its stream)
            }
            forked = b->tail_stream[0] && hipEventRecord(b->tail_fork, b->stream) == hipSuccess;
            for (int k = 0; k < kStages && forked; ++k) forked = hipStreamWaitEvent(b->tail_stream[k], b->tail_fork, 0) == hipSuccess;
        }
        auto go = [&](auto geo) {
            hipLaunchKernelGGL((k_bounce_rollout_pieces<PMAX, BLOCK, TAIL, decltype(geo)>), dim3(groups),
                               dim3(BLOCK), tile, b->stream, geo, b->d_planes, b->d_status, b->d_plies,
                               reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game, cap, b->d_steps, chunk, queue,
                               (uint32_t)park_at, pool, b->book_links,
                               reinterpret_cast<const BookEntry*>(b->book_table), book_depth, b->book_n0, tail);
            if (TAIL) {
                // (a stage's ticket counter: words 12 .. 15 of the rollout's counters, cleared with them; the sweep-up launch of a
                // failed enqueue shares the last stage's)
                auto tail_launch = [&](hipStream_t on, unsigned waves, uint32_t lo, uint32_t hi, int stage) {
                    hipLaunchKernelGGL((k_bounce_tail<PMAX, decltype(geo)>), dim3(waves), dim3(BGS_WAVE), 0, on, geo, b->d_planes,
                                       b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game,
                                       b->d_steps, tail, lo, hi, b->d_work_count + 12 + stage);
                };
                if (!forked) {
                    tail_launch(b->stream, 8192u, 0u, 0xFFFFFFFFu, kStages - 1);
                } else {
                    // stage k is released when the bulk kernel has handed over threshold[k] games (2 % of a batch outlive an
                    // 80-ply bulk pass: ~5 000 of 2^18)
                    const uint32_t expected = (uint32_t)(b->n / 50) + 1u;
                    bool ok = hipEventRecord(b->tail_bulk_done, b->stream) == hipSuccess;
                    uint32_t lo = 0;
                    int k = 0;
                    for (; k < kStages && ok; ++k) {
                        // stage k owns [lo, hi): released at `hi` entries; the last stage owns the rest and follows the bulk kernel
                        const uint32_t hi = k + 1 < kStages ? (k == 0 ? 32u : expected * (uint32_t)k / (uint32_t)(kStages - 1)) : 0xFFFFFFFFu;
                        if (k + 1 < kStages) ok = hipStreamWaitValue32(b->tail_stream[k], tail.counters + 3, hi, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess;
                        else ok = hipStreamWaitEvent(b->tail_stream[k], b->tail_bulk_done, 0) == hipSuccess;
                        if (!ok) break;
                        tail_launch(b->tail_stream[k], k + 1 < kStages ? tail.limit : 8192u, lo, hi, k);
                        ok = hipEventRecord(b->tail_join[k], b->tail_stream[k]) == hipSuccess && hipStreamWaitEvent(b->stream, b->tail_join[k], 0) == hipSuccess;
                        lo = hi;
                    }
                    if (!ok) {   // (something could not be enqueued: wait for what was, then the rest on the batch's stream)
                        for (int j = 0; j < kStages; ++j) (void)hipStreamSynchronize(b->tail_stream[j]);
                        tail_launch(b->stream, 8192u, lo, 0xFFFFFFFFu, kStages - 1);
                    }
                }
            }
        };
        // (the default board: the compile-time geometry, see bounce_unit.h -- 256-thread workgroups only, the shape every plan uses)
        if constexpr (PMAX == 12 && BLOCK == 256) {
            if (b->bounce_static_geom && bounce_is_default(b->bg)) go(DefaultBounceGeom{});
            else go(b->bg);
        } else {
            go(b->bg);
        }
        // every board was written as the positions of its pieces: the value planes for all of them, at full lanes
        hipLaunchKernelGGL(k_bounce_positions_to_planes, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes, b->n);
    };
    auto with_block = [&](auto pmax_tag) {
        if (final_cap > cap && !next_list) launch_pieces(pmax_tag, std::integral_constant<int, 256>{}, std::true_type{});
        else if (b->bounce_block >= 1024) launch_pieces(pmax_tag, std::integral_constant<int, 1024>{}, std::false_type{});
        else if (b->bounce_block >= 512) launch_pieces(pmax_tag, std::integral_constant<int, 512>{}, std::false_type{});
        else launch_pieces(pmax_tag, std::integral_constant<int, 256>{}, std::false_type{});
    };
    // K3p: from the start position, no work list, at most 16 pieces
    if (group == 1 && b->bounce_flat && b->bounce_pieces && from_initial && !worklist && b->bg.piece_count >= 1 &&
        b->n < (int64_t)0xFFFFFFFFu) {
        if (b->bg.piece_count <= 8) with_block(std::integral_constant<int, 8>{});
        else if (b->bg.piece_count <= 12) with_block(std::integral_constant<int, 12>{});
        else with_block(std::integral_constant<int, 16>{});
        return;
    }
    auto with_group = [&](auto initial_tag) {
        if (group == 1 && b->bounce_flat) launch_flat(initial_tag);   // one lane per board, flattened search
        else if (group == 1) launch(initial_tag, std::integral_constant<int, 1>{});
        else launch(initial_tag, std::integral_constant<int, 8>{});
    };
    if (from_initial) with_group(std::true_type{});
    else with_group(std::false_type{});
}

// Random Bounce games have no length bound: most end within a few dozen plies, a few run for hundreds and some
// never end (they stop at max_plies).  One launch that plays every game to the end spends most of its instruction
// issue on waves in which a single long game is still alive.  The rollout therefore runs in PASSES with growing ply
// caps: after a pass the boards that are still running are compacted into a work list (k_bounce_compact) and the next
// pass plays only those.  A board resumes exactly where it stopped (state in memory, RNG keyed by game id and ply), so
// the result is the one-launch result bit for bit.  The passes are enqueued back to back on the batch's stream; the
// list lengths never visit the host.
// The automatic plan is two passes: the kernel that suits the batch (K3p / K3f with one lane per board for large ones:
// fewest instructions per ply; 8 lanes per board for small ones) up to a short cap, then ONE BOARD PER WAVE (K3w: the
// shortest ply, and a memo of action lists -- what a handful of very long games is bound by).  Start positions of more
// than 16 pieces keep round 3's plan (one launch, or K3p + an 8-lane tail).  BGS_EXPERIMENT "bounce_plan=cap:lanes,..." overrides.
void bounce_rollout(const bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags) {
    uint32_t cap = max_plies < 0 ? 0u : (uint32_t)max_plies;
    if (cap > kMaxPlies) cap = kMaxPlies;  // plies are stored as uint16
    const bool from_initial = (flags & 1u) != 0u;
    if (b->bg.w <= kMaxTrackedColumns && !b->rollout_generic) {
        // [0, MAX_PASSES): list lengths; [MAX_PASSES, 2 * MAX_PASSES): the passes' work queues (flat kernel)
        uint32_t* list = b->d_worklist;
        uint32_t* counts = b->d_work_count;
        uint32_t* queues = b->d_work_count + BGS_BOUNCE_MAX_PASSES;
        (void)hipMemsetAsync(counts, 0, sizeof(uint32_t) * 2 * BGS_BOUNCE_MAX_PASSES, b->stream);
        // The automatic plan.  Random Bounce games are short (mean 28 plies, 1 in 10^4 beyond 256) -- except the few per
        // 2^18 that never end and run into max_plies.  Such a game is a chain of thousands of dependent plies; inside
        // the bulk launch it would keep a whole wave of the one-lane-per-board kernel alive (every ply at the cost of
        // 64 boards), so the bulk pass stops at bounce_shape().tail_cap plies and the stragglers -- compacted into a work
        // list on the device -- are finished 8 lanes to a board, the kernel with the shortest ply, at raised wave priority.
        int passes = b->bounce_passes;
        uint32_t pass_cap_of[BGS_BOUNCE_MAX_PASSES];
        int pass_group_of[BGS_BOUNCE_MAX_PASSES];
        for (int i = 0; i < passes; ++i) {
            pass_cap_of[i] = b->bounce_pass_cap[i];
            pass_group_of[i] = b->bounce_pass_group[i];
        }
        // lanes per board of the first (or only) pass: 8 for small batches (shortest ply), 1 for large ones (fewest instructions).
        // From the start position the large-batch kernel is K3p, whose ply is 17 us deep for a lone wave: one launch at a
        // time it only wins from 2^17 boards (2^16: 2.05 ms against 1.75 with 8 lanes + K3w; 2^17: 2.46 against 2.62), with
        // four and more launches in flight from 2^15 as before
        const BounceShape shape = bounce_shape(b->launches_in_flight);
        const bool piece_list = b->bounce_flat && b->bounce_pieces && from_initial && b->bg.piece_count >= 1;
        int lanes = b->bounce_group;
        if (b->bounce_group_auto && piece_list && b->launches_in_flight < 4) lanes = b->n >= 131072 ? 1 : 8;
        // K3w as the last pass: any batch whose configured start position has at most 16 pieces (piece_count = 0: more)
        const bool wave_tail = b->bounce_wave_pass && b->bg.piece_count >= 1 && b->bg.piece_count <= BGS_BOUNCE_MAX_PIECES;
        if (b->bounce_plan_auto && lanes == 1 && piece_list && cap > 2u * (uint32_t)shape.tail_cap) {
            passes = 2;
            pass_cap_of[0] = (uint32_t)shape.tail_cap;
            pass_group_of[0] = 1;
            pass_cap_of[1] = cap;
            // ... by ONE BOARD PER WAVE (K3w) where it can take them: its ply is shorter still, and it remembers the action lists
            // of the positions it has seen, which is what a game that never ends consists of.  Measured (tools/k3w_probe.sh,
            // 2^18 default boards): one launch at a time 1.40 -> 1.92-2.09 x 10^9 env-steps/s, 20 in flight 1.42 -> 1.54-1.57 x
            // 10^10; an 8-lane pass in between (to 2x / 4x the bulk cap) reads the same.  experiment bounce_wave_pass=0: the 8-lane
            // tail as before.
            pass_group_of[1] = wave_tail ? 64 : 8;
        } else if (b->bounce_plan_auto && wave_tail) {
            // every other batch (small ones, boards loaded from memory): a short first pass on the kernel it always had, then
            // K3w.  2^10 / 2^12 / 2^14 / 2^15 boards from the start, one launch at a time: 0.67 / 0.84 / 0.98 / 1.70 ms -> 0.44 /
            // 0.55 / 0.78 / 1.07 ms (first pass to 16 plies below 2^13 boards, else 32).  A large batch loaded from memory
            // (one lane per board, K3f) used to keep a whole wave alive for every game that never ends.
            const uint32_t first = lanes == 1 ? (uint32_t)shape.tail_cap : (b->n >= 8192 ? 32u : 16u);
            if (cap > 2u * first) {
                passes = 2;
                pass_cap_of[0] = first;
                pass_group_of[0] = lanes;
                pass_cap_of[1] = cap;
                pass_group_of[1] = 64;
            }
        }
        for (int i = 1; i < passes; ++i)   // (a plan from the environment: K3w only where the configured position fits its lanes)
            if (pass_group_of[i] == 64 && !(b->bg.piece_count >= 1 && b->bg.piece_count <= BGS_BOUNCE_MAX_PIECES)) pass_group_of[i] = 8;
        if (passes >= 1 && pass_group_of[0] == 64) pass_group_of[0] = 1;
        if (passes <= 1) {  // single launch (experiment bounce_plan=single, a plan with one entry, a ply cap too short for two passes)
            launch_rollout(b, seed, cap, from_initial, lanes, lanes == 1 ? b->rollout_wps : 8, nullptr, nullptr, queues);
            return;
        }
        bool listed = false;   // pass 0 left pass 1's work list behind
        for (int pass = 0; pass < passes; ++pass) {
            const bool last = pass + 1 == passes;
            const uint32_t pass_cap = (last || pass_cap_of[pass] > cap) ? cap : pass_cap_of[pass];
            const int group = pass_group_of[pass];
            if (pass == 0) {
                // K3p with K3w as the second and last pass; experiment bounce_tail=1: the tail kernel BESIDE the bulk kernel ("the
                // TAIL QUEUE" -- measured slower than the pass behind it, so no automatic plan takes it)
                const bool fused = b->bounce_tail > 0 && passes == 2 && group == 1 && pass_group_of[1] == 64 && piece_list && from_initial &&
                                   b->bounce_block < 512 && b->bg.piece_count <= BGS_BOUNCE_MAX_PIECES && b->d_pool != nullptr &&
                                   b->n < (int64_t)0xFFFFFFFFu && b->staging_bytes >= (size_t)b->n * kTailEntryWords * sizeof(uint32_t);
                if (fused) {
                    launch_rollout(b, seed, pass_cap, from_initial, group, b->rollout_wps, nullptr, nullptr, queues, cap);
                    return;
                }
                // (K3p writes the next pass's work list itself: no compaction kernel behind it)
                listed = group == 1 && piece_list && from_initial && b->bg.piece_count <= BGS_BOUNCE_MAX_PIECES && b->n < (int64_t)0xFFFFFFFFu &&
                         pass_cap < cap;
                if (listed) b->tail_flags_dirty = 1;
                launch_rollout(b, seed, pass_cap, from_initial, group, group == 1 ? b->rollout_wps : 8, nullptr, nullptr, queues,
                               listed ? cap : 0u, listed ? list : nullptr, listed ? counts + 1 : nullptr);
            } else {
                b->tail_flags_dirty = 1;   // (the work list's region is about to hold game indices)
                // boards still running below the final cap after the previous pass -> this pass's list
                if (!(pass == 1 && listed))
                hipLaunchKernelGGL(k_bounce_compact, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->d_status, b->d_plies,
                                   b->n, cap, list, counts + pass);
                launch_rollout(b, seed, pass_cap, group == 64 && from_initial, group, 0, list, counts + pass, queues + pass);
            }
            if (pass_cap >= cap) break;  // nothing can be left for a later pass
        }
        return;
    }
    // wider boards: one lane per board, two-pass enumeration
    if (flags & 1u)
        hipLaunchKernelGGL((k_bounce_play<false, true>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg,
                           b->d_planes, b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed,
                           b->first_game, cap, b->d_steps);
    else
        hipLaunchKernelGGL((k_bounce_play<false, false>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg,
                           b->d_planes, b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed,
                           b->first_game, cap, b->d_steps);
}

// ---- the opening book's life: one per (device, start position) in the process, counted by the batches that use it ----
namespace {
struct Book {
    int device = 0;
    BounceGeom key;          // (the start position and geometry it was built for)
    uint32_t* links = nullptr;
    uint32_t* hdr = nullptr;
    BookEntry* level[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int depth = 0;
    uint32_t n0 = 0;
    int users = 0;
};
std::mutex g_books_mu;
std::vector<Book*> g_books;

void free_book(Book* k) {
    (void)hipFree(k->links);
    (void)hipFree(k->hdr);
    for (int d = 1; d <= 4; ++d) (void)hipFree(k->level[d]);
    delete k;
}
}  // namespace

int bounce_book_acquire(bgs_batch* b, int max_depth) {
    b->book_links = nullptr;
    b->book_table = nullptr;
    b->book_depth = 0;
    b->book_owner = nullptr;
    if (max_depth <= 0 || b->bg.piece_count < 1 || b->bg.init_status != BGS_ST_RUNNING || b->bg.w > kMaxTrackedColumns) return 0;
    if (max_depth > 4) max_depth = 4;
    std::lock_guard<std::mutex> lock(g_books_mu);
    Book* k = nullptr;
    for (Book* c : g_books)
        if (c->device == b->device && memcmp(&c->key, &b->bg, sizeof(BounceGeom)) == 0 && c->depth <= max_depth) {
            k = c;
            break;
        }
    if (!k) {
        k = new (std::nothrow) Book();
        if (!k) return (int)hipErrorOutOfMemory;
        k->device = b->device;
        k->key = b->bg;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&k->links), sizeof(uint32_t) * kBookLinks);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&k->hdr), sizeof(uint32_t) * 16);
        const uint32_t cap[5] = {0, kBookCap1, kBookCap2, kBookCap3, kBookCap4};
        for (int d = 1; d <= 3 && e == hipSuccess; ++d) e = hipMalloc(reinterpret_cast<void**>(&k->level[d]), sizeof(BookEntry) * cap[d]);
        uint32_t hdr[8] = {0};
        if (e == hipSuccess) e = hipMemsetAsync(k->hdr, 0, sizeof(uint32_t) * 16, b->stream);
        if (e == hipSuccess) e = hipMemsetAsync(k->links, 0, sizeof(uint32_t) * kBookLinks, b->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_bounce_book_root, dim3(1), dim3(BGS_WAVE), 0, b->stream, b->bg, k->hdr);
            uint32_t* const link_at[4] = {nullptr, k->links, k->links + kBookCap1, k->links + kBookCap1 + kBookCap2};
            for (uint32_t d = 1; d <= 3; ++d) {
                hipLaunchKernelGGL(k_bounce_book_expand, dim3(cap[d] / BGS_WAVE), dim3(BGS_WAVE), 0, b->stream, b->bg, d,
                                   d > 1 ? k->level[d - 1] : nullptr, d > 1 ? link_at[d - 1] : nullptr, k->level[d], cap[d], k->hdr);
                hipLaunchKernelGGL(k_bounce_book_links, dim3(1), dim3(BGS_BLOCK), 0, b->stream, k->level[d], d, cap[d], link_at[d], k->hdr);
            }
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hdr, k->hdr, sizeof hdr, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) {
            free_book(k);
            return (int)e;
        }
        // the deepest level that fits: every level above it within its capacity, no position with more actions than a link holds
        int depth = 0;
        if (hdr[1] >= 1 && hdr[1] <= cap[1] && !(hdr[0] & 3u)) {
            depth = 1;
            if (hdr[2] <= cap[2] && !(hdr[0] & 4u)) {
                depth = 2;
                if (hdr[3] <= cap[3] && !(hdr[0] & 8u)) {
                    depth = 3;
                    if (hdr[4] >= 1 && hdr[4] <= cap[4]) depth = 4;
                }
            }
        }
        if (depth > max_depth) depth = max_depth;
        if (depth == 4) {
            e = hipMalloc(reinterpret_cast<void**>(&k->level[4]), sizeof(BookEntry) * hdr[4]);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_bounce_book_expand, dim3((hdr[4] + BGS_WAVE - 1) / BGS_WAVE), dim3(BGS_WAVE), 0, b->stream, b->bg,
                                   4u, k->level[3], k->links + kBookCap1 + kBookCap2, k->level[4], hdr[4], k->hdr);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
            }
            if (e != hipSuccess) depth = 3;   // (no memory for the last level: a shallower book)
        }
        k->depth = depth;
        k->n0 = hdr[1];
        g_books.push_back(k);
    }
    ++k->users;
    b->book_owner = k;
    if (k->depth > 0) {
        b->book_links = k->links;
        b->book_table = k->level[k->depth];
        b->book_depth = k->depth;
        b->book_n0 = k->n0;
    }
    return 0;
}

void bounce_book_release(bgs_batch* b) {
    if (!b->book_owner) return;
    std::lock_guard<std::mutex> lock(g_books_mu);
    Book* k = static_cast<Book*>(b->book_owner);
    b->book_owner = nullptr;
    b->book_links = nullptr;
    b->book_table = nullptr;
    b->book_depth = 0;
    if (--k->users == 0) {
        for (size_t i = 0; i < g_books.size(); ++i)
            if (g_books[i] == k) g_books.erase(g_books.begin() + (long)i);
        free_book(k);
    }
}

void bounce_step_actions(const bgs_batch* b, const int32_t* d_moves, int32_t* d_status_out) {
    hipLaunchKernelGGL(k_bounce_step_actions, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes,
                       b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_moves, d_status_out,
                       b->d_steps);
}

void bounce_transition(const bgs_batch* b, const int32_t* d_moves, int32_t* d_status_out, int8_t* d_grid, int8_t* d_player,
                       int8_t* d_winner, int32_t* d_plies, uint64_t* d_targets, int8_t* d_reward_out, uint32_t* d_done,
                       uint32_t ticket) {
    const bool wave_wanted = b->transition_wave != 0;
    if (b->n == 1 && wave_wanted && b->bg.h * b->bg.w <= 64) {   // the object API's engines: one board, one wave, a piece per lane
        hipLaunchKernelGGL((k_bounce_transition_wave<BGS_BOUNCE_MAX_PIECES>), dim3(1), dim3(BGS_WAVE), 0, b->stream, b->bg, b->d_planes,
                           b->d_status, b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), d_moves, d_status_out, b->d_steps, d_grid,
                           d_player, d_winner, d_plies, d_targets, reinterpret_cast<uint16_t*>(d_reward_out), d_done, ticket);
        return;
    }
    hipLaunchKernelGGL(k_bounce_transition, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes, b->d_status,
                       b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_moves, d_status_out, b->d_steps, d_grid,
                       d_player, d_winner, d_plies, d_targets, reinterpret_cast<uint16_t*>(d_reward_out),
                       grid_for(b->n) == 1 ? d_done : nullptr, ticket);
}

void bounce_unpack_grid(const bgs_batch* b, int8_t* d_grid) {
    const size_t lds = (size_t)BGS_BLOCK * b->bg.h * b->bg.w;
    hipLaunchKernelGGL(k_bounce_unpack, dim3(grid_for(b->n)), dim3(BGS_BLOCK), lds, b->stream, b->bg, b->d_planes, b->n,
                       d_grid);
}

void bounce_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies) {
    hipLaunchKernelGGL(k_bounce_meta, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->d_status, b->d_plies, b->n,
                       d_player, d_ended, d_winner, d_plies);
}

void bounce_targets(const bgs_batch* b, uint64_t* d_targets, int32_t* d_count) {
    hipLaunchKernelGGL(k_bounce_targets, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes,
                       b->d_status, b->d_plies, b->n, d_targets, d_count);
}

void bounce_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                 const int32_t* d_plies, int32_t* d_status_out) {
    hipLaunchKernelGGL(k_bounce_pack, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->bg, b->d_planes, b->d_status,
                       b->d_plies, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_grid, d_player, d_winner, d_plies,
                       d_status_out);
}

}  // namespace bgs
