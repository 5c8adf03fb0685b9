/*
 * bgs_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See bgs_oracle.h for provenance.
 *
 * Deliberately naive: array grids in the reference layout, loop-based win scan, recursive move search.
 * It shares NO code with the HIP kernels (which work on bit-planes), so agreement between the two is an
 * independent check of the bit tricks.
 *
 * Rule sources (all relative to /root/reference):
 *   Connect  : tests/test_connect.py:24-25 (cell codes, row 0 = bottom), :75-115 (first player 0, drop into
 *              the lowest empty cell, alternation, horizontal win, reward [1,-1]), :131-138 (winner = -1
 *              while running); vertical/diagonal wins and the draw are the standard Connect-k rules
 *              [UNPINNED by any reference test].
 *   Bounce   : tests/test_bounce.py:92-362 -- 16 positions with exhaustive target sets, 16 transitions,
 *              6 terminal rewards; rules as reconstructed in SURVEY.md Appendix B.
 */
#include "bgs_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* RNG: philox4x32-10                                                                               */
/* ------------------------------------------------------------------------------------------------ */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t orc_draw(uint64_t seed, uint64_t game, uint32_t ply) {
    uint32_t ctr[4] = {(uint32_t)game, (uint32_t)(game >> 32), ply >> 2, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    return out[ply & 3u];
}

uint32_t orc_sample_index(uint64_t seed, uint64_t game, uint32_t ply, uint32_t n_actions) {
    return (uint32_t)(((uint64_t)orc_draw(seed, game, ply) * n_actions) >> 32);
}

/* Connect: the word of the ply's block, stepped (ply & 3) times through x -> A x mod 2^32 (bgs_oracle.h) */
uint32_t orc_connect_draw(uint64_t seed, uint64_t game, uint32_t ply) {
    uint32_t ctr[4] = {(uint32_t)game, (uint32_t)(game >> 32), ply >> 4, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    uint32_t x = out[(ply >> 2) & 3u];
    for (uint32_t j = 0; j < (ply & 3u); ++j) x *= ORC_SUBDRAW_A;
    return x;
}

uint32_t orc_connect_sample_index(uint64_t seed, uint64_t game, uint32_t ply, uint32_t n_actions) {
    return (uint32_t)(((uint64_t)orc_connect_draw(seed, game, ply) * n_actions) >> 32);
}

int orc_reward(int64_t n, const int8_t* winner, int8_t* reward) {
    if (n < 0 || !winner || !reward) return ORC_ERR_ARG;
    for (int64_t i = 0; i < n; ++i) {
        int8_t r0 = 0, r1 = 0;
        if (winner[i] == 0) { r0 = 1; r1 = -1; }
        else if (winner[i] == 1) { r0 = -1; r1 = 1; }
        reward[2 * i] = r0;
        reward[2 * i + 1] = r1;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Connect                                                                                          */
/* ------------------------------------------------------------------------------------------------ */

static int connect_cfg_ok(int h, int w, int k) { return h >= 1 && w >= 1 && k >= 1 && h <= 64 && w <= 64; }

/* number of stones of `who` in a row through (x, y) along (dx, dy), counting (x, y) itself.
 * Follows: reference tests/test_connect.py:107-115 (horizontal pair wins at count = 2); other directions are the
 * standard Connect-k rule [UNPINNED]. */
static int connect_run(int h, int w, const int8_t* g, int x, int y, int dx, int dy, int who) {
    int count = 1;
    for (int s = 1;; ++s) {
        int xx = x + s * dx, yy = y + s * dy;
        if (xx < 0 || xx >= w || yy < 0 || yy >= h || g[yy * w + xx] != who) break;
        ++count;
    }
    for (int s = 1;; ++s) {
        int xx = x - s * dx, yy = y - s * dy;
        if (xx < 0 || xx >= w || yy < 0 || yy >= h || g[yy * w + xx] != who) break;
        ++count;
    }
    return count;
}

/* State.actions (reference src/simulator/game/connect.cpp:43): one action per column whose top cell is empty, ascending
 * [order UNPINNED], none once the game has ended (by analogy with tests/test_bounce.py:151). */
static int connect_legal_one(int h, int w, const int8_t* g, int winner, int* cols) {
    int n = 0;
    if (winner != -1) return 0;
    for (int x = 0; x < w; ++x)
        if (g[(h - 1) * w + x] == -1) cols[n++] = x;
    return n;
}

/* State.action_at + Action.sample_next_state (reference connect.cpp:44,52): the stone lands in the lowest empty cell of
 * the column, the other player is to move (tests/test_connect.py:86-104, :130-138), a k-run ends the game with reward
 * [1,-1] for the mover (:107-115); a full board without a run is a draw [UNPINNED].  Illegal: full column, column out
 * of range, ended board (RuntimeError in the reference, textual/connect.py:115-118). */
static int connect_apply_one(int h, int w, int k, int8_t* g, int8_t* player, int8_t* winner, int32_t* plies, int col) {
    if (*winner != -1 || col < 0 || col >= w) return ORC_ERR_ILLEGAL;
    int y = 0;
    while (y < h && g[y * w + col] != -1) ++y;
    if (y == h) return ORC_ERR_ILLEGAL;
    int who = *player;
    g[y * w + col] = (int8_t)who;
    *player = (int8_t)(1 - who);
    *plies += 1;
    static const int dirs[4][2] = {{1, 0}, {0, 1}, {1, 1}, {1, -1}};
    for (int d = 0; d < 4; ++d)
        if (connect_run(h, w, g, col, y, dirs[d][0], dirs[d][1], who) >= k) { *winner = (int8_t)who; return ORC_OK; }
    int full = 1;
    for (int x = 0; x < w; ++x)
        if (g[(h - 1) * w + x] == -1) full = 0;
    if (full) *winner = 2;
    return ORC_OK;
}

/* Config.sample_initial_state (reference connect.cpp:32): empty board (-1), player 0 to move, winner -1
 * (tests/test_connect.py:75-83, :131-138). */
int orc_connect_reset(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies) {
    if (!connect_cfg_ok(h, w, 1) || n < 0) return ORC_ERR_ARG;
    /* board by board, with the thread team and the static schedule of the rollout below: the first reset of a batch is what
     * touches its pages first, so every thread's share of the boards sits in memory next to the core that will play it
     * (one memset from one thread put all 2^20 boards of bench.py's baseline on one NUMA node of the host) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        memset(grid + i * h * w, 0xFF, (size_t)h * w);
        player[i] = 0;
        winner[i] = -1;
        plies[i] = 0;
    }
    return ORC_OK;
}

int orc_connect_legal(int h, int w, int64_t n, const int8_t* grid, const int8_t* winner, uint8_t* legal) {
    if (!connect_cfg_ok(h, w, 1) || n < 0) return ORC_ERR_ARG;
    for (int64_t i = 0; i < n; ++i)
        for (int x = 0; x < w; ++x)
            legal[i * w + x] = (uint8_t)(winner[i] == -1 && grid[(i * h + (h - 1)) * w + x] == -1);
    return ORC_OK;
}

int orc_connect_step_actions(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                             int32_t* plies, const int32_t* column, int32_t* status) {
    if (!connect_cfg_ok(h, w, k) || n < 0) return ORC_ERR_ARG;
    for (int64_t i = 0; i < n; ++i) {
        int st = ORC_OK;
        if (column[i] >= 0) st = connect_apply_one(h, w, k, grid + i * h * w, player + i, winner + i, plies + i, column[i]);
        if (status) status[i] = st;
    }
    return ORC_OK;
}

/* the caller loop of reference README.md:52-69 with random.choice replaced by the RNG contract of bgs_oracle.h */
static uint64_t connect_play(int h, int w, int k, int8_t* g, int8_t* player, int8_t* winner, int32_t* plies,
                             uint64_t seed, uint64_t game, int32_t max_plies, int single_ply, int per_ply) {
    int cols[64];
    uint64_t steps = 0;
    while (*winner == -1 && *plies < max_plies) {
        int n = connect_legal_one(h, w, g, *winner, cols);
        /* per_ply: the strict contract (ORC_RNG_PER_PLY) -- a philox word of its own for every ply, as Bounce draws */
        uint32_t idx = per_ply ? orc_sample_index(seed, game, (uint32_t)*plies, (uint32_t)n)
                               : orc_connect_sample_index(seed, game, (uint32_t)*plies, (uint32_t)n);
        connect_apply_one(h, w, k, g, player, winner, plies, cols[idx]);
        ++steps;
        if (single_ply) break;
    }
    return steps;
}

int orc_connect_step_random_rng(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                                int32_t* plies, uint64_t seed, uint64_t first_game, int rng, uint64_t* steps) {
    if (!connect_cfg_ok(h, w, k) || n < 0 || (rng != ORC_RNG_PER_BLOCK && rng != ORC_RNG_PER_PLY)) return ORC_ERR_ARG;
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        total += connect_play(h, w, k, grid + i * h * w, player + i, winner + i, plies + i, seed, first_game + (uint64_t)i,
                              INT32_MAX, 1, rng == ORC_RNG_PER_PLY);
    if (steps) *steps = total;
    return ORC_OK;
}

int orc_connect_rollout_rng(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                            int32_t* plies, uint64_t seed, uint64_t first_game, int32_t max_plies, int rng, uint64_t* steps) {
    if (!connect_cfg_ok(h, w, k) || n < 0 || (rng != ORC_RNG_PER_BLOCK && rng != ORC_RNG_PER_PLY)) return ORC_ERR_ARG;
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        total += connect_play(h, w, k, grid + i * h * w, player + i, winner + i, plies + i, seed, first_game + (uint64_t)i,
                              max_plies, 0, rng == ORC_RNG_PER_PLY);
    if (steps) *steps = total;
    return ORC_OK;
}

int orc_connect_step_random(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                            int32_t* plies, uint64_t seed, uint64_t first_game, uint64_t* steps) {
    return orc_connect_step_random_rng(h, w, k, n, grid, player, winner, plies, seed, first_game, ORC_RNG_PER_BLOCK, steps);
}

int orc_connect_rollout(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                        int32_t* plies, uint64_t seed, uint64_t first_game, int32_t max_plies, uint64_t* steps) {
    return orc_connect_rollout_rng(h, w, k, n, grid, player, winner, plies, seed, first_game, max_plies, ORC_RNG_PER_BLOCK, steps);
}

/* ------------------------------------------------------------------------------------------------ */
/* Bounce                                                                                           */
/* ------------------------------------------------------------------------------------------------ */

enum { DIR_FWD = 0, DIR_LEFT = 1, DIR_RIGHT = 2 };

typedef struct {
    int h, w, fwd;          /* fwd = +1 for player 0 (towards row h-1), -1 for player 1 (towards row 0) */
    int goal_row;           /* the row the mover tries to reach */
    const int8_t* g;
    uint8_t* targets;       /* h*w flags */
    uint8_t* visited;       /* [cell][remaining][lastdir] */
    int maxv;
} bounce_search;

static int bounce_cfg_ok(int h, int w) { return h >= 3 && w >= 1 && h <= 64 && w <= 64; }

/* Config(grid) (reference bounce.cpp:26): pieces only strictly between the first and last row (tests/test_bounce.py:30) */
int orc_bounce_validate(int h, int w, const int8_t* cfg) {
    if (!bounce_cfg_ok(h, w) || !cfg) return ORC_ERR_ARG;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = cfg[y * w + x];
            if (v < 0) return ORC_ERR_ARG;
            if (v > 0 && (y == 0 || y == h - 1)) return ORC_ERR_ARG; /* goal rows hold no pieces (test_bounce.py:30) */
        }
    return ORC_OK;
}

/* SURVEY Appendix B rules 4-5: a segment of `remaining` unit steps, each forward / left / right, never backward,
 * no immediate left<->right reversal; intermediate cells empty and not in a goal row; the last step may land on an
 * empty cell (target), in the mover's goal row (target), or on a piece, which starts a fresh segment of that
 * piece's value.  The board is searched as it stands: the moving piece still occupies its origin
 * [UNPINNED by the reference tests; build-defined]. */
static void bounce_walk(bounce_search* s, int x, int y, int remaining, int lastdir) {
    uint8_t* seen = &s->visited[((y * s->w + x) * (s->maxv + 1) + remaining) * 3 + lastdir];
    if (*seen) return;
    *seen = 1;
    for (int dir = 0; dir < 3; ++dir) {
        if ((dir == DIR_LEFT && lastdir == DIR_RIGHT) || (dir == DIR_RIGHT && lastdir == DIR_LEFT)) continue;
        int nx = x + (dir == DIR_RIGHT) - (dir == DIR_LEFT);
        int ny = y + (dir == DIR_FWD ? s->fwd : 0);
        if (nx < 0 || nx >= s->w || ny < 0 || ny >= s->h) continue;
        int cell = ny * s->w + nx;
        int in_goal = (ny == 0 || ny == s->h - 1);
        if (remaining == 1) {
            if (in_goal) { if (ny == s->goal_row) s->targets[cell] = 1; }
            else if (s->g[cell] == 0) s->targets[cell] = 1;
            else bounce_walk(s, nx, ny, s->g[cell], DIR_FWD);
        } else {
            if (in_goal || s->g[cell] != 0) continue;
            bounce_walk(s, nx, ny, remaining - 1, dir);
        }
    }
}

/* active row (Appendix B rule 3; reference tests/test_bounce.py:44-48 infers the player from it and every scripted
 * selection obeys it): nearest non-empty non-goal row on the mover's side, or -1 */
static int bounce_active_row(int h, int w, const int8_t* g, int player) {
    if (player == 0) {
        for (int y = 1; y < h - 1; ++y)
            for (int x = 0; x < w; ++x)
                if (g[y * w + x] > 0) return y;
    } else {
        for (int y = h - 2; y >= 1; --y)
            for (int x = 0; x < w; ++x)
                if (g[y * w + x] > 0) return y;
    }
    return -1;
}

/* State.actions_at(source) (reference bounce.cpp:41): the exhaustive target set the reference tests compare with
 * (tests/test_bounce.py:80-83) */
static int bounce_targets_one(int h, int w, const int8_t* g, int player, int sx, int sy, uint8_t* targets) {
    memset(targets, 0, (size_t)h * w);
    if (sx < 0 || sx >= w || sy < 0 || sy >= h) return ORC_ERR_ARG;
    if (g[sy * w + sx] <= 0 || sy != bounce_active_row(h, w, g, player)) return ORC_OK;
    int maxv = 0;
    for (int c = 0; c < h * w; ++c)
        if (g[c] > maxv) maxv = g[c];
    bounce_search s;
    s.h = h; s.w = w; s.g = g; s.targets = targets; s.maxv = maxv;
    s.fwd = player == 0 ? 1 : -1;
    s.goal_row = player == 0 ? h - 1 : 0;
    s.visited = (uint8_t*)calloc((size_t)h * w * (maxv + 1) * 3, 1);
    if (!s.visited) return ORC_ERR_ARG;
    bounce_walk(&s, sx, sy, g[sy * w + sx], DIR_FWD);
    free(s.visited);
    return ORC_OK;
}

int orc_bounce_targets(int h, int w, const int8_t* grid, int player, int winner, int sx, int sy, uint8_t* targets) {
    if (!bounce_cfg_ok(h, w) || !grid || !targets) return ORC_ERR_ARG;
    if (winner != -1) { memset(targets, 0, (size_t)h * w); return ORC_OK; }
    return bounce_targets_one(h, w, grid, player, sx, sy, targets);
}

/* State.actions (reference bounce.cpp:40): all (source, target) pairs; canonical order and no duplicates are
 * build-defined [UNPINNED]; empty once ended (tests/test_bounce.py:151,277,298,319,340,361) */
int orc_bounce_actions(int h, int w, const int8_t* g, int player, int winner, int cap, int32_t* src_xy, int32_t* dst_xy) {
    if (!bounce_cfg_ok(h, w) || !g) return ORC_ERR_ARG;
    if (winner != -1) return 0;
    int row = bounce_active_row(h, w, g, player);
    if (row < 0) return 0;
    uint8_t* targets = (uint8_t*)malloc((size_t)h * w);
    int count = 0;
    for (int x = 0; x < w; ++x) {
        if (g[row * w + x] <= 0) continue;
        bounce_targets_one(h, w, g, player, x, row, targets);
        for (int c = 0; c < h * w; ++c) {
            if (!targets[c]) continue;
            if (count < cap) {
                if (src_xy) { src_xy[2 * count] = x; src_xy[2 * count + 1] = row; }
                if (dst_xy) { dst_xy[2 * count] = c % w; dst_xy[2 * count + 1] = c / w; }
            }
            ++count;
        }
    }
    free(targets);
    return count;
}

int orc_bounce_count_actions(int h, int w, int64_t n, const int8_t* grid, const int8_t* player,
                             const int8_t* winner, int32_t* count) {
    if (!bounce_cfg_ok(h, w) || n < 0) return ORC_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        count[i] = orc_bounce_actions(h, w, grid + i * h * w, player[i], winner[i], 0, NULL, NULL);
    return ORC_OK;
}

/* Appendix B rule 7: goal row reached -> mover wins; else if the next player cannot move: draw when the mover
 * could not move either, otherwise the mover wins. */
static void bounce_settle(int h, int w, const int8_t* g, int mover, int ty, int8_t* winner) {
    if (ty == 0 || ty == h - 1) { *winner = (int8_t)mover; return; }
    if (orc_bounce_actions(h, w, g, 1 - mover, -1, 0, NULL, NULL) > 0) return;
    *winner = (int8_t)(orc_bounce_actions(h, w, g, mover, -1, 0, NULL, NULL) > 0 ? mover : 2);
}

/* State.action_at(source, target) + Action.sample_next_state (reference bounce.cpp:42,51): the piece moves from source
 * to target, the other player is to move (tests/test_bounce.py:106-148) */
static int bounce_apply_one(int h, int w, int8_t* g, int8_t* player, int8_t* winner, int32_t* plies, int sx, int sy,
                            int tx, int ty) {
    if (*winner != -1) return ORC_ERR_ILLEGAL;
    if (sx < 0 || sx >= w || sy < 0 || sy >= h || tx < 0 || tx >= w || ty < 0 || ty >= h) return ORC_ERR_ILLEGAL;
    uint8_t* targets = (uint8_t*)malloc((size_t)h * w);
    bounce_targets_one(h, w, g, *player, sx, sy, targets);
    int ok = targets[ty * w + tx];
    free(targets);
    if (!ok) return ORC_ERR_ILLEGAL;
    int mover = *player;
    g[ty * w + tx] = g[sy * w + sx];
    g[sy * w + sx] = 0;
    *player = (int8_t)(1 - mover);
    *plies += 1;
    bounce_settle(h, w, g, mover, ty, winner);
    return ORC_OK;
}

/* Config.sample_initial_state (reference bounce.cpp:29): a copy of the config grid, player 0 to move
 * (tests/test_bounce.py:53-60, :392-403) */
int orc_bounce_reset(int h, int w, const int8_t* cfg, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                     int32_t* plies) {
    if (orc_bounce_validate(h, w, cfg) != ORC_OK || n < 0) return ORC_ERR_ARG;
    /* a start position without any legal move is already over (nobody moved: treat player 1 as "the mover") */
    int8_t w0 = -1;
    if (orc_bounce_actions(h, w, cfg, 0, -1, 0, NULL, NULL) == 0)
        w0 = (int8_t)(orc_bounce_actions(h, w, cfg, 1, -1, 0, NULL, NULL) > 0 ? 1 : 2);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {   /* (first touch by the thread that will play the board: see orc_connect_reset) */
        memcpy(grid + i * h * w, cfg, (size_t)h * w);
        player[i] = 0;
        winner[i] = w0;
        plies[i] = 0;
    }
    return ORC_OK;
}

int orc_bounce_step_actions(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies,
                            const int32_t* move, int32_t* status) {
    if (!bounce_cfg_ok(h, w) || n < 0) return ORC_ERR_ARG;
    for (int64_t i = 0; i < n; ++i) {
        int st = ORC_OK;
        const int32_t* m = move + 4 * i;
        if (m[0] >= 0) st = bounce_apply_one(h, w, grid + i * h * w, player + i, winner + i, plies + i, m[0], m[1], m[2], m[3]);
        if (status) status[i] = st;
    }
    return ORC_OK;
}

static uint64_t bounce_play(int h, int w, int8_t* g, int8_t* player, int8_t* winner, int32_t* plies, uint64_t seed,
                            uint64_t game, int32_t max_plies, int single_ply) {
    int cap = 0;
    int32_t *src = NULL, *dst = NULL;
    uint64_t steps = 0;
    while (*winner == -1 && *plies < max_plies) {
        int n = orc_bounce_actions(h, w, g, *player, -1, cap, src, dst);
        if (n > cap) {
            cap = n + 64;
            src = (int32_t*)realloc(src, sizeof(int32_t) * 2 * cap);
            dst = (int32_t*)realloc(dst, sizeof(int32_t) * 2 * cap);
            n = orc_bounce_actions(h, w, g, *player, -1, cap, src, dst);
        }
        if (n == 0) break; /* cannot happen for a running board: bounce_settle ends blocked games */
        uint32_t idx = orc_sample_index(seed, game, (uint32_t)*plies, (uint32_t)n);
        bounce_apply_one(h, w, g, player, winner, plies, src[2 * idx], src[2 * idx + 1], dst[2 * idx], dst[2 * idx + 1]);
        ++steps;
        if (single_ply) break;
    }
    free(src);
    free(dst);
    return steps;
}

int orc_bounce_step_random(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies,
                           uint64_t seed, uint64_t first_game, uint64_t* steps) {
    if (!bounce_cfg_ok(h, w) || n < 0) return ORC_ERR_ARG;
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i)
        total += bounce_play(h, w, grid + i * h * w, player + i, winner + i, plies + i, seed, first_game + (uint64_t)i,
                             INT32_MAX, 1);
    if (steps) *steps = total;
    return ORC_OK;
}

int orc_bounce_rollout(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies,
                       uint64_t seed, uint64_t first_game, int32_t max_plies, uint64_t* steps) {
    if (!bounce_cfg_ok(h, w) || n < 0) return ORC_ERR_ARG;
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i)
        total += bounce_play(h, w, grid + i * h * w, player + i, winner + i, plies + i, seed, first_game + (uint64_t)i,
                             max_plies, 0);
    if (steps) *steps = total;
    return ORC_OK;
}
