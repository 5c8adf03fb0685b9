/*
 * bgs_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A deliberately naive, scalar, plain-C restatement of the game rules that the
 * reference (jojolebarjos/board-game-simulator-python) reaches through its nanobind
 * bindings, on the REFERENCE data layout (int8 grid[H][W], row 0 = bottom row).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (libbgs.so and the simulator package) never links, imports or calls it.
 *
 * Provenance / pinning.  The arithmetic the reference calls lives in the un-vendored
 * C++ header library github.com/jojolebarjos/board-game-simulator @ c8f8a075cc82ae91732627ca47640338736a40cb
 * (reference CMakeLists.txt:12-18), which is not in /root/reference and cannot be fetched.
 * The rules are therefore restated from the reference's binding call sites
 * (src/simulator/game/connect.cpp:24-61, src/simulator/game/bounce.cpp:24-60) and pinned
 * by EVERY fixture the reference's own tests hold for this path
 * (tests/test_connect.py:68-145, tests/test_bounce.py:92-410), transcribed as data into
 * tests/golden/reference_*.json by tests/golden/make_reference_fixtures.py and checked by
 * tests/test_oracle_golden.py.  What those fixtures do not constrain is listed as
 * "parity UNPINNED" in DESIGN.md (Connect vertical/diagonal wins, draws, action order;
 * Bounce action order/multiplicity, origin-cell vacancy).
 *
 * Conventions: every function returns 0 on success, a negative code on error.
 * winner codes: -1 = running / none, 0 / 1 = that player won, 2 = draw.
 */
#ifndef BGS_ORACLE_H
#define BGS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_ARG (-1)
#define ORC_ERR_ILLEGAL (-2)

/* ---- RNG contract (build-defined; the reference has no RNG, README.md:62 uses Python's) ---- */
/* philox4x32-10 (Salmon et al. 2011), key = (seed lo, seed hi); the sampled index is (draw * n_actions) >> 32.
 *   Bounce  (up to a few hundred actions a ply): one 32-bit word per ply -- counter = (game lo, game hi, ply >> 2, 0), the
 *           draw of a ply is output word (ply & 3);
 *   Connect (at most `width` <= 64 actions a ply; round 5): one 32-bit word per BLOCK of four plies -- counter = (game lo,
 *           game hi, ply >> 4, 0), the block's word is output word ((ply >> 2) & 3), and the draw of ply j = ply & 3 of the
 *           block is  word * A^j mod 2^32,  A = 747796405: the four draws of a block are four consecutive states of the
 *           multiplicative congruential generator x -> A x mod 2^32 started at the philox word.  Each of them is a bijection
 *           of the word, so every ply's index has exactly the distribution a word of its own would give it; what the four
 *           plies of a block share is 32 bits of entropy, and their JOINT distribution is the lattice of that generator:
 *           counted over all 2^32 words (tools/subdraw_lattice.c), every one of the 7^4 four-move sequences of a 7-column
 *           board comes within 4.2 x 10^-5 (relative) of 1 / 7^4, every pair of plies within 4 x 10^-7 of 1 / 49;
 *           13 columns: 2.9 x 10^-4, 16 columns: 4.3 x 10^-4.  Blocks are independent philox words.
 *   Connect, STRICT contract (ORC_RNG_PER_PLY; round 6, the form SURVEY.md 7.3 wrote down and the only one until round 5):
 *           a word per ply exactly as Bounce -- orc_draw / orc_sample_index -- i.e. one independent uniform choice per
 *           ply, what the reference's callers get from random.choice (README.md:62).  The library offers it as
 *           BGS_ROLLOUT_DRAW_PER_PLY / bgs_set_rng_contract. */
#define ORC_SUBDRAW_A 747796405u
#define ORC_RNG_PER_BLOCK 0   /* Connect: a word per block of four plies (the default) */
#define ORC_RNG_PER_PLY 1     /* Connect: a word per ply */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t orc_draw(uint64_t seed, uint64_t game, uint32_t ply);            /* Bounce: a word per ply */
uint32_t orc_sample_index(uint64_t seed, uint64_t game, uint32_t ply, uint32_t n_actions);
uint32_t orc_connect_draw(uint64_t seed, uint64_t game, uint32_t ply);    /* Connect: a word per four plies */
uint32_t orc_connect_sample_index(uint64_t seed, uint64_t game, uint32_t ply, uint32_t n_actions);

/* ---- Connect (reference surface: src/simulator/game/connect.cpp:24-54) ---- */
/* batch arrays: grid int8[n][h][w] (-1 empty, 0, 1), player int8[n], winner int8[n], plies int32[n] */
int orc_connect_reset(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies);
/* legal[n][w]: 1 where a stone may be dropped (all 0 once ended) -- State.actions (connect.cpp:43) */
int orc_connect_legal(int h, int w, int64_t n, const int8_t* grid, const int8_t* winner, uint8_t* legal);
/* apply column[i] to board i (State.action_at + Action.sample_next_state, connect.cpp:44,52);
 * column[i] < 0 leaves the board untouched; illegal column -> status[i] = ORC_ERR_ILLEGAL, board untouched */
int orc_connect_step_actions(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                             int32_t* plies, const int32_t* column, int32_t* status);
/* one uniformly sampled ply on every running board */
int orc_connect_step_random(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                            int32_t* plies, uint64_t seed, uint64_t first_game, uint64_t* steps);
/* play every running board to its end (or max_plies total plies) */
int orc_connect_rollout(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                        int32_t* plies, uint64_t seed, uint64_t first_game, int32_t max_plies, uint64_t* steps);
/* the same two under a named RNG contract (ORC_RNG_PER_BLOCK = the two above, ORC_RNG_PER_PLY = the strict one) */
int orc_connect_step_random_rng(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                                int32_t* plies, uint64_t seed, uint64_t first_game, int rng, uint64_t* steps);
int orc_connect_rollout_rng(int h, int w, int k, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                            int32_t* plies, uint64_t seed, uint64_t first_game, int32_t max_plies, int rng, uint64_t* steps);

/* ---- Bounce (reference surface: src/simulator/game/bounce.cpp:24-53) ---- */
/* grid int8[n][h][w]: 0 empty, v>0 a piece that moves exactly v steps; (x, y) coordinates, y=0 bottom */
int orc_bounce_validate(int h, int w, const int8_t* cfg_grid);
int orc_bounce_reset(int h, int w, const int8_t* cfg_grid, int64_t n, int8_t* grid, int8_t* player,
                     int8_t* winner, int32_t* plies);
/* targets[h*w] = 1 for every legal landing cell of the piece at (sx, sy) for `player` (State.actions_at,
 * bounce.cpp:41); all 0 when the cell holds no movable piece */
int orc_bounce_targets(int h, int w, const int8_t* grid, int player, int winner, int sx, int sy, uint8_t* targets);
/* canonical action list: sources by ascending x, targets by ascending (y, x); returns the count;
 * src_xy / dst_xy may be NULL, otherwise hold 2*cap ints */
int orc_bounce_actions(int h, int w, const int8_t* grid, int player, int winner, int cap, int32_t* src_xy,
                       int32_t* dst_xy);
int orc_bounce_count_actions(int h, int w, int64_t n, const int8_t* grid, const int8_t* player,
                             const int8_t* winner, int32_t* count);
/* move[i] = {sx, sy, tx, ty}; sx < 0 leaves board i untouched */
int orc_bounce_step_actions(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                            int32_t* plies, const int32_t* move, int32_t* status);
int orc_bounce_step_random(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner,
                           int32_t* plies, uint64_t seed, uint64_t first_game, uint64_t* steps);
int orc_bounce_rollout(int h, int w, int64_t n, int8_t* grid, int8_t* player, int8_t* winner, int32_t* plies,
                       uint64_t seed, uint64_t first_game, int32_t max_plies, uint64_t* steps);

/* reward[n][2] from winner codes (State.reward, connect.cpp:41 / bounce.cpp:38): +1/-1, draw or running 0/0 */
int orc_reward(int64_t n, const int8_t* winner, int8_t* reward);

#ifdef __cplusplus
}
#endif
#endif
