/*
 * bgs.h -- C ABI of libbgs.so, the MI355X (gfx950) batched board-game rollout engine.
 *
 * This is the drop-in boundary for the ONE hot path of jojolebarjos/board-game-simulator-python:
 *   legal-move enumeration -> uniform action sampling -> sample_next_state transition -> terminal + reward,
 * for N independent boards per launch.  Every entry point names the reference binding it replaces
 * (paths relative to the reference tree, src/simulator/game/...).  The reference itself has no batched API,
 * no RNG and no device; "batch", "seed" and "device" are this library's concepts.
 *
 * Conventions
 *   - plain C types only; no C++ or torch types cross this boundary;
 *   - every function returns BGS_OK (0) or a negative bgs_status; bgs_last_error() gives a thread-local
 *     message.  The reference raises C++ exceptions that nanobind turns into RuntimeError
 *     (textual/connect.py:115-118, textual/bounce.py:119-128): the Python shim maps BGS_ERR_ILLEGAL and
 *     BGS_ERR_RUNTIME to RuntimeError and BGS_ERR_ARG to TypeError/ValueError;
 *   - host arrays are C-contiguous, reference layout: grid int8[n][height][width], row 0 = bottom row
 *     (tensor.hpp:29-34; tests/test_connect.py:24-25); the caller owns every buffer it passes
 *     (the reference copies both ways too: tensor.hpp:63,80-84);
 *   - a batch lives on ONE device; all work is enqueued on the batch's HIP stream (default: the null
 *     stream); functions that fill host memory synchronise that stream before returning;
 *   - winner codes: -1 running, 0 / 1 that player won, 2 draw.  reward = +1 / -1 per player, 0 / 0 otherwise.
 *
 * RNG contract (build-defined: the reference has no RNG, its callers use random.choice, README.md:62).  A draw is a
 * 32-bit value, the sampled action is index (draw * n_actions) >> 32 of the canonical action list (Connect: legal columns
 * ascending; Bounce: sources by ascending x, targets by ascending (y, x)); philox = philox4x32-10 with key = seed.
 *   Bounce : draw(seed, game, ply) = philox(counter = (game lo, game hi, ply >> 2, 0))[ply & 3] -- a word per ply.
 *   Connect: word(seed, game, ply) = philox(counter = (game lo, game hi, ply >> 4, 0))[(ply >> 2) & 3] -- a word per block
 *            of four plies -- and draw = word * A^(ply & 3) mod 2^32, A = 747796405: the four draws of a block are four
 *            consecutive states of the multiplicative congruential generator x -> A x mod 2^32 started at the word.  Every
 *            one of them is a bijection of the word, so each ply's index is distributed exactly as a word of its own would
 *            make it (bias <= n / 2^32); the four plies of a block share 32 bits of entropy, and counted over ALL 2^32 words
 *            every four-move sequence of a 7-column board comes within 4.2 x 10^-5 (relative) of 1 / 7^4 and every pair of
 *            plies within 4 x 10^-7 of 1 / 49 (13 columns: 2.9 x 10^-4, 16 columns: 4.3 x 10^-4; tools/subdraw_lattice.c).
 *            Different blocks use different philox words.  (Round 5; until then Connect drew a word per ply too.  One philox
 *            call now serves sixteen plies, and the bench kernel's ply loop holds none: DESIGN.md section 3.)
 *   Connect, strict contract (round 6; bgs_set_rng_contract(b, BGS_RNG_PER_PLY) or BGS_ROLLOUT_DRAW_PER_PLY): exactly
 *            Bounce's rule -- a philox word per ply.  The word-per-block contract stays the default; both are pinned by the
 *            oracle (oracle/bgs_oracle.h: ORC_RNG_PER_BLOCK / ORC_RNG_PER_PLY) and by the -m gpu parity tests.
 * `game` = first_game + index in batch, so results do not depend on sharding, launch geometry or kernel family.
 */
#ifndef BGS_H
#define BGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libbgs.so is built with -fvisibility=hidden and a linker version script (csrc/bgs.map): the functions this header
 * declares are the whole dynamic symbol table of the library (tests/test_abi_and_host.py compares `nm -D` with it). */
#if defined(__GNUC__) || defined(__clang__)
#define BGS_API __attribute__((visibility("default")))
#else
#define BGS_API
#endif

typedef struct bgs_batch bgs_batch; /* opaque: N boards of one game configuration on one device */

typedef enum bgs_status {
    BGS_OK = 0,
    BGS_ERR_ARG = -1,         /* bad argument / unsupported geometry */
    BGS_ERR_ILLEGAL = -2,     /* illegal move (per-board code in status arrays) */
    BGS_ERR_RUNTIME = -3,     /* HIP runtime error */
    BGS_ERR_NO_DEVICE = -4    /* no usable GPU: the product path has no CPU fallback */
} bgs_status;

typedef enum bgs_buffer_id {
    BGS_BUF_PLANES = 0,  /* uint64 [planes][n]  bit-packed boards, one plane contiguous over the batch */
    BGS_BUF_STATUS = 1,  /* uint8  [n]          0 running, 1 / 2 player 0 / 1 won, 3 draw */
    BGS_BUF_PLIES = 2,   /* uint16 [n]          plies played (Bounce only; Connect derives it from the planes) */
    BGS_BUF_REWARD = 3,  /* int8   [n][2]       reward per player, valid after a board ended (else 0) */
    BGS_BUF_STEPS = 4,   /* uint64 [2048]       sharded env-step counter: the SUM of all words is the count */
    BGS_BUF_STAGING = 5  /* scratch the read/write entry points unpack through */
} bgs_buffer_id;

/* rollout flags */
#define BGS_ROLLOUT_DEFAULT 0u
#define BGS_ROLLOUT_FROM_INITIAL 1u /* ignore the stored boards: every game starts from Config.sample_initial_state() */
#define BGS_ROLLOUT_DRAW_PER_PLY 4u /* Connect: this call draws under the strict RNG contract (BGS_RNG_PER_PLY below) */

/* RNG contracts of a Connect batch (bgs_set_rng_contract; Bounce draws a word per ply under either) */
#define BGS_RNG_PER_BLOCK 0 /* default: a philox word per block of four plies, the plies' draws its sub-draws (see above) */
#define BGS_RNG_PER_PLY 1   /* strict: a philox word per ply -- draw(seed, game, ply) = philox(counter = (game lo, game hi,
                             * ply >> 2, 0))[ply & 3], exactly Bounce's: one independent uniform choice per ply, what a
                             * caller of the reference gets from random.choice (README.md:62).  Costs the rollout kernels a
                             * philox call per four plies instead of per sixteen (bench.py --rng per-ply prints both rates). */

/* ---- library ------------------------------------------------------------------------------------ */
BGS_API int bgs_version(void);
BGS_API const char* bgs_last_error(void);
BGS_API int bgs_device_count(int* count);
/* identity of the kernels this library was LINKED with (16 hex digits): every kernel translation unit embeds the hash of
 * its own source, the kernel headers and the compile flags when it is compiled, and this folds the three.  Measurement
 * files under profiles/ carry the id of the build they were taken on, and bench.py refuses to quote instruction counts
 * of another build.  `make -C csrc print-id` gives the id the sources in the tree would produce. */
BGS_API const char* bgs_build_id(void);
/* the id one kernel unit was compiled with: 0 connect_kernels, 1 bounce_kernels, 2 generic_kernels; NULL otherwise */
BGS_API const char* bgs_kernel_unit_id(int unit);

/* ---- configuration + batch lifetime ------------------------------------------------------------- */
/* replaces connect::Config(height, width, count) + Config::sample_initial_state (connect.cpp:26,32), N at a time.
 * arena: optional caller-owned device memory of at least bgs_connect_arena_bytes() bytes (256-byte aligned),
 * e.g. a torch uint8 tensor; NULL lets the library hipMalloc its own. */
BGS_API int bgs_connect_arena_bytes(int height, int width, int count, int64_t n, size_t* bytes);
BGS_API int bgs_connect_create(int height, int width, int count, int64_t n, int device, void* arena, size_t arena_bytes,
                       bgs_batch** out);
/* replaces bounce::Config(grid) + Config::sample_initial_state (bounce.cpp:26,29); cfg_grid int8[height][width] host.
 * Batches of 32768 boards and more (bit-packed boards, at most 16 pieces) also get the OPENING BOOK of their start position:
 * every path of up to four plies from it, enumerated once per start position and device by the library's own move search
 * and shared by the batches that have that start position -- the fused rollout's lanes start four plies in, with the
 * game's own draws (results are those of searching every ply, bit for bit).  Device memory OUTSIDE the arena: 21.7 MB for
 * the default 9x6 board, freed with the last batch that uses it.  BGS_BOUNCE_BOOK=0 switches it off. */
BGS_API int bgs_bounce_arena_bytes(int height, int width, int64_t n, size_t* bytes);
BGS_API int bgs_bounce_create(const int8_t* cfg_grid, int height, int width, int64_t n, int device, void* arena,
                      size_t arena_bytes, bgs_batch** out);
BGS_API int bgs_destroy(bgs_batch* b);

/* hipStream_t; NULL = null stream.  Work already enqueued for the batch on its previous stream is ordered before
 * anything enqueued on the new one (event + stream wait), so a batch may be created under one stream and used on
 * another without a host synchronisation. */
BGS_API int bgs_set_stream(bgs_batch* b, void* hip_stream);
/* a HIP stream of the library's own (non-blocking), for hosts without torch: batches that should overlap -- one per
 * host thread, say -- each get one.  Destroy it after the batches bound to it. */
BGS_API int bgs_stream_create(int device, void** hip_stream);
BGS_API int bgs_stream_destroy(int device, void* hip_stream);
BGS_API int bgs_set_first_game(bgs_batch* b, uint64_t first_game); /* global id of board 0 (sharding across GPUs) */
/* The RNG contract every later random step / rollout of this (Connect) batch draws under: BGS_RNG_PER_BLOCK (default) or
 * BGS_RNG_PER_PLY.  Build-defined like the RNG itself (the reference has none: its callers use random.choice,
 * README.md:62); a single rollout call can ask for the strict contract with BGS_ROLLOUT_DRAW_PER_PLY instead. */
BGS_API int bgs_set_rng_contract(bgs_batch* b, int contract);
/* A hint, not a rule of the game: how many rollout launches the caller keeps in flight on this batch's device (its own
 * included; 1 = one launch at a time, the default).  Results never depend on it.  The Bounce rollout shapes its launch
 * by it -- alone on the chip: a short bulk pass on many waves (shortest time to the last reward); among 16: few
 * long-lived waves (fewest instructions per ply).  bgs_pipeline_create passes its depth to its batches.  The reference
 * has no counterpart (one board per call: bounce.cpp:51). */
BGS_API int bgs_set_launches_in_flight(bgs_batch* b, int32_t launches);
BGS_API int bgs_synchronize(bgs_batch* b);
BGS_API int bgs_info(const bgs_batch* b, int* game, int* height, int* width, int* count, int64_t* n, int* planes);
/* Geometries beyond the bit-packed kernels' limits (Connect: height > 15, width > 16 or width * (height + 1) > 192;
 * Bounce: more than 64 cells or piece values above 15) are served by the generic kernels: same entry points, same
 * results, the board held as int8[n][h][w] (BGS_BUF_PLANES is then that grid).  Limits of the generic path: Connect
 * height, width <= 64; Bounce height, width <= 64, height * width <= 1024, values <= 127.
 * bgs_legal_bytes: bytes per board of the legal-move record of bgs_transition, and whether the batch is generic:
 *   Connect           uint8[width] mask;
 *   Bounce (packed)   uint64[width + 1]: target masks per column of the active row, then the active row's y;
 *   Bounce (generic)  int32 active row (-1 = none), then uint8 flags[width][height * width] (1 = legal target cell
 *                     of the piece in that column of the active row), padded to a multiple of 8 bytes. */
BGS_API int bgs_legal_bytes(const bgs_batch* b, size_t* bytes, int* generic);
/* device pointer + size of one of the batch's buffers (zero-copy hand-over to torch / RCCL) */
BGS_API int bgs_buffer(const bgs_batch* b, int buffer_id, void** device_ptr, size_t* bytes);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* all boards back to Config::sample_initial_state() (connect.cpp:32, bounce.cpp:29); zeroes the step counter */
BGS_API int bgs_reset(bgs_batch* b);
/* ONE ply on every running board: State::get_actions (connect.cpp:43, bounce.cpp:40) -> uniform choice
 * (README.md:62 random.choice) -> Action::sample_next_state (connect.cpp:52, bounce.cpp:51) -> has_ended / reward */
BGS_API int bgs_step_random(bgs_batch* b, uint64_t seed);
/* `plies` such plies on every board that is (still) running, boards held in registers in between where the kernel
 * allows it (Connect boards of one 64-bit word: the per-ply memory traffic divides by `plies`); the result is the one
 * of `plies` calls of bgs_step_random */
BGS_API int bgs_step_random_n(bgs_batch* b, uint64_t seed, int32_t plies);
/* ONE caller-chosen ply: State::get_action_at (connect.cpp:44 / bounce.cpp:42) + Action::sample_next_state.
 * Connect: actions int32[n] = column; Bounce: int32[n][4] = source x, y, target x, y.  A negative first entry
 * skips the board.  actions_on_device != 0: `actions` is a device pointer.  status (host int32[n], may be
 * NULL): BGS_OK or BGS_ERR_ILLEGAL per board; illegal moves leave the board untouched. */
BGS_API int bgs_step_actions(bgs_batch* b, const int32_t* actions, int actions_on_device, int32_t* status);
/* plies until every board ended or holds max_plies plies (README.md:52 `while not state.has_ended`), fused in
 * one launch with the board in registers */
BGS_API int bgs_rollout(bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags);
/* env-steps (transitions applied to running boards) since the last bgs_reset / bgs_reset_steps */
BGS_API int bgs_steps(bgs_batch* b, uint64_t* steps);
BGS_API int bgs_reset_steps(bgs_batch* b);

/* ---- observation: packed state -> reference layout, into HOST buffers ----------------------------- */
BGS_API int bgs_read_grid(bgs_batch* b, int8_t* grid);      /* State::get_grid   (connect.cpp:42, bounce.cpp:39) int8[n][h][w] */
BGS_API int bgs_read_player(bgs_batch* b, int8_t* player);  /* State::get_player (connect.cpp:40, bounce.cpp:37) int8[n] */
BGS_API int bgs_read_ended(bgs_batch* b, uint8_t* ended);   /* State::has_ended  (connect.cpp:39, bounce.cpp:36) uint8[n] */
BGS_API int bgs_read_winner(bgs_batch* b, int8_t* winner);  /* JSON key "winner" (tests/test_connect.py:137) int8[n] */
BGS_API int bgs_read_reward(bgs_batch* b, int8_t* reward);  /* State::get_reward (connect.cpp:41, bounce.cpp:38) int8[n][2] */
BGS_API int bgs_read_plies(bgs_batch* b, int32_t* plies);   /* plies played, int32[n] */
/* Connect: State::get_actions as a mask, uint8[n][width] (connect.cpp:43) */
BGS_API int bgs_read_legal(bgs_batch* b, uint8_t* legal);
/* number of legal actions of the side to move, int32[n] (len(state.actions)) */
BGS_API int bgs_read_action_count(bgs_batch* b, int32_t* count);
/* Bounce: State::get_actions_at for every column of the active row (bounce.cpp:41): uint64[n][width + 1]; entry
 * i < width has bit (y*width + x) set for every legal target of the piece in column i of the active row (0 if
 * none); entry [width] is the active row's y (all ones when the board has ended or nothing can move) */
BGS_API int bgs_bounce_read_targets(bgs_batch* b, uint64_t* targets);
/* the same observations into DEVICE memory, enqueued on the batch's stream (no synchronisation): what =
 * 'g' grid int8[n][h][w] (16-byte aligned destination), 'l' Connect legal mask uint8[n][w], 'c' action count int32[n],
 * 't' Bounce target masks uint64[n][w + 1], 'r' reward int8[n][2] */
BGS_API int bgs_export_device(bgs_batch* b, int what, void* device_dst);
/* N2, ONE call per policy ply: bgs_step_actions with the actions in DEVICE memory (int32[n], Bounce int32[n][4]), then --
 * in the same pass over the batch where the kernel exists (one-word Connect boards, even n; otherwise the separate
 * kernels back to back) -- what the policy needs for its next choice, of the boards AFTER the move: device_observation =
 * Connect uint8[n][width] legal mask ('l' above; 16-byte aligned), Bounce uint64[n][width + 1] target masks ('t');
 * device_ended uint8[n] State::has_ended (may be NULL); device_status int32[n] per-board result as bgs_step_actions
 * (may be NULL).  Everything is an enqueue on the batch's stream: no synchronisation, no allocation, capturable in a
 * HIP graph.  The loop README.md:57-65 / examples/agent.py:13-27 make per board, for a batch and a device-side policy:
 * observation -> policy -> bgs_step_actions_observe -> observation -> ... */
BGS_API int bgs_step_actions_observe(bgs_batch* b, const int32_t* device_actions, void* device_observation, uint8_t* device_ended,
                             int32_t* device_status);
/* ... and as the step of a VECTOR ENVIRONMENT (the learner's side of README.md:57-65 for n boards at once): the same call
 * plus device_reward int8[n][2] = State::get_reward of the boards after the move (connect.cpp:41 / bounce.cpp:38: the
 * finished game's pair where device_ended is set, 0 / 0 while it runs; may be NULL), and with BGS_ENV_AUTO_RESET a board
 * that has ended is put back to Config::sample_initial_state() (connect.cpp:32, bounce.cpp:29) in the same pass -- its
 * observation is then the new game's, its ended flag and reward still those of the game that just finished.  Bit-packed
 * boards only with BGS_ENV_AUTO_RESET. */
#define BGS_ENV_AUTO_RESET 1u
BGS_API int bgs_env_step(bgs_batch* b, const int32_t* device_actions, void* device_observation, uint8_t* device_ended,
                 int8_t* device_reward, int32_t* device_status, uint32_t flags);

/* ---- compact outcomes for the multi-GPU reward gather ------------------------------------------------ */
/* 2 bits per board (0 running, 1 / 2 that player won, 3 draw), 4 boards per byte, board 4i in the low bits:
 * device_dst uint8[(n + 3) / 4].  A reward pair (State::get_reward, connect.cpp:41 / bounce.cpp:38) is a function of
 * this code, so ranks exchange 0.25 B per game over xGMI instead of 2 B and expand after the gather. */
BGS_API int bgs_pack_outcomes(bgs_batch* b, void* device_dst);
/* bgs_rollout followed by bgs_pack_outcomes in one call; kernels that can (one-word Connect boards, compile-time
 * multi-word geometries) write the codes themselves, so no second launch follows the rollout.  device_dst: 16-byte
 * aligned, ((n + 63) / 64) * 16 bytes -- what a rank hands to the RCCL gather. */
BGS_API int bgs_rollout_pack(bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, void* device_dst);
/* inverse, batch-independent: packed codes of n boards -> reward int8[n][2] (8-byte aligned), on `device` / stream */
BGS_API int bgs_expand_outcomes(int device, void* hip_stream, const void* device_packed, int64_t n, int8_t* device_reward);

/* ---- asynchronous hand-over to HOST memory ------------------------------------------------------------
 * The reference hands `reward` out as a host ndarray on every call (State::get_reward, connect.cpp:41 / bounce.cpp:38,
 * tensor.hpp:69-87).  For a batch the hand-over is part of the path: these entry points enqueue it on the batch's
 * stream and return at once; the caller waits on a bgs_event (or bgs_synchronize) before reading the host buffer.
 * Destinations must be page-locked (bgs_host_alloc, hipHostMalloc or torch pin_memory) for the copy to be
 * asynchronous. */
typedef struct bgs_event bgs_event; /* opaque: a HIP event on the batch's device */
BGS_API int bgs_host_alloc(size_t bytes, void** host_ptr);
BGS_API int bgs_host_free(void* host_ptr);
BGS_API int bgs_event_create(int device, bgs_event** out);
BGS_API int bgs_event_destroy(bgs_event* e);
BGS_API int bgs_event_synchronize(bgs_event* e);       /* block the calling thread until the work recorded before it is done */
BGS_API int bgs_event_query(bgs_event* e, int* done);  /* *done = 1 when that work has completed */
/* reward int8[n][2] -> host_dst, then record `done` (may be NULL) */
BGS_API int bgs_read_reward_async(bgs_batch* b, int8_t* host_dst, bgs_event* done);
/* 2-bit outcome codes (as bgs_pack_outcomes) uint8[(n + 3) / 4] -> host_dst, then record `done` (may be NULL): 8x
 * fewer bytes over PCIe than the int8 pairs; bgs_expand_outcomes_host finishes the job on the host */
BGS_API int bgs_read_outcomes_async(bgs_batch* b, uint8_t* host_dst, bgs_event* done);
/* host-side inverse of bgs_pack_outcomes for games [first, first + count) (first a multiple of 4): a table look-up,
 * no game rule; reward int8[n][2] is indexed by game, so callers may split a batch over threads */
BGS_API int bgs_expand_outcomes_host(const uint8_t* packed, int64_t first, int64_t count, int8_t* reward);
/* bgs_rollout followed by bgs_read_reward_async (codes == 0) or bgs_read_outcomes_async (codes != 0): one call per
 * batch step for host loops that are launch-rate bound */
BGS_API int bgs_rollout_to_host(bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, void* host_dst, int codes,
                        bgs_event* done);

/* A reward sink delivers the rewards of successive batch steps into caller-owned host arrays int8[n][2] while the GPU
 * goes on playing: per submission the outcome codes cross PCIe into one of `slots` pinned buffers and `threads` host
 * worker threads expand them as soon as the copy has landed (with more than one thread the first one only waits for
 * the arrival events and releases the others, so the event latency of a submission overlaps the expansion of the one
 * before it).  Submissions complete in order.  Environment: BGS_SINK_SPIN_US (microseconds a waiter spins before it
 * sleeps, default 0), BGS_SINK_POLL (poll the arrival event), BGS_NO_STREAM_STORES. */
typedef struct bgs_reward_sink bgs_reward_sink;
BGS_API int bgs_sink_create(int device, int64_t max_games, int slots, int threads, bgs_reward_sink** out);
BGS_API int bgs_sink_destroy(bgs_reward_sink* s);
/* enqueue on the batch's stream: the pack kernel stores the codes straight into a page-locked slot (device-mapped host
 * memory: no copy call), an event marks their arrival, the workers expand into host_reward int8[n][2] (any host
 * memory); *ticket identifies the submission.  Blocks only while all slots are still in use. */
BGS_API int bgs_sink_submit(bgs_reward_sink* s, bgs_batch* b, int8_t* host_reward, int64_t* ticket);
/* bgs_rollout followed by bgs_sink_submit: one library call per batch step */
BGS_API int bgs_sink_rollout(bgs_reward_sink* s, bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags,
                     int8_t* host_reward, int64_t* ticket);
/* the same for packed codes that are already on the device (the RCCL-gathered codes of all ranks on rank 0):
 * device_packed uint8[(n_games + 3) / 4], copied on `hip_stream` */
BGS_API int bgs_sink_submit_packed(bgs_reward_sink* s, void* hip_stream, const void* device_packed, int64_t n_games,
                           int8_t* host_reward, int64_t* ticket);
BGS_API int bgs_sink_wait(bgs_reward_sink* s, int64_t ticket); /* until that submission's rewards are in its host array */
/* A GRID sink hands over the boards themselves -- State::get_grid (connect.cpp:42, bounce.cpp:39; the reference returns
 * a host array through the copying caster tensor.hpp:69-87) for every game of a step: the boards cross PCIe bit-packed
 * (Connect: two bit sets over the cells in reference order, 16 B per 6x7 board instead of 42; Bounce: the four value
 * bit-planes; generic batches: the int8 grid itself), one asynchronous copy per step into a page-locked slot, and the
 * worker threads expand them into the caller's int8[n][height][width] (AVX-512: two masked byte adds per 64 cells).
 * Made for batches like `like` (same game, geometry and size); bgs_sink_submit / bgs_sink_rollout / bgs_sink_wait /
 * bgs_sink_completed / bgs_sink_destroy and bgs_pipeline_* work as for a reward sink, with host_reward = the grid array. */
BGS_API int bgs_grid_sink_create(const bgs_batch* like, int slots, int threads, bgs_reward_sink** out);
/* the host half of it, for games [first, first + count) of a batch of n: `wire` holds `sets` bit sets over the cells
 * (cell = y * width + x), word j of set p of game i at ((uint64_t*)wire)[(p * nwc + j) * n + i], nwc = (cells + 63) / 64;
 * a cell's byte = offset + sum of weights[p] over the sets that contain it (Connect: offset -1, weights 1, 1 for
 * "occupied" and "player 1's"; Bounce: offset 0, weights 1, 2, 4, 8); sets = 0: the wire is the int8 grid.  A table
 * look-up per 8 cells or two masked byte adds per 64 (AVX-512; portable != 0 forces the table), no game rule. */
BGS_API int bgs_expand_grid_host(const void* wire, int64_t n, int cells, int sets, int offset, const int32_t* weights, int64_t first,
                         int64_t count, int8_t* grid, int portable);
/* Submissions to one sink may come from several threads (a ticket and its slot are reserved under the sink's lock);
 * they are delivered in ticket order.  *completed = number of submissions whose rewards are in their host arrays. */
BGS_API int bgs_sink_completed(bgs_reward_sink* s, int64_t* completed);

/* Progress words: monotonic int64 counters in host memory -- typically in a shared-memory segment several processes
 * map -- that consumers sleep on (futex on the low half) instead of polling.  bgs_sink_set_progress makes a sink
 * announce its completed count in *word after every delivery (word = NULL stops that; the word must outlive the sink or
 * be unset first); bgs_progress_store raises *word to `value` (never lowers it) and wakes the sleepers;
 * bgs_progress_wait blocks until each of the `count` words words[i * stride_words] is >= target, or fails with
 * BGS_ERR_RUNTIME after timeout_ms (< 0: no timeout), *laggard = index of the word that was behind. */
BGS_API int bgs_sink_set_progress(bgs_reward_sink* s, int64_t* word);
BGS_API int bgs_progress_store(int64_t* word, int64_t value);
/* A barrier of `count` processes on `count` such words (words[i * stride_words], word `mine` this process's): raises
 * its own word to `epoch` (1, 2, 3, ... from barrier to barrier), then waits until every word is >= epoch -- watching
 * them for up to spin_us microseconds before it sleeps as bgs_progress_wait does.  Ranks of one node that run in step
 * meet within a few microseconds, which a collective on the GPU (a launch, a kernel, a synchronise: tens of microseconds)
 * cannot match; bench.py brackets its timed region with it when the shared array exists. */
BGS_API int bgs_progress_barrier(int64_t* words, int64_t count, int64_t stride_words, int64_t mine, int64_t epoch, int64_t spin_us,
                         int64_t timeout_ms);
BGS_API int bgs_progress_wait(const int64_t* words, int64_t count, int64_t stride_words, int64_t target, int64_t timeout_ms,
                      int64_t* laggard);
/* Confine the calling thread (and the threads it creates later) to the CPUs of the NUMA node `device` hangs off,
 * intersected with what the process may use: its first-touch pages, the sink's workers and the launching thread then
 * sit next to the GPU's PCIe root.  *cpus = size of that set, 0 when the topology is unknown (nothing changed). */
BGS_API int bgs_bind_host_thread(int device, int* cpus);

/* ---- the reward gather over RCCL / xGMI, one process per GPU -------------------------------------------------------
 * The path shards without any exchange (rank r plays global game ids [r * n, (r + 1) * n), bgs_set_first_game); the one
 * collective is the hand-over: every rank's 2-bit outcome codes to rank 0's GPU, from there to rank 0's host array
 * int8[world * n][2] (State::get_reward of all games, connect.cpp:41, in global game order).  A bgs_gather owns a
 * persistent communicator (ncclCommInitRank), a communication stream and a communication thread; rank 0's also owns the
 * reward sink.  Per step the launching thread makes ONE call, bgs_gather_rollout, which enqueues the rollout on the
 * batch's stream and returns; send, receives, copy to the host and expansion follow behind it on other threads and
 * streams while the next rollouts play.  The launcher (torch.distributed, MPI, a file) only has to carry the 128-byte
 * id from rank 0 to the others.  RCCL is loaded on first use (dlopen "librccl.so.1"; BGS_RCCL_LIB=<path> names another
 * library with the same nine nccl* entry points: the tests' shared-memory stand-in, tests/c/fake_rccl.hip).
 * The machinery is per GROUP of steps (BGS_GATHER_BATCH, default slots / 2): one stream wait per launch stream, one
 * group of point-to-point calls, one copy kernel, one event.  Rank 0 receives into device memory and a copy kernel takes
 * the gathered codes to the sink's page-locked slots; BGS_GATHER_DIRECT=1 receives straight into the device-mapped slots.
 * With two ranks or more bgs_gather_create sends one message per peer through the transport in that mode and compares
 * what arrives in host memory (a direct receive that does not deliver falls back to the copy kernel, on stderr and in
 * bgs_gather_info).  A one-rank world has nothing to gather: no thread, no stream, a step is bgs_sink_rollout. */
#define BGS_UNIQUE_ID_BYTES 128
typedef struct bgs_gather bgs_gather;
BGS_API int bgs_gather_unique_id(uint8_t* id /* [BGS_UNIQUE_ID_BYTES] */);   /* rank 0; ncclGetUniqueId */
/* collective over the world (ncclCommInitRank).  n_per_rank: games per rank, a multiple of 4; slots: steps that may be
 * in flight (code buffers per rank; on rank 0 also sink slots); host_threads: rank 0's sink workers. */
BGS_API int bgs_gather_create(int device, int rank, int world, const uint8_t* id, int64_t n_per_rank, int slots, int host_threads,
                      bgs_gather** out);
/* bgs_rollout on `b`, then this rank's codes to rank 0 (and there: everybody's rewards into host_reward, which other
 * ranks pass as NULL).  Every rank makes the same sequence of calls; one thread at a time per gather.  The codes of a
 * step leave in a group with its neighbours: when BGS_GATHER_BATCH steps are there, when somebody waits for one of them
 * (bgs_gather_wait), or by themselves BGS_GATHER_FLUSH_US (default 1000) microseconds after the group's first step was
 * submitted -- so a rank that submits a few steps and then blocks on something else still delivers them.  A step that
 * cannot be enqueued on one rank fails THERE (this call, and every later call on that gather); the rank still posts the
 * step's message -- zeros: rank 0 delivers reward 0 / 0 for those rows -- so that no peer is left waiting. */
BGS_API int bgs_gather_rollout(bgs_gather* g, bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, int8_t* host_reward,
                       int64_t* ticket);
/* rank 0: that step's rewards of all ranks are in its host array; other ranks: this rank's codes have been sent */
BGS_API int bgs_gather_wait(bgs_gather* g, int64_t ticket);
/* how the gather runs: *direct 1 = receives straight into the sink's device-mapped slots, 0 = device memory + copy kernel;
 * *batch = steps per group of point-to-point calls; *transport_check 0 = none (one rank), 1 = the create-time message
 * arrived intact in the mode asked for, 2 = only after falling back from direct receives.  NULL pointers are skipped. */
BGS_API int bgs_gather_info(const bgs_gather* g, int* direct, int* batch, int* transport_check);
/* what the COMMUNICATOR says about itself, asked once when it was created: *ranks = ncclCommCount, *rank =
 * ncclCommUserRank (-1 each when the transport library has no such entry point).  The first line of an N-GPU run should
 * read ranks == N on every rank: bench.py prints it as gather_rccl.gather_info.ranks. */
BGS_API int bgs_gather_comm(const bgs_gather* g, int* ranks, int* rank);
/* name of the transport library in use ("librccl.so.1", or BGS_RCCL_LIB's path), "" when none could be loaded */
BGS_API const char* bgs_gather_transport(void);
BGS_API int bgs_gather_destroy(bgs_gather* g);

/* ---- the rollout loop as ONE call (README.md:45-72 `while not state.has_ended`, for batch after batch) ------------------
 * Step s (s = 0, 1, ... over the pipeline's life) plays every board of batches[s % depth] from the state `flags` says
 * to the end with seed seed0 + s on that batch's stream; with a hand-over the step's rewards go to
 * host_rewards[j % n_host], j = number of hand-overs so far, through `sink` (one GPU; or N ranks, each delivering its
 * rows of a shared array) or `gather` (RCCL to rank 0; other ranks pass NULL entries) -- exactly one of the two, or
 * neither (then n_host = 0 and every step stays on the device).  bgs_pipeline_enqueue returns when `count` more steps
 * are enqueued; it blocks only while the host array a step is about to reuse is still being delivered (so n_host
 * bounds how far the launching thread runs ahead).  time_stride > 0 brackets every time_stride-th launch of the call
 * with timing events on the batch's stream; bgs_pipeline_kernel_ms (after bgs_pipeline_drain) returns their mean and
 * resets them.  bgs_pipeline_drain: every enqueued step's rewards are in their host arrays and the streams are idle.
 * The batches, sink, gather and host arrays belong to the caller and must outlive the pipeline. */
typedef struct bgs_pipeline bgs_pipeline;
BGS_API int bgs_pipeline_create(bgs_batch* const* batches, int depth, bgs_reward_sink* sink, bgs_gather* gather,
                        int8_t* const* host_rewards, int n_host, uint64_t seed0, int32_t max_plies, uint32_t flags,
                        bgs_pipeline** out);
BGS_API int bgs_pipeline_enqueue(bgs_pipeline* p, int64_t count, int handover, int time_stride);
/* the same with the seed of every step given by the caller (seeds[i] for the i-th step of this call) instead of
 * seed0 + step index: a burst of a Python loop over arbitrary seeds (simulator.pipeline.RolloutPipeline.run) */
BGS_API int bgs_pipeline_enqueue_seeds(bgs_pipeline* p, const uint64_t* seeds, int64_t count, int handover);
/* The same for a consumer loop that cannot keep up with a launch per step itself (a Python generator): the seeds are FED, a
 * thread of the pipeline's own enqueues a fed step as soon as the host array it lands in is free -- hand-over j waits until
 * the caller has RELEASED hand-over j - n_host -- and the consumer only waits, reads, releases:
 *     bgs_pipeline_feed(p, seeds, K);  for j: bgs_pipeline_wait(p, j); read host_rewards[j % n_host]; bgs_pipeline_release(p, j);
 * bgs_pipeline_wait also waits for a fed hand-over to be enqueued; bgs_pipeline_drain enqueues and delivers whatever is
 * still fed (and releases every array); bgs_pipeline_destroy drops what was fed and not yet enqueued.  Not for pipelines
 * on a shared array (bgs_pipeline_set_ring).  Steps enqueued with bgs_pipeline_enqueue(_seeds) count as released. */
BGS_API int bgs_pipeline_feed(bgs_pipeline* p, const uint64_t* seeds, int64_t count);
BGS_API int bgs_pipeline_release(bgs_pipeline* p, int64_t handover_index);
/* until hand-over number `handover_index` (0, 1, ... over the pipeline's life) is in its host array
 * host_rewards[handover_index % n_host]; BGS_ERR_ARG when a later hand-over has already reused that array */
BGS_API int bgs_pipeline_wait(bgs_pipeline* p, int64_t handover_index);
BGS_API int bgs_pipeline_drain(bgs_pipeline* p);
BGS_API int bgs_pipeline_progress(const bgs_pipeline* p, int64_t* steps, int64_t* handovers);
BGS_API int bgs_pipeline_kernel_ms(bgs_pipeline* p, double* mean_ms, int* pairs);
/* The bracketed launches' start and end, in ms after the first bracket's start (after bgs_pipeline_drain, BEFORE
 * bgs_pipeline_kernel_ms resets the brackets): where the time of a short timed region goes. */
BGS_API int bgs_pipeline_timeline(bgs_pipeline* p, float* start_ms, float* end_ms, int capacity, int* pairs);
/* N ranks delivering into one shared host array (progress words, see above): rank r's sink announces its deliveries in
 * rank_words[r * word_stride] (bgs_sink_set_progress; the sink must serve this pipeline only), the consumer announces
 * the hand-overs it has released in *consumed.  Every rank: hand-over j waits for the release of hand-over j - n_host
 * before it overwrites that array.  The consumer rank (is_consumer; one per ring) also plays the consumer inside its
 * launch loop: before hand-over j it waits until ALL ranks have delivered hand-over j - lag (1 <= lag < n_host) and
 * releases it; bgs_pipeline_drain consumes the rest. */
BGS_API int bgs_pipeline_set_ring(bgs_pipeline* p, const int64_t* rank_words, int64_t word_stride, int world, int64_t* consumed,
                          int is_consumer, int lag, int64_t timeout_ms);
BGS_API int bgs_pipeline_destroy(bgs_pipeline* p);

/* ---- several GPUs of one node from one host process (no torch.distributed needed) ------------------------------
 * Device devices[r] plays Connect games with global ids [r * n_per_device, (r + 1) * n_per_device) from
 * Config::sample_initial_state() to the end (bgs_rollout), the devices' outcome codes are gathered on devices[0] with
 * RCCL point-to-point calls over xGMI, copied to the host once and expanded into host_reward
 * int8[n_devices * n_per_device][2] in global game order; *steps = env-steps of all devices.  The result equals
 * one batch of n_devices * n_per_device boards on one device.  n_per_device must be a multiple of 4.  One-shot: batches,
 * streams and communicators live for the call.  RCCL is loaded on first use (dlopen "librccl.so.1"). */
BGS_API int bgs_multi_connect_rollout(const int* devices, int n_devices, int height, int width, int count, int64_t n_per_device,
                              uint64_t seed, int8_t* host_reward, uint64_t* steps);
/* the same with everything kept between calls -- batches, streams, code buffers and the communicators (ncclCommInitAll)
 * live as long as the handle: bgs_multi_rollout plays one step (seed) on every device and returns with host_reward
 * int8[n_devices * n_per_device][2] filled and *steps = the step's env-steps */
typedef struct bgs_multi bgs_multi;
BGS_API int bgs_multi_create(const int* devices, int n_devices, int height, int width, int count, int64_t n_per_device, bgs_multi** out);
BGS_API int bgs_multi_rollout(bgs_multi* m, uint64_t seed, int8_t* host_reward, uint64_t* steps);
BGS_API int bgs_multi_destroy(bgs_multi* m);

/* ---- loading boards (State::from_json, connect.cpp:46 / bounce.cpp:45; policy-driven stepping) ---- */
/* grid int8[n][h][w]; player int8[n] (Connect: may be NULL, derived from the stone counts); winner int8[n]
 * (NULL = all running; Connect re-derives wins and draws from the grid when NULL); plies int32[n] (Bounce; NULL =
 * player parity).  status (host int32[n], may be NULL) reports malformed boards, which are left untouched. */
BGS_API int bgs_write_state(bgs_batch* b, const int8_t* grid, const int8_t* player, const int8_t* winner,
                    const int32_t* plies, int32_t* status);

/* ---- one round trip for the object API -------------------------------------------------------------- */
/* What one `State` / `Action` operation of the reference needs (connect.cpp:39-46,52; bounce.cpp:36-45,51), fused
 * into one upload, one launch sequence, one download and one synchronisation (batches of at most 4096 boards):
 *   grid != NULL     load the boards first (as bgs_write_state; player and winner required, plies optional);
 *   actions != NULL  then apply one caller-chosen move per board (as bgs_step_actions; negative first entry skips);
 *   then observe: grid int8[n][h][w], player int8[n], winner int8[n], plies int32[n] and the legal moves of the side
 *   to move -- Connect: uint8[n][width] mask, Bounce: uint64[n][width + 1] target masks (bgs_bounce_read_targets) --
 *   and reward int8[n][2] as the device holds it (State::get_reward; may be NULL).
 * status int32[n]: 0, BGS_ERR_ARG (malformed board: nothing loaded) or BGS_ERR_ILLEGAL (move refused). */
BGS_API int bgs_transition(bgs_batch* b, const int8_t* grid, const int8_t* player, const int8_t* winner, const int32_t* plies,
                   const int32_t* actions, int32_t* status, int8_t* grid_out, int8_t* player_out, int8_t* winner_out,
                   int32_t* plies_out, void* legal_out, int8_t* reward_out);

#ifdef __cplusplus
}
#endif
#endif
