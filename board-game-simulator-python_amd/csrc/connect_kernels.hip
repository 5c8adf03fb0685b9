// connect_kernels.hip -- Connect-k hot path on gfx950: enumerate -> sample -> drop -> k-in-a-row / draw -> reward.
//
// Replaces, N boards per launch, what the reference reaches through
//   State::get_actions / get_action_at   (src/simulator/game/connect.cpp:43-44)
//   Action::sample_next_state            (connect.cpp:52)
//   State::has_ended / get_reward        (connect.cpp:39,41)
// Rules as pinned by reference tests/test_connect.py:68-145 (see oracle/bgs_oracle.c for the plain restatement).
//
// Board packing.  Two bit-planes per board (stones of player 0, stones of player 1), column-major with one
// always-empty sentinel bit on top of every column: bit(x, y) = x * (H + 1) + y.  The sentinel stops vertical and
// diagonal shift-and-AND runs from leaking into the next column.  A plane is NW = ceil(W * (H + 1) / 64) uint64
// words; the batch stores plane-word j of all boards contiguously (SoA: planes[j][n]) so a wave reads 512
// contiguous bytes per word.  The side to move and the ply count are popcount parity / popcount of the planes.
//
// This is integer bit manipulation: no MFMA.  The per-ply kernels are HBM/L2 bound; the fused rollout keeps the
// board in registers and is VALU-issue bound.
#include <type_traits>

#include "bgs_common.h"
#include "bgs_internal.h"

// identity of this translation unit as compiled: hash of this file, the kernel headers and the compile flags (csrc/Makefile)
#ifndef BGS_TU_ID
#define BGS_TU_ID "unknown"
#endif
extern "C" const char bgs_tu_id_connect[] = BGS_TU_ID;

namespace bgs {
namespace {

// ------------------------------------------------------------------------------------------------
// multi-word bitboards
// ------------------------------------------------------------------------------------------------
template <int NW>
struct Bits {
    uint64_t w[NW];
};

template <int NW>
__device__ __forceinline__ Bits<NW> zero_bits() {
    Bits<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = 0;
    return r;
}

template <int NW>
__device__ __forceinline__ Bits<NW> operator&(const Bits<NW>& a, const Bits<NW>& b) {
    Bits<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] & b.w[i];
    return r;
}

template <int NW>
__device__ __forceinline__ Bits<NW> operator|(const Bits<NW>& a, const Bits<NW>& b) {
    Bits<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] | b.w[i];
    return r;
}

template <int NW>
__device__ __forceinline__ bool any(const Bits<NW>& a) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) acc |= a.w[i];
    return acc != 0;
}

template <int NW>
__device__ __forceinline__ uint32_t popcount(const Bits<NW>& a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) c += (uint32_t)__popcll(a.w[i]);
    return c;
}

// word `idx` of a (0 beyond the top); idx may be a run-time value: resolved with selects, never with
// dynamically indexed registers
template <int NW>
__device__ __forceinline__ uint64_t word_at(const Bits<NW>& a, int idx) {
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r = (idx == i) ? a.w[i] : r;
    return r;
}

// logical shift right by s bits, 0 <= s < 64 * NW (folds to constants when s is known at compile time)
template <int NW>
__device__ __forceinline__ Bits<NW> shr(const Bits<NW>& a, int s) {
    Bits<NW> r;
    if (NW == 1) {
        r.w[0] = s < 64 ? (a.w[0] >> s) : 0ull;
        return r;
    }
    const int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint64_t lo = word_at(a, i + ws);
        const uint64_t hi = word_at(a, i + ws + 1);
        r.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
    return r;
}

template <int NW>
__device__ __forceinline__ void set_bit(Bits<NW>& a, int t) {
    if (NW == 1) {
        a.w[0] |= 1ull << t;
        return;
    }
    const int ws = t >> 6;
    const uint64_t m = 1ull << (t & 63);
#pragma unroll
    for (int i = 0; i < NW; ++i) a.w[i] |= (ws == i) ? m : 0ull;
}

template <int NW>
__device__ __forceinline__ bool test_bit(const Bits<NW>& a, int t) {
    return (word_at(a, t >> 6) >> (t & 63)) & 1ull;
}

// ------------------------------------------------------------------------------------------------
// geometry: SH/SW/SK are compile-time dimensions (0 = take the run-time value)
// ------------------------------------------------------------------------------------------------
template <int NW_, int SH, int SW, int SK>
struct Geo {
    static constexpr int NW = NW_;
    static constexpr int STATIC_H = SH, STATIC_W = SW, STATIC_K = SK;
    static constexpr int MAXW = SW ? SW : BGS_CONNECT_MAX_W;  // unroll bound of per-column loops
    int rh, rw, rk;
    __device__ __forceinline__ int h() const { return SH ? SH : rh; }
    __device__ __forceinline__ int w() const { return SW ? SW : rw; }
    __device__ __forceinline__ int k() const { return SK ? SK : rk; }
    __device__ __forceinline__ uint32_t all_columns() const { return (1u << w()) - 1u; }
};

// k stones in a row anywhere on bitboard b: shift-and-AND with run doubling.
// directions: vertical 1, horizontal H+1, rising diagonal H+2, falling diagonal H.
template <class G>
__device__ __forceinline__ bool has_run(const G& g, const Bits<G::NW>& b) {
    const int k = g.k();
    const int dirs[4] = {1, g.h() + 1, g.h() + 2, g.h()};
    Bits<G::NW> hit = zero_bits<G::NW>();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        Bits<G::NW> m = b;
        int len = 1;
        while (2 * len <= k) {
            m = m & shr(m, len * dirs[d]);
            len *= 2;
        }
        if (len < k) m = m & shr(m, (k - len) * dirs[d]);
        hit = hit | m;
    }
    return any(hit);
}

// index of the i-th set bit of m (i < popcount(m))
__device__ __forceinline__ int select_bit(uint32_t m, uint32_t i, int max_bits) {
    for (int j = 0; j + 1 < max_bits; ++j) {
        const uint32_t t = m & (m - 1u);
        m = ((uint32_t)j < i) ? t : m;
    }
    return __ffs((int)m) - 1;
}

// one board in registers: stones of the side to move / of the other side, packed column heights, full columns
template <int NW>
struct Lane {
    Bits<NW> cur, opp;
    uint64_t hts;    // 4 bits per column
    uint32_t full;   // bit x set when column x is full
    uint32_t plies;
};

template <class G>
__device__ __forceinline__ Lane<G::NW> empty_lane() {
    Lane<G::NW> l;
    l.cur = zero_bits<G::NW>();
    l.opp = zero_bits<G::NW>();
    l.hts = 0;
    l.full = 0;
    l.plies = 0;
    return l;
}

template <class G>
__device__ __forceinline__ Lane<G::NW> make_lane(const G& g, const Bits<G::NW>& p0, const Bits<G::NW>& p1) {
    Lane<G::NW> l;
    l.plies = popcount(p0) + popcount(p1);
    const bool second = l.plies & 1u;
    l.cur = second ? p1 : p0;
    l.opp = second ? p0 : p1;
    const Bits<G::NW> occ = p0 | p1;
    const int h = g.h(), w = g.w();
    const uint64_t colmask = (1ull << (h + 1)) - 1ull;
    l.hts = 0;
    l.full = 0;
#pragma unroll
    for (int x = 0; x < G::MAXW; ++x) {
        if (x < w) {
            const uint32_t hx = (uint32_t)__popcll(shr(occ, x * (h + 1)).w[0] & colmask);
            l.hts |= (uint64_t)hx << (4 * x);
            l.full |= (hx == (uint32_t)h) ? (1u << x) : 0u;
        }
    }
    return l;
}

// drop the mover's stone into column `col` (not full), test for a win of the mover, then for a draw, hand the move
// to the other side; returns the new status byte (0 running, 1 / 2 winner, 3 draw)
template <class G>
__device__ __forceinline__ uint32_t drop_stone(const G& g, Lane<G::NW>& l, int col) {
    const int h = g.h();
    const uint32_t hx = (uint32_t)(l.hts >> (4 * col)) & 15u;
    set_bit(l.cur, col * (h + 1) + (int)hx);
    l.hts += 1ull << (4 * col);
    l.full |= (hx + 1u == (uint32_t)h) ? (1u << col) : 0u;
    const uint32_t mover = l.plies & 1u;
    const bool won = has_run(g, l.cur);
    const uint32_t st = won ? mover + 1u : (l.full == g.all_columns() ? BGS_ST_DRAW : BGS_ST_RUNNING);
    const Bits<G::NW> t = l.cur;
    l.cur = l.opp;
    l.opp = t;
    l.plies += 1u;
    return st;
}

// one uniformly sampled ply: the i-th legal column in ascending order
template <class G>
__device__ __forceinline__ uint32_t play_ply(const G& g, Lane<G::NW>& l, uint32_t draw) {
    const uint32_t legal = ~l.full & g.all_columns();
    const uint32_t n = (uint32_t)__popc(legal);
    return drop_stone(g, l, select_bit(legal, sample_index(draw, n), G::MAXW));
}

// a caller-chosen column; returns false (board untouched) when the column is full or out of range
template <class G>
__device__ __forceinline__ bool play_column(const G& g, Lane<G::NW>& l, int col, uint32_t& st) {
    if (col < 0 || col >= g.w() || ((l.full >> col) & 1u)) return false;
    st = drop_stone(g, l, col);
    return true;
}

template <int NW>
__device__ __forceinline__ void load_planes(const uint64_t* __restrict__ planes, int64_t n, int64_t i, Bits<NW>& p0,
                                            Bits<NW>& p1) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        p0.w[j] = planes[(int64_t)j * n + i];
        p1.w[j] = planes[(int64_t)(NW + j) * n + i];
    }
}

template <int NW>
__device__ __forceinline__ void lane_planes(const Lane<NW>& l, Bits<NW>& p0, Bits<NW>& p1) {
    const bool second = l.plies & 1u;  // side to move is player 1: cur holds player 1's stones
    p0 = second ? l.opp : l.cur;
    p1 = second ? l.cur : l.opp;
}

template <int NW>
__device__ __forceinline__ void store_planes(uint64_t* __restrict__ planes, int64_t n, int64_t i, const Bits<NW>& p0,
                                             const Bits<NW>& p1) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        planes[(int64_t)j * n + i] = p0.w[j];
        planes[(int64_t)(NW + j) * n + i] = p1.w[j];
    }
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BGS_BLOCK) k_connect_reset(uint64_t* __restrict__ planes, uint8_t* __restrict__ status,
                                                             uint16_t* __restrict__ reward, int64_t n, int words, int only_ended) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (only_ended && status[i] == 0) return;   // (bgs_env_step with BGS_ENV_AUTO_RESET: finished boards start over)
    for (int j = 0; j < words; ++j) planes[(int64_t)j * n + i] = 0;
    status[i] = 0;
    reward[i] = 0;
}

template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_step_actions(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                       int64_t n, const int32_t* __restrict__ actions, int32_t* __restrict__ result,
                       unsigned long long* __restrict__ steps) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        const int col = actions[i];
        int32_t rc = 0;
        if (col >= 0) {
            rc = -2;  // BGS_ERR_ILLEGAL
            if (status[i] == BGS_ST_RUNNING) {
                Bits<NW> p0, p1;
                load_planes<NW>(planes, n, i, p0, p1);
                Lane<NW> l = make_lane(g, p0, p1);
                uint32_t st = 0;
                if (play_column(g, l, col, st)) {
                    lane_planes(l, p0, p1);
                    store_planes<NW>(planes, n, i, p0, p1);
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

// ------------------------------------------------------------------------------------------------
// K2: fused rollout.  A board lives in registers from its first ply to its last; a wave owns a contiguous
// chunk of games and a lane that finishes its board takes the chunk's next game ("lane refill"), so lanes do
// not idle while the longest game of the wave is still running.  Refill happens every 4 plies, which keeps all
// lanes of a wave on the same word of their philox block (one philox4x32-10 per 4 plies per board).
// Results never depend on the launch geometry: RNG streams are keyed by global game id.
// ------------------------------------------------------------------------------------------------

// board policy A: any geometry, on top of Lane / play_ply
template <class G>
struct GenericGame {
    static constexpr int NW = G::NW;
    Lane<NW> l;
    __device__ __forceinline__ void init(const G&) { l = empty_lane<G>(); }
    __device__ __forceinline__ void load(const G& g, const Bits<NW>& p0, const Bits<NW>& p1) { l = make_lane(g, p0, p1); }
    uint32_t st;
    __device__ __forceinline__ uint32_t plies() const { return l.plies; }
    // returns true while the board is still running
    __device__ __forceinline__ bool ply(const G& g, uint32_t draw) {
        st = play_ply(g, l, draw);
        return st == BGS_ST_RUNNING;
    }
    __device__ __forceinline__ uint32_t status_after_ply() const { return st; }
    __device__ __forceinline__ void planes(Bits<NW>& p0, Bits<NW>& p1) const { lane_planes(l, p0, p1); }
};

// (a & b) | c in one VALU instruction (v_bitop3_b32, truth table 0xEA)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// four in a row on a one-word board, written for instruction count (every VALU instruction costs about one issue
// quad here, whatever its width): per direction two 64-bit shifts, two ANDs for the pairs, and the quads are
// accumulated with the fused (pairs & shifted pairs) | acc
__device__ __forceinline__ bool four_in_a_row(uint64_t b, int h) {
    const int dirs[4] = {1, h + 1, h + 2, h};
    uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint64_t s1 = b >> dirs[d];
        const uint32_t pl = (uint32_t)b & (uint32_t)s1, ph = (uint32_t)(b >> 32) & (uint32_t)(s1 >> 32);
        const uint64_t pairs = ((uint64_t)ph << 32) | pl;
        uint64_t s2;  // one v_lshrrev_b64 (hipcc would split this shift of two halves into alignbit + shift)
        asm("v_lshrrev_b64 %0, %1, %2" : "=v"(s2) : "s"(2 * dirs[d]), "v"(pairs));
        if (d == 0) {
            acc_lo = pl & (uint32_t)s2;
            acc_hi = ph & (uint32_t)(s2 >> 32);
        } else {
            acc_lo = and_or(pl, (uint32_t)s2, acc_lo);
            acc_hi = and_or(ph, (uint32_t)(s2 >> 32), acc_hi);
        }
    }
    return (acc_lo | acc_hi) != 0u;
}

// The same test split the way the rollout uses it: a run that the stone just dropped on `pos` completes is either
// vertical -- then it is the four cells ending at pos, one shift and one compare (a shift amount below zero wraps to
// 61..63 and leaves at most three bits, and a stone lower than row 3 has the previous column's always-empty sentinel in
// its window) -- or lies in one of the three other directions, tested on the whole board as above.
__device__ __forceinline__ bool four_in_a_row_at(uint64_t b, int h, uint32_t pos) {
    const int dirs[3] = {h + 1, h + 2, h};
    uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint64_t s1 = b >> dirs[d];
        const uint32_t pl = (uint32_t)b & (uint32_t)s1, ph = (uint32_t)(b >> 32) & (uint32_t)(s1 >> 32);
        const uint64_t pairs = ((uint64_t)ph << 32) | pl;
        uint64_t s2;
        asm("v_lshrrev_b64 %0, %1, %2" : "=v"(s2) : "s"(2 * dirs[d]), "v"(pairs));
        if (d == 0) {
            acc_lo = pl & (uint32_t)s2;
            acc_hi = ph & (uint32_t)(s2 >> 32);
        } else {
            acc_lo = and_or(pl, (uint32_t)s2, acc_lo);
            acc_hi = and_or(ph, (uint32_t)(s2 >> 32), acc_hi);
        }
    }
    uint32_t column = (uint32_t)(b >> ((pos - 3u) & 63u));  // the stone and the three cells below it
    asm("" : "+v"(column));  // (keeps the compare 32 bits wide: hipcc would otherwise widen it and add a move)
    return ((acc_lo | acc_hi) != 0u) | ((column & 15u) == 15u);
}

// board policy B: one-word boards with W <= 8 and H <= 8 (Connect4 6x7).  Column state is one nibble per column,
// v = (H + 7) - height, so bit 3 of the nibble says "column open"; the i-th open column is found without a loop:
// a multiply by 0x11111111 turns the open flags into per-nibble prefix counts, and a SWAR compare against the
// sampled index counts the columns whose prefix count is still <= index.
template <class G>
struct NibbleGame {
    static constexpr int NW = 1;
    static constexpr uint32_t ONES = 0x11111111u;
    uint64_t cur, opp;
    uint32_t hts, open, np;
    bool won;
    __device__ __forceinline__ static uint32_t top(const G& g) { return (uint32_t)g.h() + 7u; }
    __device__ __forceinline__ static uint32_t columns(const G& g) {
        return g.w() >= 8 ? ONES : (ONES & ((1u << (4 * g.w())) - 1u));
    }
    __device__ __forceinline__ void init(const G& g) {
        cur = 0;
        opp = 0;
        hts = top(g) * columns(g);
        open = columns(g);
        np = 0;
    }
    __device__ __forceinline__ void load(const G& g, const Bits<1>& p0, const Bits<1>& p1) {
        np = (uint32_t)__popcll(p0.w[0]) + (uint32_t)__popcll(p1.w[0]);
        const bool second = np & 1u;
        cur = second ? p1.w[0] : p0.w[0];
        opp = second ? p0.w[0] : p1.w[0];
        const uint64_t occ = p0.w[0] | p1.w[0];
        const int h = g.h(), w = g.w();
        hts = 0;
#pragma unroll
        for (int x = 0; x < G::MAXW; ++x) {
            if (x < w && x < 8) {
                const uint32_t hx = (uint32_t)__popcll((occ >> (x * (h + 1))) & ((1ull << (h + 1)) - 1ull));
                hts |= (top(g) - hx) << (4 * x);
            }
        }
        open = (hts >> 3) & ONES;
    }
    __device__ __forceinline__ uint32_t plies() const { return np; }
    // returns true while the board is still running
    __device__ __forceinline__ bool ply(const G& g, uint32_t draw) {
        const uint32_t n = (uint32_t)__popc(open);
        const uint32_t idx = sample_index(draw, n);
        // nibble x of t: 8 + idx - (open columns among 0..x); (idx - open) * ONES = idx * ONES - open * ONES (mod 2^32)
        const uint32_t t = (idx - open) * ONES + 0x88888888u;
        const uint32_t col = (uint32_t)__popc(t & 0x88888888u);           // columns whose prefix count is <= idx
        const uint32_t sh = col * 4u;
        const uint32_t v = (hts >> sh) & 15u;
        const uint32_t bit = (col * (uint32_t)(g.h() + 1) + top(g)) - v;
        cur |= 1ull << bit;
        hts -= 1u << sh;
        open = (hts >> 3) & ONES;
        if (g.k() == 4) {
            won = four_in_a_row(cur, g.h());
        } else {
            Bits<1> b;
            b.w[0] = cur;
            won = has_run(g, b);
        }
        const uint64_t tmp = cur;
        cur = opp;
        opp = tmp;
        np += 1u;
        return !won && open != 0u;
    }
    // status of the board right after the ply that ended it (the mover of that ply is (np - 1) & 1)
    __device__ __forceinline__ uint32_t status_after_ply() const {
        return won ? ((np - 1u) & 1u) + 1u : (open == 0u ? BGS_ST_DRAW : BGS_ST_RUNNING);
    }
    __device__ __forceinline__ void planes(Bits<1>& p0, Bits<1>& p1) const {
        const bool second = np & 1u;
        p0.w[0] = second ? opp : cur;
        p1.w[0] = second ? cur : opp;
    }
};

template <class G, class Game, bool FROM_INITIAL, bool CAPPED>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_rollout(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                  int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                  unsigned long long* __restrict__ steps, uint32_t games_per_wave, uint32_t per_ply) {
    constexpr int NW = G::NW;
    // wave-uniform queue of this wave's games: begin + [taken, avail)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    const int64_t begin = (int64_t)wave * games_per_wave;
    const int64_t end = begin + games_per_wave < n ? begin + games_per_wave : n;
    const uint32_t avail = begin < end ? (uint32_t)(end - begin) : 0u;
    uint32_t taken = 0;

    Game gm;
    gm.init(g);
    int64_t game = 0;
    uint32_t st = BGS_ST_RUNNING, first_ply = 0;
    bool live = false, finished = false;
    uint32_t stepped = 0;

    for (;;) {
        const uint64_t need = __builtin_amdgcn_ballot_w64(!live);
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (!live && taken + rank < avail) {
                game = begin + taken + rank;
                if (FROM_INITIAL) {
                    gm.init(g);
                    st = BGS_ST_RUNNING;
                } else {
                    Bits<NW> p0, p1;
                    load_planes<NW>(planes, n, game, p0, p1);
                    gm.load(g, p0, p1);
                    st = status[game];
                }
                first_ply = gm.plies();
                live = st == BGS_ST_RUNNING && (!CAPPED || first_ply < max_plies);
                finished = FROM_INITIAL && !live;
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }
        if (__builtin_amdgcn_ballot_w64(live)) {
            // the block's draws: one philox call serves the sixteen plies of four blocks (the four of this block under the
            // strict contract, per_ply: its words are the draws)
            Philox4 four = {{0u, 0u, 0u, 0u}};
            uint32_t word = 0;
            if (live) {
                four = connect_philox(per_ply, seed, first_game + (uint64_t)game, gm.plies());
                word = connect_word(four, gm.plies());
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                if (live && (FROM_INITIAL || (gm.plies() & 3u) == j)) {
                    const bool running = gm.ply(g, per_ply ? four.v[j] : sub_draw(word, j));
                    live = running && (!CAPPED || gm.plies() < max_plies);
                    if (!live) {
                        finished = true;
                        st = gm.status_after_ply();
                    }
                }
            }
        }
        if (finished) {
            Bits<NW> p0, p1;
            gm.planes(p0, p1);
            store_planes<NW>(planes, n, game, p0, p1);
            status[game] = (uint8_t)st;
            reward[game] = reward_pair(st);
            stepped += gm.plies() - first_ply;
            finished = false;
        }
        if (!__builtin_amdgcn_ballot_w64(live) && taken >= avail) break;
    }
    add_steps(steps, stepped);
}

// K1: one uniformly sampled ply per running board.  Algorithmic traffic per env-step: both planes + status in, the
// mover's plane out (+ status / reward when the board ends).  All loads are issued before the first use: with a
// dependent status -> planes chain every wave paid two memory latencies and the kernel was latency-, not
// bandwidth-bound.
template <class G, class Game>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_step_random(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                      int64_t n, uint64_t seed, uint64_t first_game, unsigned long long* __restrict__ steps, uint32_t per_ply) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        Bits<NW> p0, p1;
        const uint32_t st0 = status[i];
        load_planes<NW>(planes, n, i, p0, p1);
        if (st0 == BGS_ST_RUNNING) {
            Game gm;
            gm.load(g, p0, p1);
            const uint32_t ply = gm.plies();
            const Philox4 blk = connect_philox(per_ply, seed, first_game + (uint64_t)i, ply);
            const bool running = gm.ply(g, connect_draw(per_ply, blk, ply));
            gm.planes(p0, p1);
            const uint32_t mover = ply & 1u;  // only the mover's plane changed
#pragma unroll
            for (int j = 0; j < NW; ++j) planes[(int64_t)(mover * NW + j) * n + i] = mover ? p1.w[j] : p0.w[j];
            if (!running) {
                const uint32_t st = gm.status_after_ply();
                status[i] = (uint8_t)st;
                reward[i] = reward_pair(st);
            }
            stepped = 1;
        }
    }
    add_steps(steps, stepped);
}

// position of the k-th set bit of m (k < popcount(m)) without a loop: a popcount-guided binary search
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

// `count` uniformly sampled plies on one one-word board held in registers (K1s below).  No column heights are kept:
// with one always-empty sentinel bit on top of every column, (stones + column bottoms) carries through the stones of
// each column and leaves exactly one bit per column, on the cell the next stone would take -- masked to the real
// cells that is the list of legal moves AND the stone positions, and the idx-th legal column is its idx-th set bit.
// The idx-th set bit of `landing` (K1s): `landing` holds at most ONE bit per column field of S = h + 1 bits and never a field's
// top bit, so the search is arithmetic on the fields instead of the general popcount-guided search (forty instructions of
// the ply's hundred and fifty; round 5): a field is non-empty iff adding 2^(S-1) - 1 carries into its top bit; the number
// of non-empty fields up to field x is field x of (flags * bottoms) -- no carries between fields: a count is at most w --
// and field x of (idx - flags) * bottoms + tops keeps its top bit iff fewer than idx + 1 non-empty fields lie at or below x,
// i.e. iff the column sought lies above x: their number is that column.  K2a's nibble search, on fields of S bits.
// Needs w <= 2^(S-1) = 2^h (the counts must fit under a field's top bit): play_plies asks.
__device__ __forceinline__ uint32_t select_landing(uint64_t landing, uint64_t bottoms, uint64_t tops, uint32_t stride, uint32_t idx) {
    const uint64_t flags = ((landing + (tops - bottoms)) & tops) >> (stride - 1u);
    const uint64_t cmp = ((uint64_t)idx - flags) * bottoms + tops;
    const uint32_t col = (uint32_t)__popcll(cmp & tops);
    const uint64_t low = (1ull << stride) - 1ull;   // (uniform: a field's bits)
    return (uint32_t)__ffsll((unsigned long long)(landing & (low << (col * stride)))) - 1u;
}

template <bool SINGLE, bool PER_PLY, class G>
__device__ __forceinline__ uint32_t play_plies(const G& g, uint64_t bottoms, uint64_t cells, uint64_t& p0, uint64_t& p1,
                                               uint32_t& st, uint64_t seed, uint64_t game, uint32_t count) {
    if (st != BGS_ST_RUNNING) return 0u;
    if (SINGLE) count = 1u;  // (straight-line code: no loop, no second philox call)
    uint32_t ply = (uint32_t)__popcll(p0) + (uint32_t)__popcll(p1);
    const uint32_t full = (uint32_t)(g.h() * g.w());
    Philox4 blk = connect_philox<PER_PLY>(seed, game, ply);
    uint32_t played = 0;
    for (uint32_t q = 0; q < count; ++q) {
        const uint64_t landing = ((p0 | p1) + bottoms) & cells;
        const uint32_t idx = sample_index(connect_draw<PER_PLY>(blk, ply), (uint32_t)__popcll(landing));
        // (uniform; compile time for a static geometry.  Tall one-column boards keep the general search: fields of up to 16 bits)
        const bool by_fields = g.h() <= 15 && (uint32_t)g.w() <= (1u << g.h());
        const uint32_t pos = by_fields ? select_landing(landing, bottoms, bottoms << g.h(), (uint32_t)g.h() + 1u, idx)
                                       : select_bit64(landing, idx);
        const bool second = ply & 1u;
        uint64_t mine = (second ? p1 : p0) | (1ull << pos);
        p0 = second ? p0 : mine;
        p1 = second ? mine : p1;
        bool won;
        if (g.k() == 4) {
            won = four_in_a_row_at(mine, g.h(), pos);
        } else {
            Bits<1> b;
            b.w[0] = mine;
            won = has_run(g, b);
        }
        ++ply;
        ++played;
        if (won) { st = (second ? 2u : 1u); break; }
        if (ply == full) { st = BGS_ST_DRAW; break; }
        if ((ply & (PER_PLY ? 3u : 15u)) == 0u && q + 1u < count) blk = connect_philox<PER_PLY>(seed, game, ply);
    }
    return played;
}

// K1s: the streaming form of K1 for one-word boards and even batch sizes.  K1 is the one kernel of the path that
// really is HBM bound (a board is read and written once per ply), and it was held back by memory-level parallelism: a
// wave of K1 has 17 bytes per lane in flight and exits after 64 boards.  Here a lane owns PAIRS of boards -- 16-byte
// loads and stores on both planes -- the grid is sized to the chip and strides over the batch, and the loads of the
// next pair are issued before the current one is played, so every wave always has a pair's 34 bytes per lane in
// flight while it computes.  `count` plies are played per launch on the boards in registers (bgs_step_random_n): the
// per-ply traffic divides by count.
template <class G, bool SINGLE, bool PER_PLY>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_step_random_stream(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                             int64_t n, uint64_t seed, uint64_t first_game, unsigned long long* __restrict__ steps,
                             uint32_t count) {
    const int64_t pairs = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * BGS_BLOCK;
    uint64_t bottoms = 0, cells = 0;  // wave-uniform geometry masks
    for (int x = 0; x < g.w(); ++x) {
        bottoms |= 1ull << (x * (g.h() + 1));
        cells |= ((1ull << g.h()) - 1ull) << (x * (g.h() + 1));
    }
    const ulonglong2* plane0 = reinterpret_cast<const ulonglong2*>(planes);
    const ulonglong2* plane1 = reinterpret_cast<const ulonglong2*>(planes + n);
    const uint16_t* status2 = reinterpret_cast<const uint16_t*>(status);
    uint32_t stepped = 0;
    int64_t t = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    ulonglong2 a{0, 0}, b{0, 0};
    uint32_t s2 = 0;
    // one ply per launch touches every board exactly once: non-temporal accesses (nothing is worth keeping in L2)
    constexpr bool NT = SINGLE;
    auto load2 = [](const ulonglong2* p) -> ulonglong2 {
        if constexpr (NT) {
            const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
            return ulonglong2{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1)};
        } else {
            return *p;
        }
    };
    auto store2 = [](ulonglong2* p, ulonglong2 v) {
        if constexpr (NT) {
            unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
            __builtin_nontemporal_store(v.x, q);
            __builtin_nontemporal_store(v.y, q + 1);
        } else {
            *p = v;
        }
    };
    if (t < pairs) {
        a = load2(plane0 + t);
        b = load2(plane1 + t);
        s2 = status2[t];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (the first pair has arrived: the loop never has to wait at its top, see below)
    while (t < pairs) {
        // the next pair's loads go out before this pair is played
        const int64_t tn = t + stride;
        ulonglong2 an{0, 0}, bn{0, 0};
        uint32_t s2n = 0;
        if (tn < pairs) {
            an = load2(plane0 + tn);
            bn = load2(plane1 + tn);
            s2n = status2[tn];
        }
        uint32_t st0 = s2 & 255u, st1 = s2 >> 8;
        const uint32_t was0 = st0, was1 = st1;
        const uint64_t game = first_game + 2ull * (uint64_t)t;
        uint64_t p00 = a.x, p01 = b.x, p10 = a.y, p11 = b.y;  // board 2t: planes p00 / p01, board 2t + 1: p10 / p11
        const uint32_t n0 = play_plies<SINGLE, PER_PLY>(g, bottoms, cells, p00, p01, st0, seed, game, count);
        const uint32_t n1 = play_plies<SINGLE, PER_PLY>(g, bottoms, cells, p10, p11, st1, seed, game + 1ull, count);
        // The next pair's loads have had this pair's play to arrive: wait for them HERE, before this pair's stores go
        // out.  Left to itself the compiler waits at the top of the next iteration -- after that iteration's own loads
        // were issued, and with vmcnt(0) because a run-time number of stores sits between the two: it waited for the
        // loads it had just issued, every iteration paid a full memory latency and the "prefetch" was none (4.97 TB/s;
        // with this wait and the one in front of the loop 5.28).  Two pairs ahead (three register sets taking turns, a
        // loop unrolled by three, vmcnt(3)) was tried: 102 VGPRs, half the occupancy, and the compiler's own vmcnt(0)
        // back at two of the three joins.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
        if (n0 | n1) {
            // one ply changes only the mover's plane: when neither board of the pair changed plane 0 (or plane 1),
            // that 16-byte store is skipped -- in lock-step play both boards have the same side to move
            const bool touch0 = !SINGLE || p00 != a.x || p10 != a.y;
            const bool touch1 = !SINGLE || p01 != b.x || p11 != b.y;
            if (touch0) store2(reinterpret_cast<ulonglong2*>(planes) + t, ulonglong2{p00, p10});
            if (touch1) store2(reinterpret_cast<ulonglong2*>(planes + n) + t, ulonglong2{p01, p11});
            if (st0 != was0 || st1 != was1) {
                reinterpret_cast<uint16_t*>(status)[t] = (uint16_t)(st0 | (st1 << 8));
                reinterpret_cast<uint32_t*>(reward)[t] = (uint32_t)reward_pair(st0) | ((uint32_t)reward_pair(st1) << 16);
            }
        }
        stepped += n0 + n1;
        a = an;
        b = bn;
        s2 = s2n;
        t = tn;
    }
    add_steps(steps, stepped);
}

// N2, one launch per policy ply (round 4): apply the moves an external policy chose AND emit what it needs for its next
// choice -- the legal mask uint8[n][W] (State::get_actions as a mask, connect.cpp:43) and the ended flags -- in the same
// pass over the batch.  A policy ply used to be two library launches (bgs_export_device 'l', then bgs_step_actions), each
// reading every board; at 2^20 boards a per-ply launch is ~10 us whatever it does, so the two cost the loop twice what
// one does.  One-word boards, even batch: a lane owns a PAIR of boards as in K1s (16-byte accesses on both planes, 8-byte
// loads of the two actions); the legal bytes of the workgroup's 512 boards go through LDS and leave as 16-byte stores.
// A board that has ended ignores its action (result 0 for a negative one, BGS_ERR_ILLEGAL otherwise, as bgs_step_actions).
template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_step_observe(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                       int64_t n, const int32_t* __restrict__ actions, int32_t* __restrict__ result,
                       uint8_t* __restrict__ legal, uint8_t* __restrict__ ended, unsigned long long* __restrict__ steps,
                       uint16_t* __restrict__ reward_out, uint32_t auto_reset) {
    extern __shared__ __attribute__((aligned(16))) uint8_t legal_tile[];   // [2 * BGS_BLOCK boards][W] bytes, in output order
    const int64_t pairs = n >> 1;
    const int h = g.h(), w = g.w();
    uint64_t bottoms = 0, cells = 0;  // wave-uniform geometry masks
    for (int x = 0; x < w; ++x) {
        bottoms |= 1ull << (x * (h + 1));
        cells |= ((1ull << h) - 1ull) << (x * (h + 1));
    }
    const uint32_t full = (uint32_t)(h * w);
    const int64_t t = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (t < pairs) {
        const ulonglong2 a = reinterpret_cast<const ulonglong2*>(planes)[t];
        const ulonglong2 b = reinterpret_cast<const ulonglong2*>(planes + n)[t];
        const uint32_t s2 = reinterpret_cast<const uint16_t*>(status)[t];
        const int2 act = reinterpret_cast<const int2*>(actions)[t];
        uint64_t p[2][2] = {{a.x, b.x}, {a.y, b.y}};   // [board of the pair][player]
        uint32_t st[2] = {s2 & 255u, s2 >> 8};
        const int col[2] = {act.x, act.y};
        int32_t rc[2] = {0, 0};
        bool moved[2] = {false, false};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (col[q] >= 0) {
                rc[q] = -2;  // BGS_ERR_ILLEGAL unless the move below goes through
                const uint64_t landing = ((p[q][0] | p[q][1]) + bottoms) & cells;   // one bit per open column: its next cell
                const uint64_t column = col[q] < w ? (((1ull << h) - 1ull) << (col[q] * (h + 1))) : 0ull;
                const uint64_t stone = landing & column;
                if (st[q] == BGS_ST_RUNNING && stone) {
                    const uint32_t ply = (uint32_t)__popcll(p[q][0] | p[q][1]);
                    const uint32_t me = ply & 1u;
                    const uint64_t mine = (me ? p[q][1] : p[q][0]) | stone;
                    p[q][0] = me ? p[q][0] : mine;
                    p[q][1] = me ? mine : p[q][1];
                    bool won;
                    if (g.k() == 4) {
                        won = four_in_a_row_at(mine, h, (uint32_t)__ffsll((unsigned long long)stone) - 1u);
                    } else {
                        Bits<1> bits;
                        bits.w[0] = mine;
                        won = has_run(g, bits);
                    }
                    st[q] = won ? me + 1u : (ply + 1u == full ? (uint32_t)BGS_ST_DRAW : (uint32_t)BGS_ST_RUNNING);
                    moved[q] = true;
                    rc[q] = 0;
                    ++stepped;
                }
            }
        }
        // what the caller learns about the boards AFTER the move: ended flags and, for a vector environment, the reward pairs
        if (ended) reinterpret_cast<uint16_t*>(ended)[t] = (uint16_t)((st[0] != 0u ? 1u : 0u) | (st[1] != 0u ? 256u : 0u));
        if (reward_out) reinterpret_cast<uint32_t*>(reward_out)[t] = (uint32_t)reward_pair(st[0]) | ((uint32_t)reward_pair(st[1]) << 16);
        if (auto_reset) {   // a finished board starts over in the same pass: the observation below is the new game's
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (st[q] != BGS_ST_RUNNING) {
                    p[q][0] = p[q][1] = 0;
                    st[q] = BGS_ST_RUNNING;
                    moved[q] = true;   // (the board in memory changes)
                }
        }
        if (moved[0] | moved[1]) {
            // one ply changes only the mover's plane: a plane neither board of the pair changed is not stored
            if (p[0][0] != a.x || p[1][0] != a.y) reinterpret_cast<ulonglong2*>(planes)[t] = ulonglong2{p[0][0], p[1][0]};
            if (p[0][1] != b.x || p[1][1] != b.y) reinterpret_cast<ulonglong2*>(planes + n)[t] = ulonglong2{p[0][1], p[1][1]};
            if (st[0] != (s2 & 255u) || st[1] != (s2 >> 8)) {
                reinterpret_cast<uint16_t*>(status)[t] = (uint16_t)(st[0] | (st[1] << 8));
                reinterpret_cast<uint32_t*>(reward)[t] = (uint32_t)reward_pair(st[0]) | ((uint32_t)reward_pair(st[1]) << 16);
            }
        }
        if (result) reinterpret_cast<int2*>(result)[t] = int2{rc[0], rc[1]};
        // the observation AFTER the move: which columns the side to move may play
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint64_t landing = st[q] == BGS_ST_RUNNING ? (((p[q][0] | p[q][1]) + bottoms) & cells) : 0ull;
            uint8_t* mine = legal_tile + (2 * threadIdx.x + q) * w;
            for (int x = 0; x < w; ++x) mine[x] = (uint8_t)(((landing >> (x * (h + 1))) & ((1ull << h) - 1ull)) != 0ull);
        }
    }
    __syncthreads();
    // the workgroup's 2 * BGS_BLOCK * W legal bytes are contiguous in the output and start at a multiple of 16
    const int64_t first_board = (int64_t)blockIdx.x * BGS_BLOCK * 2;
    const int64_t boards_here = n - first_board < 2 * BGS_BLOCK ? (n & ~(int64_t)1) - first_board : 2 * BGS_BLOCK;
    if (boards_here > 0) {
        const int64_t bytes = boards_here * w;
        uint8_t* dst = legal + first_board * w;
        for (int64_t o = (int64_t)threadIdx.x * 16; o < bytes; o += (int64_t)BGS_BLOCK * 16) {
            if (o + 16 <= bytes) {
                *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(legal_tile + o);
            } else {
                for (int64_t r = o; r < bytes; ++r) dst[r] = legal_tile[r];
            }
        }
    }
    add_steps(steps, stepped);
}

// the ended flags alone (the fall-back of bgs_step_actions_observe for the batches the fused kernel does not cover)
__global__ void __launch_bounds__(BGS_BLOCK) k_status_to_ended(const uint8_t* __restrict__ status, int64_t n, uint8_t* __restrict__ ended) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i < n) ended[i] = status[i] != 0;
}

// Outcome codes of a wave's chunk, accumulated where the games end (the fused form of k_pack_outcomes): 2 bits per
// game, 16 games per dword, in a wave-private slice of LDS; when the chunk is finished the wave stores its dwords to
// `codes_out` -- for the host hand-over that is page-locked HOST memory mapped into the device, so the codes are in
// host memory when the kernel completes and no pack kernel or copy has to follow it on the stream.  A chunk starts at
// a multiple of 16 games (the launcher rounds chunks to 64), so no two waves share a dword.
struct WaveCodes {
    uint32_t* slice;   // this wave's LDS dwords
    uint32_t words;    // dwords that hold games of this wave's chunk
    __device__ __forceinline__ void init(uint32_t* lds, uint32_t games_per_wave, uint32_t avail) {
        slice = lds + (threadIdx.x >> 6) * (games_per_wave >> 4);
        words = (avail + 15u) >> 4;
        for (uint32_t i = threadIdx.x & 63u; i < words; i += 64u) slice[i] = 0u;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void add(uint32_t game, uint32_t code) {
        atomicOr(slice + (game >> 4), code << (2u * (game & 15u)));  // ds_or_b32, no return
    }
    __device__ __forceinline__ void flush(uint32_t* __restrict__ codes_out, int64_t begin) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the ds_or of every lane before the reads below
        __builtin_amdgcn_wave_barrier();
        uint32_t* dst = codes_out + (begin >> 4);
        for (uint32_t i = threadIdx.x & 63u; i < words; i += 64u) dst[i] = slice[i];
    }
};

// K2a: the same rollout for boards that start from the initial state on a one-word geometry (W <= 8, H <= 8),
// written without per-ply control flow.  Every game starts at a 4-ply boundary, so inside a block the mover of
// sub-step j is player j & 1: no plane swap, no ply counter.  A lane that is not playing executes the same
// instructions with `live` = 0 -- the stone it drops is (live << position) = 0 -- so the four plies of a block and
// the philox call in front of them form ONE basic block for the scheduler; only refill and store are conditional.
template <class G, bool CAPPED, bool FROM_INITIAL, bool CODES, bool PER_PLY>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_rollout_aligned(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                          int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                          unsigned long long* __restrict__ steps, uint32_t games_per_wave, uint32_t* __restrict__ codes_out) {
    extern __shared__ uint32_t code_lds[];  // CODES: games_per_wave / 16 dwords per wave
    constexpr uint32_t ONES = 0x11111111u;
    const uint32_t top = (uint32_t)g.h() + 7u;
    const uint32_t columns = g.w() >= 8 ? ONES : (ONES & ((1u << (4 * g.w())) - 1u));
    const uint32_t stride = (uint32_t)g.h() + 1u;
    const uint32_t column_top = 8u * columns;  // bit 3 of the nibbles of real columns only: the count stays <= W
    const uint64_t eights = 0x88888888ull;  // the addend of the column select's multiply-add, in a register pair

    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    const int64_t begin = (int64_t)wave * games_per_wave;
    const int64_t end = begin + games_per_wave < n ? begin + games_per_wave : n;
    const uint32_t avail = begin < end ? (uint32_t)(end - begin) : 0u;
    uint32_t taken = 0;
    // the chunk's slices of the batch arrays (wave-uniform bases; lanes index them with their 32-bit game offset)
    uint64_t* __restrict__ const plane0 = planes + begin;
    uint64_t* __restrict__ const plane1 = planes + n + begin;
    uint8_t* __restrict__ const status_out = status + begin;
    uint16_t* __restrict__ const reward_out = reward + begin;

    uint64_t p[2] = {0, 0};      // stones of player 0 / player 1
    uint32_t hts = 0;            // nibble per column: (H + 7) - height; bit 3 = column open
    uint32_t blk = 0;            // 4-ply blocks this game has played
    uint32_t live = 0;           // all ones while this lane's game is running, else 0 (a mask: see the stone below)
    uint32_t st = 0;             // winner code once somebody won
    uint32_t game = 0;           // offset of this lane's game in the wave's chunk
    uint32_t skip = 0;           // loaded boards: sub-steps to sit out in the first block (= plies already in it)
    uint32_t stepped = 0;

    if (avail == 0u) return;  // (whole wave: the chunk is empty; nothing was counted)
    WaveCodes codes;
    if (CODES) codes.init(code_lds, games_per_wave, avail);
    do {
        // ---- refill: idle lanes take the next games of the chunk
        const uint64_t need = __builtin_amdgcn_ballot_w64(live == 0);
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (live == 0 && taken + rank < avail) {
                game = taken + rank;
                st = 0;
                if (FROM_INITIAL) {
                    p[0] = 0;
                    p[1] = 0;
                    hts = top * columns;
                    blk = 0;
                    live = (!CAPPED || max_plies > 0u) ? ~0u : 0u;
                    if (CAPPED && live == 0) {  // max_plies == 0: the boards still have to be written once
                        plane0[game] = 0;
                        plane1[game] = 0;
                        status_out[game] = 0;
                        reward_out[game] = 0;
                    }
                } else {
                    // a board from memory joins at its own ply: ply p is sub-step p & 3 of block p >> 2, so the lane
                    // sits out the first (p & 3) sub-steps of its first block and the mover of sub-step j is still
                    // player j & 1
                    p[0] = plane0[game];
                    p[1] = plane1[game];
                    const uint32_t ply0 = (uint32_t)__popcll(p[0]) + (uint32_t)__popcll(p[1]);
                    const uint64_t occ = p[0] | p[1];
                    hts = 0;
#pragma unroll
                    for (int x = 0; x < G::MAXW; ++x) {
                        if (x < g.w() && x < 8) {
                            const uint32_t hx = (uint32_t)__popcll((occ >> (x * (int)stride)) & ((1ull << stride) - 1ull));
                            hts |= (top - hx) << (4 * x);
                        }
                    }
                    blk = ply0 >> 2;
                    skip = ply0 & 3u;
                    const uint32_t st0 = status_out[game];
                    live = (st0 == BGS_ST_RUNNING && (!CAPPED || ply0 < max_plies)) ? ~0u : 0u;
                    if (CODES && live == 0) codes.add(game, st0);  // a board that does not play keeps its outcome
                }
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }

        // ---- the block's word (one philox call covers four blocks; lanes sit in different blocks, so it is made every
        // time), four plies, no control flow
        const uint32_t was_live = live;
        const BlockDraws<PER_PLY> draws(seed, first_game + (uint64_t)(begin + game), blk);
        uint32_t open = (hts >> 3) & ONES;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t cnt = (uint32_t)__popc(open);
            const uint32_t idx = sample_index(draws.draw(j), cnt);
            // nibble x of cmp = 8 + idx - (open columns among 0..x): (idx - open) * ONES is idx * ONES - open * ONES
            uint64_t cmp64, carry;  // (idx - open) * ONES + 0x88888888 as one v_mad_u64_u32 (hipcc would pick mul_lo + add)
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(cmp64), "=s"(carry) : "v"(idx - open), "s"(ONES), "v"(eights));
            const uint32_t cmp = (uint32_t)cmp64;
            uint32_t col = (uint32_t)__popc(cmp & column_top);  // columns whose prefix count is still <= idx
            if (g.w() >= 8) col &= 7u;                         // (a full 8-wide board would give 8)
            const uint32_t sh = col * 4u;
            const uint32_t v = (hts >> sh) & 15u;
            uint32_t base = col * stride + top;
            asm("" : "+v"(base));  // keep (col * stride + top) one multiply-add; then one subtract
            uint32_t pos = base - v;
            if ((uint32_t)g.w() * stride + top > 63u) pos &= 63u;  // only a lane that is not playing can exceed 63
            const uint32_t act = (FROM_INITIAL || j >= skip) ? live : 0u;  // mask: this lane plays this sub-step
            uint64_t& mine = p[j & 1u];
            // the stone of a lane that is not playing is masked away: (bit & act) | mine, one v_bitop3 per half
            const uint64_t bit = 1ull << pos;
            mine = ((uint64_t)and_or((uint32_t)(bit >> 32), act, (uint32_t)(mine >> 32)) << 32) |
                   and_or((uint32_t)bit, act, (uint32_t)mine);
            hts += act << sh;  // act is 0 or -1: one stone more in the column = its nibble one lower
            open = (hts >> 3) & ONES;
            bool won;
            if (g.k() == 4) {
                won = four_in_a_row_at(mine, g.h(), pos);
            } else {
                Bits<1> b;
                b.w[0] = mine;
                won = has_run(g, b);
            }
            stepped -= act;
            // no "act &&": a lane that is not playing re-tests a plane that did not change.  A running game holds no
            // run, so `won` can only be true there for the plane that ended this lane's game -- it names the same
            // winner again -- or on a lane whose result is not stored any more
            st = won ? (j & 1u) + 1u : st;
            live = (won || open == 0u) ? 0u : live;
            if (CAPPED) live = (4u * blk + j + 1u < max_plies) ? live : 0u;
        }
        blk += 1u;
        skip = 0;

        // ---- boards that ended in this block go to memory
        if (was_live != 0 && live == 0) {
            const uint32_t code = st ? st : (open == 0u ? BGS_ST_DRAW : BGS_ST_RUNNING);
            // scalar base + 32-bit byte offset: the addressing mode that needs no 64-bit VALU arithmetic per lane
            *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(plane0) + (game * 8u)) = p[0];
            *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(plane1) + (game * 8u)) = p[1];
            *(reinterpret_cast<uint8_t*>(status_out) + game) = (uint8_t)code;
            *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(reward_out) + (game * 2u)) = reward_pair(code);
            if (CODES) codes.add(game, code);
        }
    } while (__builtin_amdgcn_ballot_w64(live != 0) || taken < avail);
    if (CODES) codes.flush(codes_out, begin);
    add_steps(steps, stepped);
}

// K2o: K2a with an opening stage.  Two things cost K2a more than the plies themselves:
//   * every ply pays for a run test and for the search of the idx-th open column.  Neither is needed at the start of
//     a game: in the first four plies (OPEN_BLOCKS >= 1; needs H >= 4, K >= 3, W >= 2) nobody can win, no column can
//     fill and the board cannot fill, so a ply is "column = floor(draw * W / 2^32), drop" -- ten instructions; with
//     K >= 4 and H >= 6 (OPEN_BLOCKS >= 2) the same holds for plies 4 and 5, and plies 6 and 7 are full plies.
//   * the lane-refill loop carries ~45 instructions of refill / store code around every 4-ply block.  The wave
//     therefore opens 64 games AT ONCE, all lanes busy and no refill code, whenever its pool of opened boards runs
//     dry, and parks them in a wave-private ring in LDS (planes + column nibbles + the words of the blocks to come); idle
//     lanes refill from the ring and join the main loop at block OPEN_BLOCKS (blocks 2.. of the opening are full blocks
//     in lock step).
//     A game that ends inside the opening is stored by the opening stage and parked as a dead slot (column word 0).
// Status and reward are not stored per game either: a finished game leaves ONE outcome byte in the wave's LDS slice (a
// plain ds_write_b8), and when the chunk is done a lane reads four games as one dword -- their four status bytes as they
// go to memory -- and derives their reward pairs and their byte of 2-bit codes for the hand-over from it, coalesced.  Results are those of K2a bit for bit: the draws are keyed by (game, block).
//
// (Tried and measured, not kept: sharing the drain inside a workgroup -- a wave whose chunk is exhausted parks its last
// <= 32 boards in LDS for the waves still running and leaves.  It cut another 6 % of the instructions and made the
// kernel slower, alone and three in flight: the adopted boards lengthen ONE wave per workgroup, i.e. one SIMD of the
// CU, and with two workgroups per CU that imbalance is not averaged out.)
// Round 5: the ply loop holds NO philox.  With a word per block of four plies (bgs_common.h) one philox call serves
// sixteen plies, a whole game of a board of at most 48 cells needs three -- what the opening stage computed for its own
// three blocks before -- and the opening parks the words of the blocks still to come (block OPEN_BLOCKS .. the last one a
// full board can reach: eight for Connect4) next to the board.  A lane that takes a board takes its words into registers;
// lanes sit in different blocks of their games, so the words go to a lane-private column of LDS and a block reads one.
//   222 -> ~180 VALU a block (42 of philox and 8 of per-ply winner / step bookkeeping out, 3 sub-draw multiplies in).
// The pool is a ring of 64 slots (128 before: with the words a slot is 52 bytes, and six workgroups a CU -- three launches
// in flight -- leave each wave 6 KB of LDS): when it runs dry the lanes that need a board first take what is left, THEN
// all 64 lanes open the chunk's next 64 games over the emptied slots, then the remaining needy lanes take from those.
template <class G, int OPEN_BLOCKS, bool PER_PLY = false>
struct OpenedWords {
    // the last block a game can reach; run-time geometries: the launcher admits boards of at most 48 cells (12 blocks)
    static constexpr int LAST = G::STATIC_H > 0 ? (G::STATIC_H * G::STATIC_W + 3) / 4 - 1 : 11;
    // words parked with a board (the strict contract parks none: its loop makes a philox call a block, as round 4's did)
    static constexpr int COUNT = PER_PLY ? 1 : (LAST - OPEN_BLOCKS + 1 > 1 ? LAST - OPEN_BLOCKS + 1 : 1);
    static constexpr int QUADS = (COUNT + 3) / 4;
    static_assert(LAST <= 11 && OPEN_BLOCKS >= 1 && OPEN_BLOCKS <= 4, "three philox calls cover blocks 0 .. 11");
};

template <int QUADS>
struct OpenedPool {  // per wave
    static constexpr uint32_t SLOTS = 64;
    uint64_t plane[2][SLOTS];
    uint32_t cols[SLOTS];          // column nibbles, 0 = dead slot (the opening stage has stored that game)
    uint4 words[QUADS][SLOTS];     // the words of blocks OPEN_BLOCKS, OPEN_BLOCKS + 1, ...
};

// PER_PLY (round 6): the same kernel under the strict RNG contract -- a philox word per ply.  Nothing is parked beside a
// board: a block makes its own philox call (its four words are its draws), the opening one per block.
template <class G, int OPEN_BLOCKS, bool CODES, bool PER_PLY>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_rollout_opened(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                         int64_t n, uint64_t seed, uint64_t first_game, unsigned long long* __restrict__ steps,
                         uint32_t games_per_wave, uint32_t* __restrict__ codes_out) {
    using OW = OpenedWords<G, OPEN_BLOCKS, PER_PLY>;
    constexpr int NWORDS = OW::QUADS * 4;
    using Pool = OpenedPool<OW::QUADS>;
    extern __shared__ uint32_t code_lds[];  // one outcome BYTE per game of the wave's chunk: games_per_wave / 4 dwords per wave
    __shared__ Pool pools[BGS_BLOCK / BGS_WAVE];
    // the words of a lane's blocks to come, [word][lane] per wave (the bank is the lane: no conflicts whatever word a lane
    // reads): a lane that takes a board copies them here, a block reads ONE of them -- its address moves on by a row.  (In
    // registers the eight words had to move down one every block: 8 moves; this is an add, a min and an LDS read that is
    // issued a block ahead of its use.)
    __shared__ uint32_t lane_words[BGS_BLOCK / BGS_WAVE][NWORDS][BGS_WAVE];
    constexpr uint32_t ONES = 0x11111111u;
    const uint32_t top = (uint32_t)g.h() + 7u;
    const uint32_t columns = g.w() >= 8 ? ONES : (ONES & ((1u << (4 * g.w())) - 1u));
    const uint32_t stride = (uint32_t)g.h() + 1u;
    const uint32_t column_top = 8u * columns;
    const uint64_t eights = 0x88888888ull;  // the addend of the column select's multiply-add, in a register pair

    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t begin = (int64_t)wave * games_per_wave;
    const int64_t end = begin + games_per_wave < n ? begin + games_per_wave : n;
    const uint32_t avail = begin < end ? (uint32_t)(end - begin) : 0u;
    uint32_t taken = 0;    // games handed to lanes
    uint32_t opened = 0;   // games whose opening has been played (a multiple of 64); opened - taken <= 64 games wait in the pool
    uint64_t* __restrict__ const plane0 = planes + begin;
    uint64_t* __restrict__ const plane1 = planes + n + begin;
    Pool& pool = pools[threadIdx.x >> 6];

    uint64_t p[2] = {0, 0};
    uint32_t hts = 0, live = 0, game = 0, stepped = 0;
    bool anywon = false;   // somebody has a run on this lane's board (only read when the game has just ended)
    uint32_t wnext = 0;    // the word of this lane's next block (read from lane_words a block ahead)
    uint32_t* const my_words = &lane_words[threadIdx.x >> 6][0][lane];
    uint32_t wrow = 0;     // the row of lane_words the block after next reads
    uint32_t myblk = 0;    // PER_PLY: the block this lane's game plays next

    if (avail == 0u) return;
    // The outcome of game i of the chunk is byte i of the wave's LDS slice (a plain byte store where the game ends, no
    // read-modify-write): at the end a lane reads four games as one dword -- which IS their four status bytes.
    uint8_t* const outcome = reinterpret_cast<uint8_t*>(code_lds + (threadIdx.x >> 6) * (games_per_wave >> 2));
    for (uint32_t i = lane; i < ((avail + 3u) >> 2); i += BGS_WAVE) reinterpret_cast<uint32_t*>(outcome)[i] = 0u;
    __builtin_amdgcn_wave_barrier();

    // one full ply of sub-step J on (q, h4, op, alive, won_any); the body of K2a's block
    // Neither the winner nor the env-steps are tracked ply by ply: a game's plies are the stones on its board, and when it
    // has ended with a run the winner is whoever placed the last stone -- both read off the planes once, where the game ends
    // (round 5: 8 VALU a block less than a per-ply winner select and step count).
    auto full_ply = [&](auto j_tag, uint32_t draw, uint64_t (&q)[2], uint32_t& h4, uint32_t& op, uint32_t& alive,
                        bool& won_any) {
        constexpr uint32_t J = decltype(j_tag)::value;
        const uint32_t cnt = (uint32_t)__popc(op);
        const uint32_t idx = sample_index(draw, cnt);
        // nibble x of cmp = 8 + idx - (open columns among 0..x): (idx - open) * ONES + 0x88888888 as ONE v_mad_u64_u32
        // (hipcc, needing only the low word, picks v_mul_lo_u32 + v_add_u32: two instructions, 7.6 instead of 5.3 cycles)
        uint64_t cmp64;
        {
            uint64_t carry;
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(cmp64), "=s"(carry) : "v"(idx - op), "s"(ONES), "v"(eights));
        }
        const uint32_t cmp = (uint32_t)cmp64;
        uint32_t col = (uint32_t)__popc(cmp & column_top);
        if (g.w() >= 8) col &= 7u;
        const uint32_t sh = col * 4u;
        const uint32_t v = (h4 >> sh) & 15u;
        uint32_t base = col * stride + top;
        asm("" : "+v"(base));
        uint32_t pos = base - v;
        if ((uint32_t)g.w() * stride + top > 63u) pos &= 63u;
        const uint32_t act = alive;
        uint64_t& mine = q[J & 1u];
        const uint64_t bit = 1ull << pos;
        mine = ((uint64_t)and_or((uint32_t)(bit >> 32), act, (uint32_t)(mine >> 32)) << 32) |
               and_or((uint32_t)bit, act, (uint32_t)mine);
        h4 += act << sh;
        op = (h4 >> 3) & ONES;
        bool won;
        if (g.k() == 4) {
            won = four_in_a_row_at(mine, g.h(), pos);
        } else {
            Bits<1> b;
            b.w[0] = mine;
            won = has_run(g, b);
        }
        won_any = won_any || won;   // (lane masks: scalar ORs)
        alive = (won || op == 0u) ? 0u : alive;
    };
    // the outcome of a board that has just ended, and its plies into the step count
    auto outcome_of = [&](const uint64_t (&q)[2], bool won_any) -> uint32_t {
        const uint32_t stones = (uint32_t)__popcll(q[0]) + (uint32_t)__popcll(q[1]);
        stepped += stones;
        return won_any ? ((stones - 1u) & 1u) + 1u : (uint32_t)BGS_ST_DRAW;   // no cap: a board that stopped without a run is full
    };
    // one opening ply: every column is open and nobody can win yet
    auto cheap_ply = [&](uint32_t j, uint32_t draw, uint64_t (&q)[2], uint32_t& h4) {
        const uint32_t col = sample_index(draw, (uint32_t)g.w());
        const uint32_t sh = col * 4u;
        const uint32_t v = (h4 >> sh) & 15u;
        const uint32_t pos = col * stride + top - v;
        q[j & 1u] |= 1ull << pos;
        h4 -= 1u << sh;
    };
    // one block of four plies on this lane's board from the word w[0], then boards that ended go to memory (their status
    // and reward follow from the codes); the lane's words move down one
    auto play_block = [&]() {
        const uint32_t was_live = live;
        Philox4 four = {{0u, 0u, 0u, 0u}};
        uint32_t word = 0u;
        if constexpr (PER_PLY) {
            four = philox4x32_10(seed, first_game + (uint64_t)(begin + game), myblk);
            myblk += 1u;
        } else {
            word = wnext;
            wnext = my_words[wrow * BGS_WAVE];   // (used by the NEXT block: the read has a whole block to arrive)
            wrow = wrow + 1u < (uint32_t)NWORDS - 1u ? wrow + 1u : (uint32_t)NWORDS - 1u;   // (a finished lane idles on the last row)
        }
        uint32_t open = (hts >> 3) & ONES;
        full_ply(std::integral_constant<uint32_t, 0>{}, PER_PLY ? four.v[0] : sub_draw<0>(word), p, hts, open, live, anywon);
        full_ply(std::integral_constant<uint32_t, 1>{}, PER_PLY ? four.v[1] : sub_draw<1>(word), p, hts, open, live, anywon);
        full_ply(std::integral_constant<uint32_t, 2>{}, PER_PLY ? four.v[2] : sub_draw<2>(word), p, hts, open, live, anywon);
        full_ply(std::integral_constant<uint32_t, 3>{}, PER_PLY ? four.v[3] : sub_draw<3>(word), p, hts, open, live, anywon);
        if (was_live != 0 && live == 0) {
            *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(plane0) + (game * 8u)) = p[0];
            *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(plane1) + (game * 8u)) = p[1];
            outcome[game] = (uint8_t)outcome_of(p, anywon);
        }
    };
    // an idle lane takes game `which` of the chunk out of the pool: board, column nibbles, the words of its blocks to come
    auto take = [&](uint32_t which) {
        game = which;
        const uint32_t slot = which & (Pool::SLOTS - 1u);
        p[0] = pool.plane[0][slot];
        p[1] = pool.plane[1][slot];
        hts = pool.cols[slot];
        if constexpr (PER_PLY) {
            myblk = (uint32_t)OPEN_BLOCKS;
        } else {
#pragma unroll
            for (int k = 0; k < OW::QUADS; ++k) {
                const uint4 v = pool.words[k][slot];
                if (k == 0) wnext = v.x; else my_words[(4 * k) * BGS_WAVE] = v.x;
                my_words[(4 * k + 1) * BGS_WAVE] = v.y;
                my_words[(4 * k + 2) * BGS_WAVE] = v.z;
                my_words[(4 * k + 3) * BGS_WAVE] = v.w;
            }
        }
        wrow = 1u;
        anywon = false;
        live = hts != 0u ? ~0u : 0u;  // (a dead slot: the opening stage has stored that game)
    };
    // all 64 lanes open the chunk's next 64 games (blocks 0 .. OPEN_BLOCKS - 1 in lock step) and park them with their words
    auto open_games = [&]() {
        const uint32_t og = opened + lane;
        const uint64_t id = first_game + (uint64_t)(begin + og);
        uint64_t q[2] = {0, 0};
        uint32_t h4 = top * columns, alive = ~0u, op = columns;
        bool won_any = false;
        uint32_t words[12];   // the words of blocks 0 .. 11: three philox calls (fewer when the board cannot last that long)
        // the draws of block b of the opening: sub-draws of its word, or (strict contract) the four words of its own call
        Philox4 own = philox4x32_10(seed, id, 0u);
        using J0 = std::integral_constant<uint32_t, 0>; using J1 = std::integral_constant<uint32_t, 1>;
        using J2 = std::integral_constant<uint32_t, 2>; using J3 = std::integral_constant<uint32_t, 3>;
#define BGS_DRAW_OF(JT, b) (PER_PLY ? own.v[JT::value] : sub_draw<JT::value>(words[b]))
        if constexpr (!PER_PLY) {
            words[0] = own.v[0]; words[1] = own.v[1]; words[2] = own.v[2]; words[3] = own.v[3];
#pragma unroll
            for (int c = 1; c < 3; ++c) {
                if (4 * c <= OW::LAST) {
                    const Philox4 d = philox4x32_10(seed, id, (uint32_t)c);
                    words[4 * c] = d.v[0]; words[4 * c + 1] = d.v[1]; words[4 * c + 2] = d.v[2]; words[4 * c + 3] = d.v[3];
                } else {
                    words[4 * c] = words[4 * c + 1] = words[4 * c + 2] = words[4 * c + 3] = 0u;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 12; ++c) words[c] = 0u;
        }
        cheap_ply(0u, BGS_DRAW_OF(J0, 0), q, h4);
        cheap_ply(1u, BGS_DRAW_OF(J1, 0), q, h4);
        cheap_ply(2u, BGS_DRAW_OF(J2, 0), q, h4);
        cheap_ply(3u, BGS_DRAW_OF(J3, 0), q, h4);
        if (OPEN_BLOCKS >= 2) {
            if constexpr (PER_PLY) own = philox4x32_10(seed, id, 1u);
            cheap_ply(0u, BGS_DRAW_OF(J0, 1), q, h4);
            cheap_ply(1u, BGS_DRAW_OF(J1, 1), q, h4);
            op = (h4 >> 3) & ONES;
            full_ply(J2{}, BGS_DRAW_OF(J2, 1), q, h4, op, alive, won_any);
            full_ply(J3{}, BGS_DRAW_OF(J3, 1), q, h4, op, alive, won_any);
        }
#pragma unroll
        for (int ob = 2; ob < OPEN_BLOCKS; ++ob) {  // further blocks in lock step
            if constexpr (PER_PLY) own = philox4x32_10(seed, id, (uint32_t)ob);
            full_ply(J0{}, BGS_DRAW_OF(J0, ob), q, h4, op, alive, won_any);
            full_ply(J1{}, BGS_DRAW_OF(J1, ob), q, h4, op, alive, won_any);
            full_ply(J2{}, BGS_DRAW_OF(J2, ob), q, h4, op, alive, won_any);
            full_ply(J3{}, BGS_DRAW_OF(J3, ob), q, h4, op, alive, won_any);
        }
#undef BGS_DRAW_OF
        if (OPEN_BLOCKS >= 2 && og < avail && alive == 0) {  // ended inside the opening: a win, or a small board is full
            plane0[og] = q[0];
            plane1[og] = q[1];
            outcome[og] = (uint8_t)outcome_of(q, won_any);
        }   // (a lane past the end of the chunk played for nobody: nothing of it is stored or counted)
        const uint32_t slot = og & (Pool::SLOTS - 1u);
        pool.plane[0][slot] = q[0];
        pool.plane[1][slot] = q[1];
        pool.cols[slot] = alive ? h4 : 0u;
#pragma unroll
        for (int k = 0; k < OW::QUADS; ++k) {
            auto at = [&](int i) { return OPEN_BLOCKS + i <= 11 ? words[OPEN_BLOCKS + i <= 11 ? OPEN_BLOCKS + i : 11] : 0u; };
            pool.words[k][slot] = make_uint4(at(4 * k), at(4 * k + 1), at(4 * k + 2), at(4 * k + 3));
        }
        opened += 64u;
    };

    // ---- the chunk: idle lanes take the next opened games
    while (taken < avail) {
        const uint64_t need = __builtin_amdgcn_ballot_w64(live == 0);
        if (need) {
            const uint32_t wanted = (uint32_t)__popcll(need);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t have = opened - taken;   // wave-uniform: boards waiting in the pool (<= 64)
            if (wanted > have && opened < avail) {
                // ---- the pool runs dry: the first `have` needy lanes empty it, all 64 lanes open the chunk's next 64 games
                // into the freed slots, the other needy lanes take from those
                const bool first = live == 0 && rank < have;
                const bool second = live == 0 && !first;
                if (first && taken + rank < avail) take(taken + rank);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the reads above before the stores of the opening
                __builtin_amdgcn_wave_barrier();
                open_games();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (second && taken + rank < avail) take(taken + rank);
            } else if (live == 0 && taken + rank < avail) {
                take(taken + rank);
            }
            taken = avail - taken < wanted ? avail : taken + wanted;
        }
        play_block();
    }
    // ---- the drain: no games left to hand out, the boards in flight play to their end
    while (__builtin_amdgcn_ballot_w64(live != 0)) play_block();

    // ---- the chunk is done: four games per lane -- their status dword as it lies in LDS, their reward pairs, and their
    // byte of 2-bit codes for the hand-over (64 lanes: 64 contiguous bytes per store, into host memory when mapped)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the byte stores of every lane before the reads below
    __builtin_amdgcn_wave_barrier();
    uint8_t* __restrict__ const status_out = status + begin;
    uint16_t* __restrict__ const reward_out = reward + begin;
    uint8_t* __restrict__ const packed_out = reinterpret_cast<uint8_t*>(codes_out) + (begin >> 2);
    for (uint32_t g4 = lane * 4u; g4 < avail; g4 += BGS_WAVE * 4u) {
        const uint32_t four = reinterpret_cast<const uint32_t*>(outcome)[g4 >> 2];  // (bytes past avail are 0)
        if (CODES) packed_out[g4 >> 2] = (uint8_t)((four & 3u) | ((four >> 6) & 0xCu) | ((four >> 12) & 0x30u) | ((four >> 18) & 0xC0u));
        if (g4 + 4u <= avail) {
            *reinterpret_cast<uint32_t*>(status_out + g4) = four;
            uint2 r;
            r.x = (uint32_t)reward_pair(four & 255u) | ((uint32_t)reward_pair((four >> 8) & 255u) << 16);
            r.y = (uint32_t)reward_pair((four >> 16) & 255u) | ((uint32_t)reward_pair(four >> 24) << 16);
            *reinterpret_cast<uint2*>(reward_out + g4) = r;
        } else {
            for (uint32_t k = 0; g4 + k < avail; ++k) {
                const uint32_t ck = (four >> (8u * k)) & 255u;
                status_out[g4 + k] = (uint8_t)ck;
                reward_out[g4 + k] = reward_pair(ck);
            }
        }
    }
    add_steps(steps, stepped);
}

// K2b: the block-aligned, branch-free rollout for every other geometry (multi-word planes, up to 16 columns, up to 15
// rows).  Same structure as K2a; column state is a nibble of height per column (64 bits) plus "column open" flags
// kept nibble-spread in two 32-bit halves (columns 0-7 and 8-15), so the idx-th open column is one SWAR select inside
// the half that holds it.
template <class G, bool CAPPED>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_rollout_aligned_wide(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status,
                               uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game,
                               uint32_t max_plies, unsigned long long* __restrict__ steps, uint32_t games_per_wave,
                               uint32_t per_ply) {
    constexpr int NW = G::NW;
    constexpr uint32_t ONES = 0x11111111u;
    const int h = g.h(), w = g.w();
    const uint32_t open_lo0 = w >= 8 ? ONES : (ONES & ((1u << (4 * w)) - 1u));
    const uint32_t open_hi0 = w <= 8 ? 0u : (w >= 16 ? ONES : (ONES & ((1u << (4 * (w - 8))) - 1u)));

    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    const int64_t begin = (int64_t)wave * games_per_wave;
    const int64_t end = begin + games_per_wave < n ? begin + games_per_wave : n;
    const uint32_t avail = begin < end ? (uint32_t)(end - begin) : 0u;
    uint32_t taken = 0;

    Bits<NW> p[2] = {zero_bits<NW>(), zero_bits<NW>()};  // stones of player 0 / player 1
    uint64_t hts = 0;                                      // nibble per column: its height
    uint32_t open_lo = 0, open_hi = 0;                     // nibble-spread "column open" flags
    uint32_t blk = 0, st = 0, game = 0, stepped = 0;
    uint64_t live = 0;

    if (avail == 0u) return;
    do {
        const uint64_t need = __builtin_amdgcn_ballot_w64(live == 0);
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (live == 0 && taken + rank < avail) {
                game = taken + rank;
                p[0] = zero_bits<NW>();
                p[1] = zero_bits<NW>();
                hts = 0;
                open_lo = open_lo0;
                open_hi = open_hi0;
                blk = 0;
                st = 0;
                live = (!CAPPED || max_plies > 0u) ? 1u : 0u;
                if (CAPPED && live == 0) {
                    store_planes<NW>(planes, n, begin + game, p[0], p[1]);
                    status[begin + game] = 0;
                    reward[begin + game] = 0;
                }
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }

        const uint64_t was_live = live;
        // (per_ply, wave-uniform: the strict contract -- the call's four words are the block's draws)
        const Philox4 four = philox4x32_10(seed, first_game + (uint64_t)(begin + game), per_ply ? blk : blk >> 2);
        const uint32_t word = philox_word(four, blk);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t count_lo = (uint32_t)__popc(open_lo);
            const uint32_t idx = sample_index(per_ply ? four.v[j] : sub_draw(word, j), count_lo + (uint32_t)__popc(open_hi));
            const bool in_lo = idx < count_lo;
            const uint32_t part = in_lo ? open_lo : open_hi;
            const uint32_t rank_in_part = in_lo ? idx : idx - count_lo;
            const uint32_t cmp = (rank_in_part - part) * ONES + 0x88888888u;  // nibble x: 8 + rank - (open among 0..x)
            const uint32_t col = ((uint32_t)__popc(cmp & 0x88888888u) & 7u) + (in_lo ? 0u : 8u);
            const uint32_t v = (uint32_t)(hts >> (4u * col)) & 15u;
            const uint32_t pos = col * (uint32_t)(h + 1) + v;
            Bits<NW>& mine = p[j & 1u];
            const uint64_t stone = live << (pos & 63u);
#pragma unroll
            for (int i = 0; i < NW; ++i) mine.w[i] |= ((pos >> 6) == (uint32_t)i) ? stone : 0ull;
            hts += live << (4u * col);
            const uint32_t filled = (live != 0 && v + 1u == (uint32_t)h) ? (1u << (4u * (col & 7u))) : 0u;
            open_lo &= ~(in_lo ? filled : 0u);
            open_hi &= ~(in_lo ? 0u : filled);
            const bool won = has_run(g, mine);
            stepped += (uint32_t)live;
            st = (live != 0 && won) ? (j & 1u) + 1u : st;
            live = (won || (open_lo | open_hi) == 0u) ? 0 : live;
            if (CAPPED) live = (4u * blk + j + 1u < max_plies) ? live : 0;
        }
        blk += 1u;

        if (was_live != 0 && live == 0) {
            const int64_t i = begin + game;
            const uint32_t code = st ? st : ((open_lo | open_hi) == 0u ? BGS_ST_DRAW : BGS_ST_RUNNING);
            store_planes<NW>(planes, n, i, p[0], p[1]);
            status[i] = (uint8_t)code;
            reward[i] = reward_pair(code);
        }
    } while (__builtin_amdgcn_ballot_w64(live != 0) || taken < avail);
    add_steps(steps, stepped);
}

// ------------------------------------------------------------------------------------------------
// K2c: the rollout for large boards of a compile-time geometry (Connect(12,13,5): 3 words per plane), with the boards
// staged in LDS.  The bit-planes of a lane's game live in the workgroup's LDS tile, one dword COLUMN per lane
// (dword d of lane t at [d][t]: the bank is the lane, so a per-lane dynamic dword index never conflicts).  A ply then
//   drops its stone with one ds_or_b32 into the dword that holds the cell, and
//   tests for a run only where one can have appeared: it reads the 128-bit window of the mover's plane centred on
//     the stone (5 dwords from a dynamic index, aligned with v_alignbit so the stone sits at bit R = (k-1)(H+2)),
//     runs the shift-and-AND run test on those 4 dwords in the three slanted / horizontal directions, and checks the
//     column with one field extract.
// The generic kernel K2b re-scans all three 64-bit words of the plane in four directions every ply and selects the
// word that takes the stone with compares; this one needs neither.  Zero padding around the plane (PB dwords below,
// PT above) makes the window valid at the board's edges.
// ------------------------------------------------------------------------------------------------
template <class G>
struct LdsBoard {
    static constexpr int H = G::STATIC_H, W = G::STATIC_W, K = G::STATIC_K;
    static constexpr int R = (K - 1) * (H + 2);          // reach of a run through the stone, in bits
    static constexpr bool fits = H > 0 && K >= 2 && 2 * R <= 127;
    static constexpr int PB = (R + 31) / 32;             // zero dwords below the plane
    static constexpr int PD = 2 * G::NW;                 // dwords of the plane itself
    static constexpr int ND = PB + PD + PB + 2;          // dwords per plane column (covers the window of any lane)
    static constexpr size_t lds_bytes = (size_t)2 * ND * BGS_BLOCK * sizeof(uint32_t);
};

struct Win {
    uint32_t d[4];
};

// logical right shift of the 128-bit window by the compile-time amount C
template <int C>
__device__ __forceinline__ Win win_shr(const Win& x) {
    constexpr int q = C >> 5, r = C & 31;
    Win y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (i + q < 4) ? x.d[(i + q) & 3] : 0u;
        const uint32_t hi = (i + q + 1 < 4) ? x.d[(i + q + 1) & 3] : 0u;
        y.d[i] = r ? __builtin_amdgcn_alignbit(hi, lo, r) : lo;
    }
    return y;
}

__device__ __forceinline__ Win win_and(const Win& a, const Win& b) {
    Win y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y.d[i] = a.d[i] & b.d[i];
    return y;
}

// runs of K stones with stride D anywhere in the window: run doubling, LEN = length already established in m
template <int D, int LEN, int K>
__device__ __forceinline__ Win win_runs(const Win& m) {
    if constexpr (LEN >= K) {
        return m;
    } else if constexpr (2 * LEN <= K) {
        return win_runs<D, 2 * LEN, K>(win_and(m, win_shr<LEN * D>(m)));
    } else {
        return win_and(m, win_shr<(K - LEN) * D>(m));
    }
}

// ENTRY: how a lane's next game starts -- kEntryMemory: the board in memory, at whatever ply it holds; kEntryInitial: the
// empty board; kEntryOpened: the board k_connect_open_lds left in memory, kOpenedPlies plies into the game, with its
// column heights in `opened_heights`.
enum { kEntryMemory = 0, kEntryInitial = 1, kEntryOpened = 2 };
constexpr uint32_t kOpenedBlocks = 2, kOpenedPlies = 4 * kOpenedBlocks;

// The opening of K2c.  In the first 2 K - 2 plies of a game nobody can have K stones, and no column of height >= 8 can
// fill in 8 plies: for Connect(12,13,5) the first two 4-ply blocks are "column = floor(draw * W / 2^32), drop", a dozen
// instructions a ply against ~125 for a full one.  Inside the rollout kernel that saving is lost again -- lanes start
// their games at different times, so a wave would run the cheap code AND the full code every iteration -- and a pool of
// pre-opened boards (K2o's way) would double the LDS tile and halve the occupancy.  So the opening is its own launch:
// every lane opens one game, all lanes busy, boards and column heights go to memory, and the rollout kernel's lanes
// pick them up from there: one 64-byte record a game (2 x 3 plane words, the column heights), i.e. one cache line, four
// 16-byte loads (the batch's own plane arrays would be six lines a game).
struct OpenedBoard {
    uint64_t w[8];  // [0, 2 NW): the planes of player 0, then player 1; [6]: column heights, a nibble each; [7]: unused
};
static_assert(sizeof(OpenedBoard) == 64, "one cache line");

template <class G, bool PER_PLY>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_open_lds(G g, OpenedBoard* __restrict__ opened, int64_t n, uint64_t seed, uint64_t first_game,
                   unsigned long long* __restrict__ steps) {
    constexpr int NW = G::NW, H = G::STATIC_H, W = G::STATIC_W;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        Bits<NW> p[2] = {zero_bits<NW>(), zero_bits<NW>()};
        uint64_t hts = 0;
        static_assert(kOpenedBlocks <= 4, "one philox call covers four blocks");
        const Philox4 words = philox4x32_10(seed, first_game + (uint64_t)i, 0u);   // the words of blocks 0 .. 3
#pragma unroll
        for (uint32_t blk = 0; blk < kOpenedBlocks; ++blk) {
            Philox4 own = words;   // the strict contract: a call per block, its four words are the block's draws
            if (PER_PLY && blk > 0u) own = philox4x32_10(seed, first_game + (uint64_t)i, blk);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t col = sample_index(PER_PLY ? own.v[j] : sub_draw(words.v[blk], j), (uint32_t)W);   // every column is open
                const uint32_t v = (uint32_t)(hts >> (4u * col)) & 15u;
                set_bit(p[j & 1u], (int)(col * (uint32_t)(H + 1) + v));
                hts += 1ull << (4u * col);
            }
        }
        static_assert(2 * NW <= 6, "the record holds at most three words a plane");
        uint4* out = reinterpret_cast<uint4*>(opened + i);
        uint64_t w[8] = {0, 0, 0, 0, 0, 0, hts, 0};
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            w[k] = p[0].w[k];
            w[NW + k] = p[1].w[k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[q] = make_uint4((uint32_t)w[2 * q], (uint32_t)(w[2 * q] >> 32), (uint32_t)w[2 * q + 1], (uint32_t)(w[2 * q + 1] >> 32));
        stepped = kOpenedPlies;
    }
    add_steps(steps, stepped);
}

template <class G, bool CAPPED, bool CODES, int ENTRY, bool PER_PLY>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_rollout_lds(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward,
                      int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
                      unsigned long long* __restrict__ steps, uint32_t games_per_wave, uint32_t* __restrict__ codes_out,
                      const OpenedBoard* __restrict__ opened) {
    constexpr bool FROM_INITIAL = ENTRY != kEntryMemory;   // (no plies to sit out in a game's first block)
    using L = LdsBoard<G>;
    constexpr int NW = G::NW, H = L::H, W = L::W, K = L::K, R = L::R, PB = L::PB, ND = L::ND;
    constexpr uint32_t ONES = 0x11111111u;
    constexpr uint32_t open_lo0 = W >= 8 ? ONES : (ONES & ((1u << (4 * (W & 7))) - 1u));
    constexpr uint32_t open_hi0 = W <= 8 ? 0u : (W >= 16 ? ONES : (ONES & ((1u << (4 * ((W - 8) & 7))) - 1u)));
    extern __shared__ uint32_t lds_tile[];   // [2 players][ND dwords][256 lanes], then the CODES slices
    uint32_t* const column = lds_tile + threadIdx.x;   // this lane's dword column
    auto cell = [&](int player, uint32_t dword) -> uint32_t& { return column[(player * ND + dword) * BGS_BLOCK]; };

    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BGS_BLOCK / BGS_WAVE) + (threadIdx.x >> 6));
    const int64_t begin = (int64_t)wave * games_per_wave;
    const int64_t end = begin + games_per_wave < n ? begin + games_per_wave : n;
    const uint32_t avail = begin < end ? (uint32_t)(end - begin) : 0u;
    uint32_t taken = 0;

    uint64_t hts = 0;                     // nibble per column: its height
    uint32_t open_lo = 0, open_hi = 0;    // nibble-spread "column open" flags (columns 0-7 / 8-15)
    uint32_t blk = 0, st = 0, game = 0, stepped = 0;
    uint32_t skip = 0;                    // loaded boards: sub-steps to sit out in the first block (= plies already in it)
    uint32_t live = 0;                    // all ones while this lane's game is running

    if (avail == 0u) return;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int d = 0; d < ND; ++d) cell(q, d) = 0u;   // padding (and planes) start out empty
    WaveCodes codes;
    if (CODES) codes.init(lds_tile + 2 * ND * BGS_BLOCK, games_per_wave, avail);
    // kEntryOpened: a lane's next board is REQUESTED (four 16-byte loads into registers) at the end of the iteration in
    // which its game ended, and installed in LDS at the top of the next one: the loads have a whole 4-ply block to arrive
    // in, instead of stalling the wave inside the refill.
    uint4 pre[4];
    pre[0] = pre[1] = pre[2] = pre[3] = make_uint4(0, 0, 0, 0);
    uint32_t next_game = 0;
    bool fetched = false;
    auto request = [&]() {
        if constexpr (ENTRY == kEntryOpened) {
            const uint64_t need = __builtin_amdgcn_ballot_w64(live == 0 && !fetched);
            if (need && taken < avail) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (live == 0 && !fetched && taken + rank < avail) {
                    next_game = taken + rank;
                    const uint4* rec = reinterpret_cast<const uint4*>(opened + (begin + next_game));
#pragma unroll
                    for (int q = 0; q < 4; ++q) pre[q] = rec[q];
                    fetched = true;
                }
                const uint32_t wanted = (uint32_t)__popcll(need);
                taken = avail - taken < wanted ? avail : taken + wanted;
            }
        }
    };
    request();
    do {
        if constexpr (ENTRY == kEntryOpened) {
            if (live == 0 && fetched) {
                const uint32_t dw[16] = {pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w,
                                         pre[2].x, pre[2].y, pre[2].z, pre[2].w, pre[3].x, pre[3].y, pre[3].z, pre[3].w};
#pragma unroll
                for (int k = 0; k < 2 * NW; ++k) {
                    cell(0, PB + k) = dw[k];
                    cell(1, PB + k) = dw[2 * NW + k];
                }
                hts = ((uint64_t)dw[13] << 32) | dw[12];
                game = next_game;
                st = 0;
                open_lo = open_lo0;
                open_hi = open_hi0;
                blk = kOpenedBlocks;
                live = ~0u;
                fetched = false;
            }
        }
        const uint64_t need = ENTRY == kEntryOpened ? 0ull : __builtin_amdgcn_ballot_w64(live == 0);
        if (need && taken < avail) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (live == 0 && taken + rank < avail) {
                game = taken + rank;
                st = 0;
                if (ENTRY == kEntryInitial) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int d = 0; d < 2 * NW; ++d) cell(q, PB + d) = 0u;
                    hts = 0;
                    open_lo = open_lo0;
                    open_hi = open_hi0;
                    blk = 0;
                    live = (!CAPPED || max_plies > 0u) ? ~0u : 0u;
                    if (CAPPED && live == 0) {
                        const Bits<NW> none = zero_bits<NW>();
                        store_planes<NW>(planes, n, begin + game, none, none);
                        status[begin + game] = 0;
                        reward[begin + game] = 0;
                    }
                } else {
                    // a board from memory joins at its own ply (see k_connect_rollout_aligned): planes into the LDS
                    // column, column heights from the occupied cells, the first (ply & 3) sub-steps of its first block
                    // are sat out
                    Bits<NW> p0, p1;
                    load_planes<NW>(planes, n, begin + game, p0, p1);
                    uint32_t ply0 = 0;
#pragma unroll
                    for (int wd = 0; wd < NW; ++wd) {
                        cell(0, PB + 2 * wd) = (uint32_t)p0.w[wd];
                        cell(0, PB + 2 * wd + 1) = (uint32_t)(p0.w[wd] >> 32);
                        cell(1, PB + 2 * wd) = (uint32_t)p1.w[wd];
                        cell(1, PB + 2 * wd + 1) = (uint32_t)(p1.w[wd] >> 32);
                        ply0 += (uint32_t)__popcll(p0.w[wd]) + (uint32_t)__popcll(p1.w[wd]);
                    }
                    const Bits<NW> occ = p0 | p1;
                    hts = 0;
                    open_lo = 0;
                    open_hi = 0;
#pragma unroll
                    for (int x = 0; x < W; ++x) {
                        // the column's H cells start at bit x (H + 1): at most two words hold them
                        constexpr uint64_t field = (1ull << H) - 1ull;
                        const int lo_bit = x * (H + 1), wd = lo_bit >> 6, sh = lo_bit & 63;
                        uint64_t bits = occ.w[wd] >> sh;
                        if (sh + H > 64 && wd + 1 < NW) bits |= occ.w[wd + 1] << (64 - sh);
                        const uint32_t hx = (uint32_t)__popcll(bits & field);
                        hts |= (uint64_t)hx << (4 * x);
                        const uint32_t flag = hx < (uint32_t)H ? (1u << (4 * (x & 7))) : 0u;
                        if (x < 8) open_lo |= flag;
                        else open_hi |= flag;
                    }
                    blk = ply0 >> 2;
                    skip = ply0 & 3u;
                    const uint32_t st0 = status[begin + game];
                    live = (st0 == BGS_ST_RUNNING && (!CAPPED || ply0 < max_plies)) ? ~0u : 0u;
                    if (CODES && live == 0) codes.add(game, st0);  // a board that does not play keeps its outcome
                }
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            taken = avail - taken < wanted ? avail : taken + wanted;
        }

        const uint32_t was_live = live;
        const BlockDraws<PER_PLY> draws(seed, first_game + (uint64_t)(begin + game), blk);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t count_lo = (uint32_t)__popc(open_lo);
            const uint32_t idx = sample_index(draws.draw(j), count_lo + (uint32_t)__popc(open_hi));
            const bool in_lo = idx < count_lo;
            const uint32_t part = in_lo ? open_lo : open_hi;
            const uint32_t rank_in_part = in_lo ? idx : idx - count_lo;
            const uint32_t cmp = (rank_in_part - part) * ONES + 0x88888888u;  // nibble x: 8 + rank - (open among 0..x)
            const uint32_t col = ((uint32_t)__popc(cmp & 0x88888888u) & 7u) + (in_lo ? 0u : 8u);
            const uint32_t v = (uint32_t)(hts >> (4u * col)) & 15u;
            const uint32_t at = col * (uint32_t)(H + 1) + v + 32u * PB;   // the cell's bit index in the padded column
            const int me = (int)(j & 1u);
            const uint32_t act = (FROM_INITIAL || j >= skip) ? live : 0u;  // mask: this lane plays this sub-step
            // the stone: one LDS OR into the dword that holds the cell (nothing for a lane that is not playing)
            atomicOr(&cell(me, at >> 5), act & (1u << (at & 31u)));
            hts += (uint64_t)(act & 1u) << (4u * col);
            const uint32_t filled = (act != 0 && v + 1u == (uint32_t)H) ? (1u << (4u * (col & 7u))) : 0u;
            open_lo &= ~(in_lo ? filled : 0u);
            open_hi &= ~(in_lo ? 0u : filled);
            // the 128-bit window of the mover's plane that starts R bits below the stone
            const uint32_t from = at - (uint32_t)R, first = from >> 5, shift = from & 31u;
            uint32_t y[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) y[i] = cell(me, first + i);
            Win x;
#pragma unroll
            for (int i = 0; i < 4; ++i) x.d[i] = __builtin_amdgcn_alignbit(y[i + 1], y[i], shift);
            const Win r1 = win_runs<H + 1, 1, K>(x), r2 = win_runs<H + 2, 1, K>(x), r3 = win_runs<H, 1, K>(x);
            // r marks where a run STARTS, and a run through the new stone (window bit R) starts at or below it: only the
            // dwords that hold bits 0 .. R are looked at (a run that starts above the stone would have ended the game
            // earlier), and the compiler drops what only the others needed -- the last doubling step shrinks from four
            // dwords to R / 32 + 1, the one before to one more
            constexpr int START_DWORDS = R / 32 + 1;
            uint32_t any_run = 0;
#pragma unroll
            for (int i = 0; i < START_DWORDS && i < 4; ++i) any_run |= r1.d[i] | r2.d[i] | r3.d[i];
            // the column: the stone (window bit R) and the K - 1 cells below it
            constexpr int lowest = R - (K - 1);
            constexpr uint32_t kmask = (1u << K) - 1u;
            const uint64_t pair = ((uint64_t)x.d[(lowest >> 5) + 1] << 32) | x.d[lowest >> 5];
            const uint32_t below = (uint32_t)(pair >> (lowest & 31)) & kmask;
            const bool won = any_run != 0u || below == kmask;
            stepped -= act;
            st = won ? (j & 1u) + 1u : st;   // (a lane that is not playing re-finds at most its own finished game's run)
            live = (won || (open_lo | open_hi) == 0u) ? 0u : live;
            if (CAPPED) live = (4u * blk + j + 1u < max_plies) ? live : 0u;
        }
        blk += 1u;
        skip = 0;

        if (was_live != 0 && live == 0) {
            const int64_t i = begin + game;
            const uint32_t code = st ? st : ((open_lo | open_hi) == 0u ? BGS_ST_DRAW : BGS_ST_RUNNING);
            Bits<NW> p0, p1;
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) {
                p0.w[wd] = ((uint64_t)cell(0, PB + 2 * wd + 1) << 32) | cell(0, PB + 2 * wd);
                p1.w[wd] = ((uint64_t)cell(1, PB + 2 * wd + 1) << 32) | cell(1, PB + 2 * wd);
            }
            store_planes<NW>(planes, n, i, p0, p1);
            status[i] = (uint8_t)code;
            reward[i] = reward_pair(code);
            if (CODES) codes.add(game, code);
        }
        request();
    } while (__builtin_amdgcn_ballot_w64(live != 0 || fetched) || taken < avail);
    if (CODES) codes.flush(codes_out, begin);
    add_steps(steps, stepped);
}

// K4: packed planes -> reference layout int8[n][H][W] (row 0 = bottom; -1 empty, 0, 1).  One lane expands one board
// into the workgroup's LDS tile (256 boards x H*W bytes, already in output order); the workgroup then streams the
// tile to HBM with 16-byte stores.
template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_unpack(G g, const uint64_t* __restrict__ planes, int64_t n, int8_t* __restrict__ grid) {
    constexpr int NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    const int h = g.h(), w = g.w(), hw = h * w;
    const int64_t base = (int64_t)blockIdx.x * BGS_BLOCK;
    const int64_t i = base + threadIdx.x;
    if (i < n) {
        Bits<NW> p0, p1;
        load_planes<NW>(planes, n, i, p0, p1);
        uint8_t* mine = tile + threadIdx.x * hw;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int bit = x * (h + 1) + y;
                const uint32_t b0 = test_bit(p0, bit), b1 = test_bit(p1, bit);
                mine[y * w + x] = (uint8_t)(b0 + 2u * b1 - 1u);  // 0xFF empty, 0 player 0, 1 player 1
            }
    }
    __syncthreads();
    const int64_t boards = n - base < BGS_BLOCK ? n - base : BGS_BLOCK;
    tile_to_global(tile, reinterpret_cast<uint8_t*>(grid) + base * hw, (uint32_t)(boards * hw));
}

// Wire format of the asynchronous grid hand-over (bgs_host.hip): per board two bit sets over the cells in the
// REFERENCE order (cell = y * W + x, row 0 = bottom), "occupied" and "player 1's", NWC = ceil(H W / 64) words each,
// stored plane-major over the batch like the boards themselves: word j of set q at dst[(q * NWC + j) * n + i].  The host
// turns a word into 64 grid bytes with two masked byte adds (-1 + occupied + player 1's), 16 B per 6x7 board cross PCIe
// instead of 42.
template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_cell_planes(G g, const uint64_t* __restrict__ planes, int64_t n, uint64_t* __restrict__ dst, int nwc) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    Bits<NW> p0, p1;
    load_planes<NW>(planes, n, i, p0, p1);
    const int h = g.h(), w = g.w();
    uint64_t occ = 0, who = 0;
    int cell = 0, word = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int bit = x * (h + 1) + y;
            const uint64_t b0 = test_bit(p0, bit), b1 = test_bit(p1, bit);
            occ |= (b0 | b1) << (cell & 63);
            who |= b1 << (cell & 63);
            if ((++cell & 63) == 0) {
                dst[(int64_t)word * n + i] = occ;
                dst[(int64_t)(nwc + word) * n + i] = who;
                occ = who = 0;
                ++word;
            }
        }
    if (cell & 63) {
        dst[(int64_t)word * n + i] = occ;
        dst[(int64_t)(nwc + word) * n + i] = who;
    }
}

__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_meta(ConnectGeom cg, const uint64_t* __restrict__ planes, const uint8_t* __restrict__ status, int64_t n,
               int8_t* __restrict__ player, uint8_t* __restrict__ ended, int8_t* __restrict__ winner,
               int32_t* __restrict__ plies) {
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t count = 0;
    for (int j = 0; j < 2 * cg.nw; ++j) count += (uint32_t)__popcll(planes[(int64_t)j * n + i]);
    const uint32_t st = status[i];
    if (player) player[i] = (int8_t)(count & 1u);
    if (ended) ended[i] = st != 0;
    if (winner) winner[i] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
    if (plies) plies[i] = (int32_t)count;
}

template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_legal(G g, const uint64_t* __restrict__ planes, const uint8_t* __restrict__ status, int64_t n,
                uint8_t* __restrict__ legal, int32_t* __restrict__ count) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    Bits<NW> p0, p1;
    load_planes<NW>(planes, n, i, p0, p1);
    const Lane<NW> l = make_lane(g, p0, p1);
    const uint32_t open = status[i] == BGS_ST_RUNNING ? (~l.full & g.all_columns()) : 0u;
    if (legal)
        for (int x = 0; x < g.w(); ++x) legal[i * g.w() + x] = (open >> x) & 1u;
    if (count) count[i] = __popc(open);
}

// One-workgroup kernels whose records go to host memory the device addresses: after every thread's stores, thread 0
// publishes the call's ticket behind them (release at system scope) -- the host reads the records as soon as it sees
// the ticket, without waiting for the stream's completion signal.
__device__ __forceinline__ void publish_ticket(uint32_t* done, uint32_t ticket) {
    if (!done) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The object API's round trip on a small batch (bgs_transition, n <= 64): the chosen move, if any, and then everything a
// State shows -- grid, player, winner, plies, open columns, reward pair -- in ONE launch instead of five (step, unpack,
// meta, legal, reward copy).  A thread owns a board; the outputs are plain per-board records (the caller passes host
// memory the device addresses: a few hundred bytes over PCIe).  Same device functions as the kernels it stands in for.
template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_transition(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward, int64_t n,
                     const int32_t* __restrict__ actions, int32_t* __restrict__ result, unsigned long long* __restrict__ steps,
                     int8_t* __restrict__ grid, int8_t* __restrict__ player, int8_t* __restrict__ winner,
                     int32_t* __restrict__ plies, uint8_t* __restrict__ legal, uint16_t* __restrict__ reward_out,
                     uint32_t* __restrict__ done, uint32_t ticket) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        Bits<NW> p0, p1;
        load_planes<NW>(planes, n, i, p0, p1);
        uint32_t st = status[i];
        uint16_t pair = reward[i];
        if (actions) {
            const int col = actions[i];
            int32_t rc = 0;
            if (col >= 0) {
                rc = -2;  // BGS_ERR_ILLEGAL
                if (st == BGS_ST_RUNNING) {
                    Lane<NW> l = make_lane(g, p0, p1);
                    uint32_t after = 0;
                    if (play_column(g, l, col, after)) {
                        lane_planes(l, p0, p1);
                        store_planes<NW>(planes, n, i, p0, p1);
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
        const int h = g.h(), w = g.w();
        int8_t* cells = grid + i * h * w;
        uint32_t count = 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int bit = x * (h + 1) + y;
                const uint32_t b0 = test_bit(p0, bit), b1 = test_bit(p1, bit);
                cells[y * w + x] = (int8_t)(b0 + 2u * b1 - 1u);  // -1 empty, 0 player 0, 1 player 1
                count += b0 + b1;
            }
        player[i] = (int8_t)(count & 1u);
        winner[i] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
        plies[i] = (int32_t)count;
        const Lane<NW> l = make_lane(g, p0, p1);
        const uint32_t open = st == BGS_ST_RUNNING ? (~l.full & g.all_columns()) : 0u;
        for (int x = 0; x < w; ++x) legal[i * w + x] = (open >> x) & 1u;
        reward_out[i] = pair;
    }
    add_steps(steps, stepped);
    publish_ticket(done, ticket);
}

// reference layout -> packed planes, with validation (gravity, stone counts, cell codes)
template <class G>
__global__ void __launch_bounds__(BGS_BLOCK)
k_connect_pack(G g, uint64_t* __restrict__ planes, uint8_t* __restrict__ status, uint16_t* __restrict__ reward, int64_t n,
               const int8_t* __restrict__ grid, const int8_t* __restrict__ player, const int8_t* __restrict__ winner,
               int32_t* __restrict__ result) {
    constexpr int NW = G::NW;
    const int64_t i = (int64_t)blockIdx.x * BGS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int h = g.h(), w = g.w();
    const int8_t* cells = grid + i * h * w;
    Bits<NW> p0 = zero_bits<NW>(), p1 = zero_bits<NW>();
    bool ok = true;
    int c0 = 0, c1 = 0;
    for (int x = 0; x < w; ++x) {
        bool open = false;  // an empty cell was seen lower in this column
        for (int y = 0; y < h; ++y) {
            const int v = cells[y * w + x];
            if (v == -1) {
                open = true;
            } else if (v == 0 || v == 1) {
                if (open) ok = false;
                if (v == 0) { set_bit(p0, x * (h + 1) + y); ++c0; }
                else { set_bit(p1, x * (h + 1) + y); ++c1; }
            } else {
                ok = false;
            }
        }
    }
    if (!(c0 == c1 || c0 == c1 + 1)) ok = false;
    if (player && ok && player[i] != ((c0 + c1) & 1)) ok = false;
    // the status the grid itself implies: a k-run belongs to whoever moved last (a game stops at its first run, so
    // a run of the side to move, or runs of both sides, cannot arise), a full board without a run is a draw
    const bool run0 = has_run(g, p0), run1 = has_run(g, p1);
    if ((run0 && run1) || (run0 && c0 != c1 + 1) || (run1 && c0 != c1)) ok = false;
    uint32_t st = run0 ? 1u : (run1 ? 2u : (c0 + c1 == h * w ? BGS_ST_DRAW : BGS_ST_RUNNING));
    if (winner) {
        // an explicit winner has to agree with the grid (winner = -1 on a board that holds a k-run would make the
        // rollout kernels treat the board differently from one another)
        const int wv = winner[i];
        if (wv < -1 || wv > 2) ok = false;
        else if ((wv == -1 ? 0u : (wv == 2 ? BGS_ST_DRAW : (uint32_t)(wv + 1))) != st) ok = false;
    }
    if (ok) {
        store_planes<NW>(planes, n, i, p0, p1);
        status[i] = (uint8_t)st;
        reward[i] = reward_pair(st);
    }
    if (result) result[i] = ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// dispatch: static instantiations for the BASELINE geometries, run-time geometry otherwise
// ------------------------------------------------------------------------------------------------
inline unsigned grid_for(int64_t n) { return (unsigned)((n + BGS_BLOCK - 1) / BGS_BLOCK); }

template <class T>
struct Tag {
    using type = T;
};

template <class F>
void dispatch(const ConnectGeom& cg, F&& f) {
    if (cg.h == 6 && cg.w == 7 && cg.k == 4) { f(Geo<1, 6, 7, 4>{6, 7, 4}); return; }
    if (cg.h == 12 && cg.w == 13 && cg.k == 5) { f(Geo<3, 12, 13, 5>{12, 13, 5}); return; }
    switch (cg.nw) {
        case 1: f(Geo<1, 0, 0, 0>{cg.h, cg.w, cg.k}); return;
        case 2: f(Geo<2, 0, 0, 0>{cg.h, cg.w, cg.k}); return;
        default: f(Geo<3, 0, 0, 0>{cg.h, cg.w, cg.k}); return;
    }
}

}  // namespace

void connect_reset(const bgs_batch* b) {
    hipLaunchKernelGGL(k_connect_reset, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->d_planes, b->d_status,
                       reinterpret_cast<uint16_t*>(b->d_reward), b->n, b->planes, 0);
}

void connect_reset_ended(const bgs_batch* b) {
    hipLaunchKernelGGL(k_connect_reset, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->d_planes, b->d_status,
                       reinterpret_cast<uint16_t*>(b->d_reward), b->n, b->planes, 1);
}

// geometry + board policy: nibble column state where it applies (one word, W <= 8, H <= 8), generic otherwise
template <class F>
void dispatch_game(const ConnectGeom& cg, F&& f) {
    const bool nibble_ok = cg.nw == 1 && cg.w <= 8 && cg.h <= 8;
    dispatch(cg, [&](auto g) {
        using G = decltype(g);
        if constexpr (G::NW == 1) {
            if (nibble_ok) { f(g, Tag<NibbleGame<G>>{}); return; }
        }
        f(g, Tag<GenericGame<G>>{});
    });
}

void connect_step_random(const bgs_batch* b, uint64_t seed, uint32_t count) {
    // one-word boards, even batch: the streaming kernel (pairs of boards per lane, `count` plies per launch)
    if (b->cg.nw == 1 && (b->n & 1) == 0 && !b->rollout_generic) {
        const int64_t pairs = b->n >> 1;
        int64_t blocks = (pairs + BGS_BLOCK - 1) / BGS_BLOCK;
        // 16 workgroups of 4 waves per CU: twice what is resident, so a CU that finishes its share early takes more
        // (round 3, r3_k1.sh in the git history at 2^24 boards, one ply: 4 / 6 / 8 / 12 / 16 / 32 per CU = 4.35 / 4.68 / 4.75 / 4.96 / 4.94 / 4.83
        // TB/s; with the non-temporal accesses 16 per CU reads 5.04)
        const int64_t resident = (int64_t)b->num_cus * 16;
        if (blocks > resident) blocks = resident;
        dispatch(b->cg, [&](auto g) {
            using G = decltype(g);
            if constexpr (G::NW == 1) {
                auto launch = [&](auto single_tag, auto rng_tag) {
                    constexpr bool SINGLE = decltype(single_tag)::value, PER_PLY = decltype(rng_tag)::value;
                    hipLaunchKernelGGL((k_connect_step_random_stream<G, SINGLE, PER_PLY>), dim3((unsigned)blocks), dim3(BGS_BLOCK), 0,
                                       b->stream, g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n,
                                       seed, b->first_game, b->d_steps, count);
                };
                if (count == 1u) {
                    if (b->rng_per_ply) launch(std::true_type{}, std::true_type{});
                    else launch(std::true_type{}, std::false_type{});
                } else {
                    if (b->rng_per_ply) launch(std::false_type{}, std::true_type{});
                    else launch(std::false_type{}, std::false_type{});
                }
            }
        });
        return;
    }
    for (uint32_t q = 0; q < count; ++q)
    dispatch_game(b->cg, [&](auto g, auto game_tag) {
        using G = decltype(g);
        using Game = typename decltype(game_tag)::type;
        hipLaunchKernelGGL((k_connect_step_random<G, Game>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g,
                           b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game,
                           b->d_steps, b->rng_per_ply ? 1u : 0u);
    });
}

void connect_step_actions(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out) {
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_step_actions<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g, b->d_planes,
                           b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_actions, d_status_out, b->d_steps);
    });
}

// true: the fused kernel ran (one-word board, even batch, 16-byte aligned legal destination); false: nothing was enqueued
bool connect_step_observe(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out, uint8_t* d_legal, uint8_t* d_ended,
                          int8_t* d_reward_out, bool auto_reset) {
    if (b->cg.nw != 1 || (b->n & 1) || b->n < 2 || ((uintptr_t)d_legal & 15u) || ((uintptr_t)d_actions & 7u) ||
        ((uintptr_t)d_status_out & 7u) || ((uintptr_t)d_ended & 1u) || ((uintptr_t)d_reward_out & 3u))
        return false;
    const size_t tile = (size_t)2 * BGS_BLOCK * b->cg.w;
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        if constexpr (G::NW == 1) {
            hipLaunchKernelGGL((k_connect_step_observe<G>), dim3(grid_for(b->n >> 1)), dim3(BGS_BLOCK), tile, b->stream, g, b->d_planes,
                               b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_actions, d_status_out, d_legal, d_ended,
                               b->d_steps, reinterpret_cast<uint16_t*>(d_reward_out), auto_reset ? 1u : 0u);
        }
    });
    return true;
}

void status_to_ended(const bgs_batch* b, uint8_t* d_ended) {
    hipLaunchKernelGGL(k_status_to_ended, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->d_status, b->n, d_ended);
}

// codes_out != nullptr asks the kernel to deliver the 2-bit outcome codes itself (uint32[(n + 15) / 16], device or
// device-mapped host memory); returns whether the kernel chosen for this batch does so (otherwise the caller runs
// k_pack_outcomes behind it)
bool connect_rollout(const bgs_batch* b, uint64_t seed, int32_t max_plies, uint32_t flags, uint32_t* codes_out) {
    const uint32_t cap = max_plies < 0 ? 0u : (uint32_t)max_plies;
    // the RNG contract of this call: the batch's (bgs_set_rng_contract), or the strict one by flag (BGS_ROLLOUT_DRAW_PER_PLY)
    const bool per_ply = b->rng_per_ply || (flags & 4u);
    auto with_rng = [&](auto&& f) {
        if (per_ply) f(std::true_type{});
        else f(std::false_type{});
    };
    // resident waves: CUs x 4 SIMDs x waves per SIMD; every wave gets an equal contiguous chunk of games, a multiple
    // of 64 (whole dwords of outcome codes per wave, whole refill rounds)
    const int64_t resident = (int64_t)b->num_cus * 4 * b->rollout_wps;
    int64_t per_wave = b->rollout_chunk > 0 ? b->rollout_chunk : (b->n + resident - 1) / resident;
    per_wave = (per_wave + BGS_WAVE - 1) / BGS_WAVE * BGS_WAVE;
    const int64_t waves = (b->n + per_wave - 1) / per_wave;
    // the wave-private LDS slices of the fused codes: 4 waves x per_wave / 16 dwords per workgroup
    const size_t code_lds = (size_t)4 * (per_wave / 16) * sizeof(uint32_t);
    const bool fuse_codes = codes_out != nullptr && code_lds <= (32u << 10);
    bool fused = false;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const bool nibble_ok = b->cg.nw == 1 && b->cg.w <= 8 && b->cg.h <= 8;
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        auto launch = [&](auto game_tag, auto initial_tag, auto capped_tag) {
            using Game = typename decltype(game_tag)::type;
            constexpr bool INITIAL = decltype(initial_tag)::value;
            constexpr bool CAPPED = decltype(capped_tag)::value;
            hipLaunchKernelGGL((k_connect_rollout<G, Game, INITIAL, CAPPED>), dim3(blocks), dim3(BGS_BLOCK), 0, b->stream,
                               g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed,
                               b->first_game, cap, b->d_steps, (uint32_t)per_wave, per_ply ? 1u : 0u);
        };
        // a cap of height * width plies or more can never bind: drop the per-ply test
        const bool capped = cap < (uint32_t)(b->cg.h * b->cg.w);
        auto with_game = [&](auto game_tag) {
            if (flags & 1u) {
                if (capped) launch(game_tag, std::true_type{}, std::true_type{});
                else launch(game_tag, std::true_type{}, std::false_type{});
            } else {
                if (capped) launch(game_tag, std::false_type{}, std::true_type{});
                else launch(game_tag, std::false_type{}, std::false_type{});
            }
        };
        if constexpr (G::NW == 1) {
            if (nibble_ok && !b->rollout_generic) {
                // one-word boards: the block-aligned, branch-free kernel (from the initial state or from memory)
                auto launch_aligned = [&](auto capped_tag, auto initial_tag) {
                    constexpr bool CAPPED = decltype(capped_tag)::value;
                    constexpr bool INITIAL = decltype(initial_tag)::value;
                    with_rng([&](auto rng_tag) {
                        constexpr bool PER_PLY = decltype(rng_tag)::value;
                        if (fuse_codes) {
                            hipLaunchKernelGGL((k_connect_rollout_aligned<G, CAPPED, INITIAL, true, PER_PLY>), dim3(blocks), dim3(BGS_BLOCK),
                                               code_lds, b->stream, g, b->d_planes, b->d_status,
                                               reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game, cap, b->d_steps,
                                               (uint32_t)per_wave, codes_out);
                            fused = true;
                        } else {
                            hipLaunchKernelGGL((k_connect_rollout_aligned<G, CAPPED, INITIAL, false, PER_PLY>), dim3(blocks), dim3(BGS_BLOCK),
                                               0, b->stream, g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward),
                                               b->n, seed, b->first_game, cap, b->d_steps, (uint32_t)per_wave, nullptr);
                        }
                    });
                };
                const size_t outcome_lds = (size_t)4 * per_wave;  // K2o: one outcome byte per game
                // (boards of at most 48 cells: a game is at most 12 blocks of four plies, whose words three philox calls give)
                if ((flags & 1u) && !capped && b->rollout_opening && b->cg.h >= 4 && b->cg.w >= 2 && b->cg.k >= 3 &&
                    b->cg.h * b->cg.w <= 48 && outcome_lds <= (32u << 10)) {
                    // from the initial state, no cap: the kernel with the lock-step opening stage (K2o)
                    auto launch_opened = [&](auto blocks_tag, auto codes_tag) {
                        constexpr int OPEN_BLOCKS = decltype(blocks_tag)::value;
                        constexpr bool CODES = decltype(codes_tag)::value;
                        with_rng([&](auto rng_tag) {
                            constexpr bool PER_PLY = decltype(rng_tag)::value;
                            hipLaunchKernelGGL((k_connect_rollout_opened<G, OPEN_BLOCKS, CODES, PER_PLY>), dim3(blocks), dim3(BGS_BLOCK),
                                               outcome_lds, b->stream, g, b->d_planes, b->d_status,
                                               reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game, b->d_steps,
                                               (uint32_t)per_wave, CODES ? codes_out : nullptr);
                        });
                        fused = CODES;
                    };
                    const bool deep = b->cg.k >= 4 && b->cg.h >= 6;
                    auto with_blocks = [&](auto blocks_tag) {
                        if (fuse_codes) launch_opened(blocks_tag, std::true_type{});
                        else launch_opened(blocks_tag, std::false_type{});
                    };
                    switch (deep ? b->rollout_opening : 1) {
                        case 1: with_blocks(std::integral_constant<int, 1>{}); break;
                        case 2: with_blocks(std::integral_constant<int, 2>{}); break;
                        case 3: with_blocks(std::integral_constant<int, 3>{}); break;
                        default: with_blocks(std::integral_constant<int, 4>{}); break;
                    }
                    return;
                }
                if (flags & 1u) {
                    if (capped) launch_aligned(std::true_type{}, std::true_type{});
                    else launch_aligned(std::false_type{}, std::true_type{});
                } else {
                    if (capped) launch_aligned(std::true_type{}, std::false_type{});
                    else launch_aligned(std::false_type{}, std::false_type{});
                }
                return;
            }
            if (nibble_ok) { with_game(Tag<NibbleGame<G>>{}); return; }
        }
        if constexpr (LdsBoard<G>::fits && G::NW > 1) {
            if (!b->rollout_generic && !b->rollout_no_lds) {
                // a compile-time multi-word geometry: boards staged in LDS, run test on the window around the stone --
                // from the initial state or from memory, with or without a ply cap
                const size_t tile = LdsBoard<G>::lds_bytes;
                const OpenedBoard* heights = reinterpret_cast<const OpenedBoard*>(b->d_staging);
                auto launch_lds = [&](auto capped_tag, auto entry_tag) {
                    constexpr bool CAPPED = decltype(capped_tag)::value;
                    constexpr int ENTRY = decltype(entry_tag)::value;
                    with_rng([&](auto rng_tag) {
                        constexpr bool PER_PLY = decltype(rng_tag)::value;
                        if (fuse_codes) {
                            hipLaunchKernelGGL((k_connect_rollout_lds<G, CAPPED, true, ENTRY, PER_PLY>), dim3(blocks), dim3(BGS_BLOCK),
                                               tile + code_lds, b->stream, g, b->d_planes, b->d_status,
                                               reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed, b->first_game, cap, b->d_steps,
                                               (uint32_t)per_wave, codes_out, heights);
                            fused = true;
                        } else {
                            hipLaunchKernelGGL((k_connect_rollout_lds<G, CAPPED, false, ENTRY, PER_PLY>), dim3(blocks), dim3(BGS_BLOCK), tile,
                                               b->stream, g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward),
                                               b->n, seed, b->first_game, cap, b->d_steps, (uint32_t)per_wave, nullptr, heights);
                        }
                    });
                };
                // the opening as a launch of its own: nobody can win and no column can fill in the first kOpenedPlies plies
                constexpr bool can_open = G::STATIC_K > 0 && 2 * G::STATIC_K - 2 >= (int)kOpenedPlies &&
                                          G::STATIC_H >= (int)kOpenedPlies && G::STATIC_H <= 15;
                if ((flags & 1u) && !capped && can_open && G::NW <= 3 && b->rollout_opening &&
                    b->staging_bytes >= (size_t)b->n * sizeof(OpenedBoard)) {
                    with_rng([&](auto rng_tag) {
                        hipLaunchKernelGGL((k_connect_open_lds<G, decltype(rng_tag)::value>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0,
                                           b->stream, g, reinterpret_cast<OpenedBoard*>(b->d_staging), b->n, seed, b->first_game,
                                           b->d_steps);
                    });
                    launch_lds(std::false_type{}, std::integral_constant<int, kEntryOpened>{});
                } else if (flags & 1u) {
                    if (capped) launch_lds(std::true_type{}, std::integral_constant<int, kEntryInitial>{});
                    else launch_lds(std::false_type{}, std::integral_constant<int, kEntryInitial>{});
                } else {
                    if (capped) launch_lds(std::true_type{}, std::integral_constant<int, kEntryMemory>{});
                    else launch_lds(std::false_type{}, std::integral_constant<int, kEntryMemory>{});
                }
                return;
            }
        }
        if ((flags & 1u) && !b->rollout_generic) {
            if (capped)
                hipLaunchKernelGGL((k_connect_rollout_aligned_wide<G, true>), dim3(blocks), dim3(BGS_BLOCK), 0, b->stream,
                                   g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed,
                                   b->first_game, cap, b->d_steps, (uint32_t)per_wave, per_ply ? 1u : 0u);
            else
                hipLaunchKernelGGL((k_connect_rollout_aligned_wide<G, false>), dim3(blocks), dim3(BGS_BLOCK), 0, b->stream,
                                   g, b->d_planes, b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, seed,
                                   b->first_game, cap, b->d_steps, (uint32_t)per_wave, per_ply ? 1u : 0u);
            return;
        }
        with_game(Tag<GenericGame<G>>{});
    });
    return fused;
}

void connect_unpack_grid(const bgs_batch* b, int8_t* d_grid) {
    const size_t lds = (size_t)BGS_BLOCK * b->cg.h * b->cg.w;
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_unpack<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), lds, b->stream, g, b->d_planes,
                           b->n, d_grid);
    });
}

void connect_cell_planes(const bgs_batch* b, uint64_t* d_dst) {
    const int nwc = (b->cg.h * b->cg.w + 63) / 64;
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_cell_planes<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g, b->d_planes, b->n,
                           d_dst, nwc);
    });
}

void connect_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies) {
    hipLaunchKernelGGL(k_connect_meta, dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, b->cg, b->d_planes, b->d_status,
                       b->n, d_player, d_ended, d_winner, d_plies);
}

void connect_legal(const bgs_batch* b, uint8_t* d_legal, int32_t* d_count) {
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_legal<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g, b->d_planes,
                           b->d_status, b->n, d_legal, d_count);
    });
}

void connect_transition(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out, int8_t* d_grid, int8_t* d_player,
                        int8_t* d_winner, int32_t* d_plies, uint8_t* d_legal, int8_t* d_reward_out, uint32_t* d_done,
                        uint32_t ticket) {
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_transition<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g, b->d_planes,
                           b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_actions, d_status_out, b->d_steps,
                           d_grid, d_player, d_winner, d_plies, d_legal, reinterpret_cast<uint16_t*>(d_reward_out),
                           grid_for(b->n) == 1 ? d_done : nullptr, ticket);
    });
}

void connect_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                  int32_t* d_status_out) {
    dispatch(b->cg, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((k_connect_pack<G>), dim3(grid_for(b->n)), dim3(BGS_BLOCK), 0, b->stream, g, b->d_planes,
                           b->d_status, reinterpret_cast<uint16_t*>(b->d_reward), b->n, d_grid, d_player, d_winner,
                           d_status_out);
    });
}

}  // namespace bgs
