// generic_kernels.hip -- the hot path for geometries the bit-packed kernels do not cover: Connect boards taller than
// 15 rows, wider than 16 columns or beyond 192 bits per plane (e.g. Config(20, 20, 5)), Bounce boards with more than
// 64 cells or piece values above 15 (e.g. a 10 x 8 grid, values up to 127).  The reference accepts any
// Config(height, width, count) (src/simulator/game/connect.cpp:26) and any int8 grid (bounce.cpp:26), so the drop-in
// has to as well.
//
// Correct first, not fast: the board lives in HBM in the REFERENCE layout (int8[n][H][W], row 0 = bottom row) -- the
// region the packed kernels use for their bit-planes -- one lane plays one board with plain loops, and the Bounce
// move search runs on multi-word cell masks kept in per-lane scratch.  Same rules, same RNG contract, same canonical
// action order as the packed kernels (connect_kernels.hip, bounce_kernels.hip); parity-tested against the oracle
// and, on geometries both paths cover, against the packed kernels (BGS_FORCE_GENERIC=1).
#include <type_traits>

#include "bgs_common.h"
#include "bgs_internal.h"

// identity of this translation unit as compiled: hash of this file, the kernel headers and the compile flags (csrc/Makefile)
#ifndef BGS_TU_ID
#define BGS_TU_ID "unknown"
#endif
extern "C" const char bgs_tu_id_generic[] = BGS_TU_ID;

namespace bgs {
namespace {

constexpr int kBlock = 64;  // one wave per workgroup: these kernels diverge freely

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// the draw of a ply (RNG contract: bgs_common.h).  Bounce: a philox word per ply, one call per 4 plies; Connect: a word per
// block of four plies, one call per 16 plies, the ply's draw a sub-draw of its block's word.  Recomputed when the ply
// leaves the plies the held call covers.
// (Connect under the strict contract -- per_ply, wave-uniform -- draws as Bounce does)
template <bool CONNECT>
struct Draws {
    Philox4 blk;
    uint32_t have;  // index of the call held + 1 (0 = none)
    uint32_t per_ply = 0;
    __device__ __forceinline__ uint32_t at(uint64_t seed, uint64_t game, uint32_t ply) {
        const uint32_t shift = (CONNECT && !per_ply) ? 4u : 2u;
        if (have != (ply >> shift) + 1u) {
            blk = philox4x32_10(seed, game, ply >> shift);
            have = (ply >> shift) + 1u;
        }
        return CONNECT ? connect_draw(per_ply, blk, ply) : philox_word(blk, ply);
    }
};

// ------------------------------------------------------------------------------------------------
// Connect
// ------------------------------------------------------------------------------------------------
struct GConnect {
    int h, w, k;
};

// stones of `who` in a row through (x, y) along (dx, dy), counting (x, y) itself
__device__ int gc_run(const GConnect& c, const int8_t* g, int x, int y, int dx, int dy, int who) {
    int count = 1;
    for (int s = 1; s < c.k; ++s) {
        const int xx = x + s * dx, yy = y + s * dy;
        if (xx < 0 || xx >= c.w || yy < 0 || yy >= c.h || g[yy * c.w + xx] != who) break;
        ++count;
    }
    for (int s = 1; s < c.k; ++s) {
        const int xx = x - s * dx, yy = y - s * dy;
        if (xx < 0 || xx >= c.w || yy < 0 || yy >= c.h || g[yy * c.w + xx] != who) break;
        ++count;
    }
    return count;
}

__device__ bool gc_wins_at(const GConnect& c, const int8_t* g, int x, int y, int who) {
    return gc_run(c, g, x, y, 1, 0, who) >= c.k || gc_run(c, g, x, y, 0, 1, who) >= c.k ||
           gc_run(c, g, x, y, 1, 1, who) >= c.k || gc_run(c, g, x, y, 1, -1, who) >= c.k;
}

// the mover's stone into column x (not full); returns the new status
__device__ uint32_t gc_drop(const GConnect& c, int8_t* g, int x, uint32_t& plies) {
    int y = 0;
    while (g[y * c.w + x] != -1) ++y;
    const int who = (int)(plies & 1u);
    g[y * c.w + x] = (int8_t)who;
    ++plies;
    if (gc_wins_at(c, g, x, y, who)) return (uint32_t)who + 1u;
    return plies == (uint32_t)(c.h * c.w) ? BGS_ST_DRAW : BGS_ST_RUNNING;
}

__device__ int gc_legal_count(const GConnect& c, const int8_t* g) {
    int n = 0;
    for (int x = 0; x < c.w; ++x) n += g[(c.h - 1) * c.w + x] == -1;
    return n;
}

__global__ void __launch_bounds__(kBlock)
g_connect_reset(GConnect c, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies,
                uint16_t* __restrict__ reward, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int8_t* g = grid + i * c.h * c.w;
    for (int t = 0; t < c.h * c.w; ++t) g[t] = -1;
    status[i] = 0;
    plies[i] = 0;
    reward[i] = 0;
}

// plies until the board ends, holds max_plies plies, or `count` plies were played
__global__ void __launch_bounds__(kBlock)
g_connect_play(GConnect c, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
               uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game, uint32_t max_plies,
               uint32_t count, int from_initial, unsigned long long* __restrict__ steps, uint32_t per_ply) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        int8_t* g = grid + i * c.h * c.w;
        uint32_t st, plies;
        if (from_initial) {
            for (int t = 0; t < c.h * c.w; ++t) g[t] = -1;
            st = 0;
            plies = 0;
        } else {
            st = status[i];
            plies = plies_buf[i];
        }
        Draws<true> draws;
        draws.have = 0;
        draws.per_ply = per_ply;
        while (st == BGS_ST_RUNNING && plies < max_plies && stepped < count) {
            const uint32_t idx = sample_index(draws.at(seed, first_game + (uint64_t)i, plies), (uint32_t)gc_legal_count(c, g));
            int x = 0;
            for (uint32_t seen = 0; x < c.w; ++x) {
                if (g[(c.h - 1) * c.w + x] == -1) {
                    if (seen == idx) break;
                    ++seen;
                }
            }
            if (x >= c.w) {  // no open column on a board that claims to be running (cannot arise from validated loads)
                st = BGS_ST_DRAW;
                break;
            }
            st = gc_drop(c, g, x, plies);
            ++stepped;
        }
        if (from_initial || stepped || st != status[i]) {
            status[i] = (uint8_t)st;
            plies_buf[i] = (uint16_t)plies;
            reward[i] = reward_pair(st);
        }
    }
    add_steps(steps, stepped);
}

__global__ void __launch_bounds__(kBlock)
g_connect_step_actions(GConnect c, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                       uint16_t* __restrict__ reward, int64_t n, const int32_t* __restrict__ actions,
                       int32_t* __restrict__ result, unsigned long long* __restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        const int col = actions[i];
        int32_t rc = 0;
        if (col >= 0) {
            rc = -2;  // BGS_ERR_ILLEGAL
            int8_t* g = grid + i * c.h * c.w;
            if (status[i] == BGS_ST_RUNNING && col < c.w && g[(c.h - 1) * c.w + col] == -1) {
                uint32_t plies = plies_buf[i];
                const uint32_t st = gc_drop(c, g, col, plies);
                plies_buf[i] = (uint16_t)plies;
                if (st != BGS_ST_RUNNING) {
                    status[i] = (uint8_t)st;
                    reward[i] = reward_pair(st);
                }
                stepped = 1;
                rc = 0;
            }
        }
        if (result) result[i] = rc;
    }
    add_steps(steps, stepped);
}

__global__ void __launch_bounds__(kBlock)
g_connect_legal(GConnect c, const int8_t* __restrict__ grid, const uint8_t* __restrict__ status, int64_t n,
                uint8_t* __restrict__ legal, int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int8_t* g = grid + i * c.h * c.w;
    const bool running = status[i] == BGS_ST_RUNNING;
    int total = 0;
    for (int x = 0; x < c.w; ++x) {
        const int open = running && g[(c.h - 1) * c.w + x] == -1;
        if (legal) legal[i * c.w + x] = (uint8_t)open;
        total += open;
    }
    if (count) count[i] = total;
}

// reference layout in, with validation (cell codes, gravity, stone counts, the winner the grid implies)
__global__ void __launch_bounds__(kBlock)
g_connect_pack(GConnect c, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
               uint16_t* __restrict__ reward, int64_t n, const int8_t* __restrict__ src, const int8_t* __restrict__ player,
               const int8_t* __restrict__ winner, int32_t* __restrict__ result) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int hw = c.h * c.w;
    const int8_t* s = src + i * hw;
    bool ok = true;
    int c0 = 0, c1 = 0;
    for (int x = 0; x < c.w; ++x) {
        bool open = false;
        for (int y = 0; y < c.h; ++y) {
            const int v = s[y * c.w + x];
            if (v == -1) open = true;
            else if (v == 0 || v == 1) {
                if (open) ok = false;
                if (v == 0) ++c0; else ++c1;
            } else ok = false;
        }
    }
    if (!(c0 == c1 || c0 == c1 + 1)) ok = false;
    if (player && ok && player[i] != ((c0 + c1) & 1)) ok = false;
    bool run0 = false, run1 = false;
    if (ok) {
        // a run starts at (x, y) and goes right, up, or along either diagonal
        const int dx[4] = {1, 0, 1, 1}, dy[4] = {0, 1, 1, -1};
        for (int y = 0; y < c.h; ++y)
            for (int x = 0; x < c.w; ++x) {
                const int who = s[y * c.w + x];
                if (who < 0) continue;
                for (int d = 0; d < 4; ++d) {
                    int len = 1;
                    while (len < c.k) {
                        const int xx = x + len * dx[d], yy = y + len * dy[d];
                        if (xx < 0 || xx >= c.w || yy < 0 || yy >= c.h || s[yy * c.w + xx] != who) break;
                        ++len;
                    }
                    if (len >= c.k) { if (who == 0) run0 = true; else run1 = true; }
                }
            }
    }
    if ((run0 && run1) || (run0 && c0 != c1 + 1) || (run1 && c0 != c1)) ok = false;
    const uint32_t st = run0 ? 1u : (run1 ? 2u : (c0 + c1 == hw ? BGS_ST_DRAW : BGS_ST_RUNNING));
    if (winner) {
        const int wv = winner[i];
        if (wv < -1 || wv > 2) ok = false;
        else if ((wv == -1 ? 0u : (wv == 2 ? BGS_ST_DRAW : (uint32_t)(wv + 1))) != st) ok = false;
    }
    if (ok) {
        int8_t* g = grid + i * hw;
        for (int t = 0; t < hw; ++t) g[t] = s[t];
        status[i] = (uint8_t)st;
        plies_buf[i] = (uint16_t)(c0 + c1);
        reward[i] = reward_pair(st);
    }
    if (result) result[i] = ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// Bounce
// ------------------------------------------------------------------------------------------------
constexpr int kMaskWords = BGS_GENERIC_BOUNCE_MAX_CELLS / 64;

// NW = 64-bit words per cell mask, a compile-time choice (1, 2, 4, 8 or 16 by board size): with static trip counts the
// masks of small boards live in registers instead of scratch
template <int NW>
struct Mask {
    uint64_t w[NW];
};

struct GBounce {
    int h, w;
    const uint64_t* masks;      // device: [6][kMaskWords] all, interior, x > 0, x < w - 1, top row, bottom row
    uint32_t init_status;
};

#define FOR_WORDS _Pragma("unroll") for (int i = 0; i < NW; ++i)
template <int NW> __device__ __forceinline__ void m_zero(Mask<NW>& a) { FOR_WORDS a.w[i] = 0; }
// bit c of the mask; the word is picked with selects, never with a dynamically indexed register array
template <int NW> __device__ __forceinline__ bool m_test(const Mask<NW>& a, int c) {
    uint64_t word = 0;
    FOR_WORDS word = (c >> 6) == i ? a.w[i] : word;
    return (word >> (c & 63)) & 1ull;
}
template <int NW> __device__ __forceinline__ void m_set(Mask<NW>& a, int c) {
    const uint64_t bit = 1ull << (c & 63);
    FOR_WORDS a.w[i] |= (c >> 6) == i ? bit : 0ull;
}
template <int NW> __device__ __forceinline__ void m_clear(Mask<NW>& a, int c) {
    const uint64_t bit = 1ull << (c & 63);
    FOR_WORDS a.w[i] &= (c >> 6) == i ? ~bit : ~0ull;
}
template <int NW> __device__ __forceinline__ int m_count(const Mask<NW>& a) { int t = 0; FOR_WORDS t += __popcll(a.w[i]); return t; }
template <int NW> __device__ __forceinline__ int m_first(const Mask<NW>& a) {
    int found = -1;
    FOR_WORDS if (found < 0 && a.w[i]) found = i * 64 + __ffsll((unsigned long long)a.w[i]) - 1;
    return found;
}
// the k-th set bit (k < count)
template <int NW> __device__ __forceinline__ int m_select(const Mask<NW>& a, int k) {
    int found = -1;
    FOR_WORDS {
        const int cnt = __popcll(a.w[i]);
        if (found < 0 && k < cnt) {
            uint64_t t = a.w[i];
            for (int j = 0; j < k; ++j) t &= t - 1;
            found = i * 64 + __ffsll((unsigned long long)t) - 1;
        }
        k -= found < 0 ? cnt : 0;
    }
    return found;
}
// dst = src shifted towards higher cell indices by s bits, 1 <= s <= 64 (a board is at most 64 cells wide): the word
// offset is 0 or 1, so every source word is a static register
template <int NW> __device__ __forceinline__ void m_shl(Mask<NW>& dst, const Mask<NW>& src, int s) {
    const bool whole = s == 64;
    const int bs = s & 63;
    FOR_WORDS {
        const uint64_t here = src.w[i], below = i > 0 ? src.w[i > 0 ? i - 1 : 0] : 0ull;
        dst.w[i] = whole ? below : ((here << bs) | (below >> (64 - bs)));
    }
}
template <int NW> __device__ __forceinline__ void m_shr(Mask<NW>& dst, const Mask<NW>& src, int s) {
    const bool whole = s == 64;
    const int bs = s & 63;
    FOR_WORDS {
        const uint64_t here = src.w[i], above = i + 1 < NW ? src.w[i + 1 < NW ? i + 1 : 0] : 0ull;
        dst.w[i] = whole ? above : ((here >> bs) | (above << (64 - bs)));
    }
}

// board-level masks of one lane's board, rebuilt from the grid when a kernel picks the board up and kept up to date by
// gb_move
template <int NW>
struct GBoard {
    Mask<NW> occ;
};

template <int NW>
__device__ void gb_load(const GBounce& b, const int8_t* g, GBoard<NW>& bd) {
    m_zero(bd.occ);
    for (int c = 0; c < b.h * b.w; ++c)
        if (g[c] > 0) m_set(bd.occ, c);
}

// every legal landing cell of the piece on cell `src` for `player` (SURVEY Appendix B rules 4-5; the multi-word twin
// of reach() in bounce_kernels.hip)
template <int NW>
__device__ void gb_reach(const GBounce& b, const int8_t* g, const GBoard<NW>& bd, uint32_t player, int src, Mask<NW>& targets) {
    const uint64_t* all = b.masks;
    const uint64_t* interior = b.masks + kMaskWords;
    const uint64_t* not_col0 = b.masks + 2 * kMaskWords;
    const uint64_t* not_collast = b.masks + 3 * kMaskWords;
    Mask<NW> pending, done, a0, al, ar, t1, t2, nf, nl, nr, land;
    m_zero(pending);
    m_zero(done);
    m_zero(targets);
    m_set(pending, src);
    const uint64_t* goal = b.masks + (player ? 5 : 4) * kMaskWords;  // the mover's goal row
    for (int c = m_first(pending); c >= 0; c = m_first(pending)) {
        m_clear(pending, c);
        m_set(done, c);
        const int v = g[c];
        m_zero(a0);
        m_zero(al);
        m_zero(ar);
        m_zero(land);
        m_set(a0, c);
        for (int s = 1; s <= v; ++s) {
            FOR_WORDS {
                t1.w[i] = a0.w[i] | al.w[i];                 // may go on left (and forward)
                t2.w[i] = (a0.w[i] | ar.w[i]);               // may go on right (and forward)
                land.w[i] = t1.w[i] | ar.w[i];               // (scratch: everybody may go forward)
            }
            if (player) m_shr(nf, land, b.w); else m_shl(nf, land, b.w);
            FOR_WORDS { t1.w[i] &= not_col0[i]; t2.w[i] &= not_collast[i]; }
            m_shr(nl, t1, 1);
            m_shl(nr, t2, 1);
            if (s < v) {
                uint64_t left = 0;
                FOR_WORDS {
                    const uint64_t free_i = ~bd.occ.w[i] & interior[i];
                    a0.w[i] = nf.w[i] & free_i;
                    al.w[i] = nl.w[i] & free_i;
                    ar.w[i] = nr.w[i] & free_i;
                    left |= a0.w[i] | al.w[i] | ar.w[i];
                }
                m_zero(land);
                if (!left) break;
            } else {
                FOR_WORDS land.w[i] = (nf.w[i] | nl.w[i] | nr.w[i]) & all[i];
            }
        }
        FOR_WORDS {
            const uint64_t free_i = ~bd.occ.w[i] & interior[i];
            targets.w[i] |= land.w[i] & free_i;
            pending.w[i] |= land.w[i] & bd.occ.w[i] & interior[i] & ~done.w[i];
        }
        FOR_WORDS targets.w[i] |= land.w[i] & goal[i];  // the mover's goal row is a legal landing too
    }
}

// y of the occupied non-goal row nearest the mover's own side, or -1
__device__ int gb_active_row(const GBounce& b, const int8_t* g, uint32_t player) {
    if (player == 0) {
        for (int y = 1; y < b.h - 1; ++y)
            for (int x = 0; x < b.w; ++x)
                if (g[y * b.w + x] > 0) return y;
    } else {
        for (int y = b.h - 2; y >= 1; --y)
            for (int x = 0; x < b.w; ++x)
                if (g[y * b.w + x] > 0) return y;
    }
    return -1;
}

template <int NW>
__device__ int gb_count_actions(const GBounce& b, const int8_t* g, const GBoard<NW>& bd, uint32_t player) {
    const int row = gb_active_row(b, g, player);
    if (row < 0) return 0;
    int total = 0;
    Mask<NW> t;
    for (int x = 0; x < b.w; ++x)
        if (g[row * b.w + x] > 0) {
            gb_reach(b, g, bd, player, row * b.w + x, t);
            total += m_count(t);
        }
    return total;
}

// the idx-th action of the canonical list: sources by ascending x, targets by ascending cell index
template <int NW>
__device__ void gb_pick(const GBounce& b, const int8_t* g, const GBoard<NW>& bd, uint32_t player, int idx, int& s, int& d) {
    const int row = gb_active_row(b, g, player);
    s = d = -1;
    Mask<NW> t;
    for (int x = 0; x < b.w && row >= 0; ++x)
        if (g[row * b.w + x] > 0) {
            gb_reach(b, g, bd, player, row * b.w + x, t);
            const int cnt = m_count(t);
            if (idx < cnt) {
                s = row * b.w + x;
                d = m_select(t, idx);
                return;
            }
            idx -= cnt;
        }
}

template <int NW>
__device__ void gb_move(const GBounce& b, int8_t* g, GBoard<NW>& bd, int s, int d) {
    g[d] = g[s];
    g[s] = 0;
    m_clear(bd.occ, s);
    m_set(bd.occ, d);
}

// status after `mover` moved to cell d (Appendix B rule 7); n_next = action count of the next player
template <int NW>
__device__ uint32_t gb_settle(const GBounce& b, const int8_t* g, const GBoard<NW>& bd, uint32_t mover, int d, int& n_next) {
    const int y = d / b.w;
    n_next = 0;
    if (y == 0 || y == b.h - 1) return mover + 1u;
    n_next = gb_count_actions(b, g, bd, 1u - mover);
    if (n_next) return BGS_ST_RUNNING;
    return gb_count_actions(b, g, bd, mover) ? mover + 1u : BGS_ST_DRAW;
}

__global__ void __launch_bounds__(kBlock)
g_bounce_reset(GBounce b, const int8_t* __restrict__ cfg, int8_t* __restrict__ grid, uint8_t* __restrict__ status,
               uint16_t* __restrict__ plies, uint16_t* __restrict__ reward, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int8_t* g = grid + i * b.h * b.w;
    for (int t = 0; t < b.h * b.w; ++t) g[t] = cfg[t];
    status[i] = (uint8_t)b.init_status;
    plies[i] = 0;
    reward[i] = reward_pair(b.init_status);
}

template <int NW>
__global__ void __launch_bounds__(kBlock)
g_bounce_play(GBounce b, const int8_t* __restrict__ cfg, int8_t* __restrict__ grid, uint8_t* __restrict__ status,
              uint16_t* __restrict__ plies_buf, uint16_t* __restrict__ reward, int64_t n, uint64_t seed, uint64_t first_game,
              uint32_t max_plies, uint32_t count, int from_initial, unsigned long long* __restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        int8_t* g = grid + i * b.h * b.w;
        uint32_t st, plies;
        if (from_initial) {
            for (int t = 0; t < b.h * b.w; ++t) g[t] = cfg[t];
            st = b.init_status;
            plies = 0;
        } else {
            st = status[i];
            plies = plies_buf[i];
        }
        const uint32_t st_in = st;
        if (st == BGS_ST_RUNNING) {
            GBoard<NW> bd;
            gb_load(b, g, bd);
            Draws<false> draws;
            draws.have = 0;
            int n_act = gb_count_actions(b, g, bd, plies & 1u);
            if (n_act == 0) {  // a running board whose side to move is blocked (loaded or start position)
                st = gb_count_actions(b, g, bd, 1u - (plies & 1u)) ? (1u - (plies & 1u)) + 1u : BGS_ST_DRAW;
            }
            while (st == BGS_ST_RUNNING && plies < max_plies && stepped < count) {
                const uint32_t mover = plies & 1u;
                const int idx = (int)sample_index(draws.at(seed, first_game + (uint64_t)i, plies), (uint32_t)n_act);
                int s, d;
                gb_pick(b, g, bd, mover, idx, s, d);
                if (s < 0 || d < 0) {  // (cannot arise: idx < n_act, both computed by the same search)
                    st = BGS_ST_DRAW;
                    break;
                }
                gb_move(b, g, bd, s, d);
                ++plies;
                ++stepped;
                st = gb_settle(b, g, bd, mover, d, n_act);
            }
        }
        if (from_initial || stepped || st != st_in) {
            status[i] = (uint8_t)st;
            plies_buf[i] = (uint16_t)plies;
            reward[i] = reward_pair(st);
        }
    }
    add_steps(steps, stepped);
}

template <int NW>
__global__ void __launch_bounds__(kBlock)
g_bounce_step_actions(GBounce b, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
                      uint16_t* __restrict__ reward, int64_t n, const int32_t* __restrict__ moves, int32_t* __restrict__ result,
                      unsigned long long* __restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t stepped = 0;
    if (i < n) {
        const int sx = moves[4 * i], sy = moves[4 * i + 1], tx = moves[4 * i + 2], ty = moves[4 * i + 3];
        int32_t rc = 0;
        if (sx >= 0) {
            rc = -2;
            const bool inside = sx < b.w && sy >= 0 && sy < b.h && tx >= 0 && tx < b.w && ty >= 0 && ty < b.h;
            if (inside && status[i] == BGS_ST_RUNNING && plies_buf[i] < 65535u) {
                int8_t* g = grid + i * b.h * b.w;
                uint32_t plies = plies_buf[i];
                const uint32_t mover = plies & 1u;
                const int s = sy * b.w + sx, d = ty * b.w + tx;
                if (g[s] > 0 && sy == gb_active_row(b, g, mover)) {
                    GBoard<NW> bd;
                    gb_load(b, g, bd);
                    Mask<NW> t;
                    gb_reach(b, g, bd, mover, s, t);
                    if (m_test(t, d)) {
                        gb_move(b, g, bd, s, d);
                        ++plies;
                        int n_next;
                        const uint32_t st = gb_settle(b, g, bd, mover, d, n_next);
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
        }
        if (result) result[i] = rc;
    }
    add_steps(steps, stepped);
}

// legal moves of the side to move: count[n]; wide[n] = { int32 active row (-1: none), uint8 flags[W][H * W] } where
// flags[x][c] = 1 when cell c is a legal target of the piece in column x of the active row (stride: see
// generic_bounce_legal_bytes)
template <int NW>
__global__ void __launch_bounds__(kBlock)
g_bounce_targets(GBounce b, const int8_t* __restrict__ grid, const uint8_t* __restrict__ status,
                 const uint16_t* __restrict__ plies_buf, int64_t n, uint8_t* __restrict__ wide, int64_t stride,
                 int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int8_t* g = grid + i * b.h * b.w;
    const uint32_t player = plies_buf[i] & 1u;
    const int cells = b.h * b.w;
    int row = -1;
    if (status[i] == BGS_ST_RUNNING) row = gb_active_row(b, g, player);
    uint8_t* out = wide ? wide + i * stride : nullptr;
    if (out) {
        *reinterpret_cast<int32_t*>(out) = row;
        for (int t = 0; t < b.w * cells; ++t) out[4 + t] = 0;
    }
    int total = 0;
    if (row >= 0) {
        GBoard<NW> bd;
        gb_load(b, g, bd);
        Mask<NW> t;
        for (int x = 0; x < b.w; ++x)
            if (g[row * b.w + x] > 0) {
                gb_reach(b, g, bd, player, row * b.w + x, t);
                total += m_count(t);
                if (out)
                    for (int c = 0; c < cells; ++c) out[4 + x * cells + c] = (uint8_t)m_test(t, c);
            }
    }
    if (count) count[i] = total;
}

template <int NW>
__global__ void __launch_bounds__(kBlock)
g_bounce_pack(GBounce b, int8_t* __restrict__ grid, uint8_t* __restrict__ status, uint16_t* __restrict__ plies_buf,
              uint16_t* __restrict__ reward, int64_t n, const int8_t* __restrict__ src, const int8_t* __restrict__ player,
              const int8_t* __restrict__ winner, const int32_t* __restrict__ plies, int32_t* __restrict__ result) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int hw = b.h * b.w;
    const int8_t* s = src + i * hw;
    bool ok = true, goal_piece = false;
    for (int c = 0; c < hw; ++c) {
        if (s[c] < 0) ok = false;
        if (s[c] > 0 && (c < b.w || c >= hw - b.w)) goal_piece = true;
    }
    const int wv = winner ? winner[i] : -1;
    if (wv < -1 || wv > 2) ok = false;
    uint32_t st = wv == -1 ? 0u : (wv == 2 ? BGS_ST_DRAW : (uint32_t)(wv + 1));
    if (st == BGS_ST_RUNNING && goal_piece) ok = false;
    const int pl = player ? player[i] : 0;
    if (pl != 0 && pl != 1) ok = false;
    const int32_t np = plies ? plies[i] : pl;
    if (np < 0 || np > 65535 || (np & 1) != pl) ok = false;
    if (ok) {
        int8_t* g = grid + i * hw;
        for (int t = 0; t < hw; ++t) g[t] = s[t];
        if (st == BGS_ST_RUNNING) {
            GBoard<NW> bd;
            gb_load(b, g, bd);
            if (gb_count_actions(b, g, bd, (uint32_t)pl) == 0)
                st = gb_count_actions(b, g, bd, 1u - (uint32_t)pl) ? (1u - (uint32_t)pl) + 1u : BGS_ST_DRAW;
        }
        status[i] = (uint8_t)st;
        plies_buf[i] = (uint16_t)np;
        reward[i] = reward_pair(st);
    }
    if (result) result[i] = ok ? 0 : -1;
}

// player / ended / winner / plies from status + plies (both games)
__global__ void __launch_bounds__(kBlock)
g_meta(const uint8_t* __restrict__ status, const uint16_t* __restrict__ plies_buf, int64_t n, int8_t* __restrict__ player,
       uint8_t* __restrict__ ended, int8_t* __restrict__ winner, int32_t* __restrict__ plies) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t st = status[i], p = plies_buf[i];
    if (player) player[i] = (int8_t)(p & 1u);
    if (ended) ended[i] = st != 0;
    if (winner) winner[i] = st == 0 ? -1 : (st == BGS_ST_DRAW ? 2 : (int8_t)(st - 1));
    if (plies) plies[i] = (int32_t)p;
}

GConnect gconnect(const bgs_batch* b) { return GConnect{b->gen_h, b->gen_w, b->gen_k}; }
GBounce gbounce(const bgs_batch* b) { return GBounce{b->gen_h, b->gen_w, b->d_gen_masks, b->gen_init_status}; }

// mask words by board size: the smallest of 1, 2, 4, 8, 16 that holds height * width cells
template <class F>
void with_words(const bgs_batch* b, F&& f) {
    const int words = (b->gen_h * b->gen_w + 63) / 64;
    if (words <= 1) f(std::integral_constant<int, 1>{});
    else if (words <= 2) f(std::integral_constant<int, 2>{});
    else if (words <= 4) f(std::integral_constant<int, 4>{});
    else if (words <= 8) f(std::integral_constant<int, 8>{});
    else f(std::integral_constant<int, 16>{});
}
int8_t* cells(const bgs_batch* b) { return reinterpret_cast<int8_t*>(b->d_planes); }
uint16_t* rewards(const bgs_batch* b) { return reinterpret_cast<uint16_t*>(b->d_reward); }

}  // namespace

size_t generic_bounce_legal_bytes(int h, int w) { return ((size_t)4 + (size_t)w * h * w + 7) / 8 * 8; }

void generic_reset(const bgs_batch* b) {
    if (b->game == BGS_GAME_CONNECT)
        hipLaunchKernelGGL(g_connect_reset, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gconnect(b), cells(b), b->d_status,
                           b->d_plies, rewards(b), b->n);
    else
        hipLaunchKernelGGL(g_bounce_reset, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gbounce(b), b->d_gen_cfg, cells(b),
                           b->d_status, b->d_plies, rewards(b), b->n);
}

// count plies per board (count = UINT32_MAX with max_plies as the bound = rollout)
void generic_play(const bgs_batch* b, uint64_t seed, uint32_t max_plies, uint32_t count, bool from_initial, bool per_ply) {
    if (max_plies > 65535u) max_plies = 65535u;  // plies are stored as uint16
    if (b->game == BGS_GAME_CONNECT)
        hipLaunchKernelGGL(g_connect_play, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gconnect(b), cells(b), b->d_status,
                           b->d_plies, rewards(b), b->n, seed, b->first_game, max_plies, count, from_initial ? 1 : 0, b->d_steps,
                           (per_ply || b->rng_per_ply) ? 1u : 0u);
    else
        with_words(b, [&](auto words) {
            hipLaunchKernelGGL((g_bounce_play<decltype(words)::value>), dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream,
                               gbounce(b), b->d_gen_cfg, cells(b), b->d_status, b->d_plies, rewards(b), b->n, seed, b->first_game,
                               max_plies, count, from_initial ? 1 : 0, b->d_steps);
        });
}

void generic_step_actions(const bgs_batch* b, const int32_t* d_actions, int32_t* d_status_out) {
    if (b->game == BGS_GAME_CONNECT)
        hipLaunchKernelGGL(g_connect_step_actions, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gconnect(b), cells(b),
                           b->d_status, b->d_plies, rewards(b), b->n, d_actions, d_status_out, b->d_steps);
    else
        with_words(b, [&](auto words) {
            hipLaunchKernelGGL((g_bounce_step_actions<decltype(words)::value>), dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream,
                               gbounce(b), cells(b), b->d_status, b->d_plies, rewards(b), b->n, d_actions, d_status_out, b->d_steps);
        });
}

void generic_unpack_grid(const bgs_batch* b, int8_t* d_grid) {
    (void)hipMemcpyAsync(d_grid, cells(b), (size_t)b->n * b->gen_h * b->gen_w, hipMemcpyDeviceToDevice, b->stream);
}

void generic_meta(const bgs_batch* b, int8_t* d_player, uint8_t* d_ended, int8_t* d_winner, int32_t* d_plies) {
    hipLaunchKernelGGL(g_meta, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, b->d_status, b->d_plies, b->n, d_player,
                       d_ended, d_winner, d_plies);
}

void generic_connect_legal(const bgs_batch* b, uint8_t* d_legal, int32_t* d_count) {
    hipLaunchKernelGGL(g_connect_legal, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gconnect(b), cells(b), b->d_status,
                       b->n, d_legal, d_count);
}

void generic_bounce_targets(const bgs_batch* b, uint8_t* d_wide, int32_t* d_count) {
    with_words(b, [&](auto words) {
        hipLaunchKernelGGL((g_bounce_targets<decltype(words)::value>), dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream,
                           gbounce(b), cells(b), b->d_status, b->d_plies, b->n, d_wide,
                           (int64_t)generic_bounce_legal_bytes(b->gen_h, b->gen_w), d_count);
    });
}

void generic_pack(const bgs_batch* b, const int8_t* d_grid, const int8_t* d_player, const int8_t* d_winner,
                  const int32_t* d_plies, int32_t* d_status_out) {
    if (b->game == BGS_GAME_CONNECT)
        hipLaunchKernelGGL(g_connect_pack, dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream, gconnect(b), cells(b), b->d_status,
                           b->d_plies, rewards(b), b->n, d_grid, d_player, d_winner, d_status_out);
    else
        with_words(b, [&](auto words) {
            hipLaunchKernelGGL((g_bounce_pack<decltype(words)::value>), dim3(blocks_for(b->n)), dim3(kBlock), 0, b->stream,
                               gbounce(b), cells(b), b->d_status, b->d_plies, rewards(b), b->n, d_grid, d_player, d_winner, d_plies,
                               d_status_out);
        });
}

}  // namespace bgs
