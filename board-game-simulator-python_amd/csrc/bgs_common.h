// bgs_common.h -- shared device/host definitions for libbgs (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define BGS_WAVE 64
#define BGS_BLOCK 256

// status byte of a board: 0 running, 1 player 0 won, 2 player 1 won, 3 draw
#define BGS_ST_RUNNING 0u
#define BGS_ST_DRAW 3u

// limits of the packed representations
enum {
    BGS_CONNECT_MAX_WORDS = 3,   // width * (height + 1) <= 192 bits per plane
    BGS_CONNECT_MAX_H = 15,      // column heights are kept 4 bits per column
    BGS_CONNECT_MAX_W = 16,
    BGS_BOUNCE_MAX_CELLS = 64,   // height * width <= 64: one bit per cell in a uint64
    BGS_BOUNCE_MAX_VALUE = 15,   // 4 value bit-planes
    BGS_BOUNCE_MAX_PASSES = 8,   // passes of the multi-pass Bounce rollout
    BGS_BOUNCE_MAX_PIECES = 16,  // piece-list rollout kernel (K3p): pieces on the configured start position
    // the generic (reference-layout) kernels take over beyond the packed limits
    BGS_GENERIC_CONNECT_MAX_DIM = 64,      // height, width <= 64 (the oracle's own limit: nothing larger can be checked)
    BGS_GENERIC_BOUNCE_MAX_CELLS = 1024,   // height * width <= 1024, piece values <= 127 (int8)
    BGS_GENERIC_BOUNCE_MAX_DIM = 64
};

namespace bgs {

// ------------------------------------------------------------------------------------------------
// RNG contract (include/bgs.h): philox4x32-10, key = seed.
//   Bounce : counter = (game lo, game hi, ply >> 2, 0); the draw of a ply is output word (ply & 3): a word per ply.
//   Connect: counter = (game lo, game hi, ply >> 4, 0); the WORD of the four-ply block ply >> 2 is output word
//            (ply >> 2) & 3, and the draw of ply j = ply & 3 of the block is word * kSubDraw[j] mod 2^32, kSubDraw[j] =
//            A^j, A = 747796405 -- four consecutive states of the multiplicative congruential generator x -> A x.  One
//            philox call serves SIXTEEN plies of a game, so a Connect4 game (<= 42 plies) needs three, all of which the
//            bench kernel computes in its lock-step opening: its ply loop holds no philox at all.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kSubDrawA = 747796405u;
constexpr uint32_t kSubDraw1 = kSubDrawA, kSubDraw2 = kSubDrawA * kSubDrawA, kSubDraw3 = kSubDrawA * kSubDrawA * kSubDrawA;
static_assert(kSubDraw2 == 4201498105u && kSubDraw3 == 3399858189u, "A^2, A^3 mod 2^32");
struct Philox4 {
    uint32_t v[4];
};

// a ^ b ^ c in one VALU instruction: v_bitop3_b32 with truth table 0x96 (gfx950 has no v_xor3_b32; hipcc emits two
// v_xor_b32 here)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return a ^ b ^ c;
#endif
}

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t game, uint32_t block) {
    uint32_t c0 = (uint32_t)game, c1 = (uint32_t)(game >> 32), c2 = block, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);  // wave-uniform: the key schedule stays in SGPRs
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

// word (ply & 3) of a philox result, for a ply that differs from lane to lane.  Written with masks, not with selects or
// an index: hipcc turns `j & 1 ? p.v[1] : p.v[0]` into an indexed read of the four words, puts them into LDS for it
// (ds_write2 + ds_read + a wait in the middle of the ply) and the HBM-bound per-ply kernel lost a third of its rate to it
// (round 5, 78 -> 111 us at 2^24 boards).
__device__ __forceinline__ uint32_t philox_word(const Philox4& p, uint32_t ply) {
    uint32_t odd = 0u - (ply & 1u), upper = 0u - ((ply >> 1) & 1u);   // all ones or zero
    asm("" : "+v"(odd), "+v"(upper));                                   // (keeps them masks)
    const uint32_t lo = (p.v[1] & odd) | (p.v[0] & ~odd);
    const uint32_t hi = (p.v[3] & odd) | (p.v[2] & ~odd);
    return (hi & upper) | (lo & ~upper);
}

// Connect: the draw of sub-step j (0..3) of a block from the block's word -- j is a compile-time constant in the rollout
// kernels (one v_mul_lo_u32 with a literal), a run-time value in the per-ply kernels
template <uint32_t J>
__host__ __device__ __forceinline__ uint32_t sub_draw(uint32_t word) {
    static_assert(J < 4u, "four plies a block");
    return J == 0u ? word : word * (J == 1u ? kSubDraw1 : J == 2u ? kSubDraw2 : kSubDraw3);
}
__host__ __device__ __forceinline__ uint32_t sub_draw(uint32_t word, uint32_t j) {
    const uint32_t m = (j & 1u) ? kSubDraw1 : 1u;
    return word * ((j & 2u) ? m * kSubDraw2 : m);
}
// Connect: the word of ply's block out of the philox call that covers it (philox4x32_10(seed, game, ply >> 4))
__device__ __forceinline__ uint32_t connect_word(const Philox4& p, uint32_t ply) { return philox_word(p, ply >> 2); }

// ---- Connect under a NAMED contract (round 6).  The word-per-block contract above is the default; the strict one --
// BGS_ROLLOUT_DRAW_PER_PLY / bgs_set_rng_contract(b, BGS_RNG_PER_PLY) -- gives every ply a philox word of its own, exactly
// as Bounce draws: draw(seed, game, ply) = philox(counter = (game lo, game hi, ply >> 2, 0))[ply & 3].
// The draw of one ply, for kernels that make a philox call per ply or per launch:
template <bool PER_PLY>
__device__ __forceinline__ Philox4 connect_philox(uint64_t seed, uint64_t game, uint32_t ply) {
    return philox4x32_10(seed, game, ply >> (PER_PLY ? 2 : 4));
}
template <bool PER_PLY>
__device__ __forceinline__ uint32_t connect_draw(const Philox4& p, uint32_t ply) {
    return PER_PLY ? philox_word(p, ply) : sub_draw(connect_word(p, ply), ply & 3u);
}
// ... the same with the contract as a wave-uniform run-time value (the kernels off the fast paths): per_ply = 0 / 1
__device__ __forceinline__ Philox4 connect_philox(uint32_t per_ply, uint64_t seed, uint64_t game, uint32_t ply) {
    return philox4x32_10(seed, game, ply >> (per_ply ? 2u : 4u));
}
__device__ __forceinline__ uint32_t connect_draw(uint32_t per_ply, const Philox4& p, uint32_t ply) {
    return sub_draw(philox_word(p, per_ply ? ply : ply >> 2), per_ply ? 0u : ply & 3u);
}
// The four draws of block `blk` (plies 4 blk .. 4 blk + 3) of a game, for the block-aligned rollout kernels: ONE philox
// call either way -- per block of four plies under the strict contract (its four words ARE the draws), per four blocks
// otherwise (a lane-dependent word of it, multiplied up)
template <bool PER_PLY>
struct BlockDraws {
    Philox4 p;
    uint32_t word;
    __device__ __forceinline__ BlockDraws(uint64_t seed, uint64_t game, uint32_t blk) {
        p = philox4x32_10(seed, game, PER_PLY ? blk : blk >> 2);
        word = PER_PLY ? 0u : philox_word(p, blk);
    }
    template <uint32_t J>
    __device__ __forceinline__ uint32_t draw() const {
        return PER_PLY ? p.v[J] : sub_draw<J>(word);
    }
    __device__ __forceinline__ uint32_t draw(uint32_t j) const {   // j: a constant once the ply loop is unrolled
        return PER_PLY ? p.v[j] : sub_draw(word, j);
    }
};

__host__ __device__ __forceinline__ uint32_t sample_index(uint32_t draw, uint32_t n_actions) {
    return (uint32_t)(((uint64_t)draw * n_actions) >> 32);
}

// reward pair packed as two int8 in one uint16 (little endian: byte 0 = player 0)
__host__ __device__ __forceinline__ uint16_t reward_pair(uint32_t status) {
    return status == 1u ? (uint16_t)0xFF01u : status == 2u ? (uint16_t)0x01FFu : (uint16_t)0u;
}

// The env-step counter is sharded: BGS_STEP_SHARDS words, one per 64-byte line, summed by the reader.  Atomics on
// ONE address serialise at about 12 ns each (MI355X_MICROARCH.md, row "fanin"): 16384 waves on a single word would
// cost 0.2 ms per launch, more than the kernels themselves.
#define BGS_STEP_SHARDS 256
#define BGS_STEP_STRIDE 8  // uint64 words between shards (64 bytes)

// add this lane's step count into the batch counter: one atomic per wave, on the wave's shard
__device__ __forceinline__ void add_steps(unsigned long long* counter, uint32_t mine) {
    uint32_t total = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, BGS_WAVE);
    if ((threadIdx.x & (BGS_WAVE - 1)) == 0 && total) {
        const uint32_t wave = blockIdx.x * (blockDim.x / BGS_WAVE) + (threadIdx.x / BGS_WAVE);
        atomicAdd(counter + (size_t)(wave % BGS_STEP_SHARDS) * BGS_STEP_STRIDE, (unsigned long long)total);
    }
}

// A workgroup's tile of `bytes` bytes, staged in LDS in its final layout, goes to global memory with 16-byte stores
// (dst 16-byte aligned).  Used by the layout-conversion kernels: each lane expands one board into LDS, then the
// whole workgroup streams the tile out coalesced instead of every lane scattering H*W single bytes.
__device__ __forceinline__ void tile_to_global(const uint8_t* tile, uint8_t* dst, uint32_t bytes) {
    for (uint32_t o = threadIdx.x * 16u; o < bytes; o += BGS_BLOCK * 16u) {
        if (o + 16u <= bytes) {
            *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(tile + o);
        } else {
            for (uint32_t k = o; k < bytes; ++k) dst[k] = tile[k];
        }
    }
}

}  // namespace bgs
