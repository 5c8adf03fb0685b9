// ubench.hip -- per-instruction VALU issue cost on gfx950 for the integer ops the rollout kernels are made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
// Each kernel issues ITER * 32 copies of one instruction over 8 independent register sets per lane; reported is
// cycles per wave-instruction per SIMD at the clock measured with s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048

#define BODY32(OP)                                                                  \
    for (int it = 0; it < ITER; ++it) {                                             \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                             \
            OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)                 \
        }                                                                           \
    }

#define KERNEL32(NAME, OP)                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {     \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7,    \
                 a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;            \
        uint32_t k = seed | 1u, s = (seed & 7u) + 1u;                               \
        (void)k; (void)s;                                                           \
        BODY32(OP)                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }

#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_AND_LIT(x) asm volatile("v_and_b32 %0, 0x71111117, %0" : "+v"(x));
#define OP_AND_S(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(k));
#define OP_AND_INL(x) asm volatile("v_and_b32 %0, 15, %0" : "+v"(x));
#define OP_BITOP3_S(x) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x80" : "+v"(x) : "s"(k));
#define OP_CMP_E32(x) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %0" : "+v"(x) : "v"(k) : "vcc");
#define OP_CMP_E64(x) asm volatile("v_cmp_lt_u32_e64 s[10:11], %1, %0" : "+v"(x) : "v"(k) : "s10", "s11");
#define OP_SHR(x) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
#define OP_SHRV(x) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(s));
#define OP_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(k));
#define OP_BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x));
#define OP_FFBL(x) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
#define OP_BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x80" : "+v"(x) : "v"(k));
#define OP_AND_OR(x) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_LSHL_ADD(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(k));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : );
#define OP_CMP_CND(x) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_XAD(x) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));

#define DEP(OP) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0)
#define KERNEL32DEP(NAME, OP)                                                       \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {     \
        uint32_t a0 = threadIdx.x + seed;                                           \
        uint32_t k = seed | 1u, s = (seed & 7u) + 1u;                               \
        (void)k; (void)s;                                                           \
        for (int it = 0; it < ITER; ++it) {                                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { DEP(OP) }               \
        }                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0;                                   \
    }
#define KERNEL64DEP(NAME, OP)                                                       \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {     \
        uint64_t a0 = threadIdx.x + seed;                                           \
        uint64_t k = ((uint64_t)seed << 32) | seed | 1u;                            \
        uint32_t s = (seed & 7u) + 1u;                                              \
        (void)k; (void)s;                                                           \
        for (int it = 0; it < ITER; ++it) {                                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { DEP(OP) }               \
        }                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32); \
    }
KERNEL32(k_add, OP_ADD)
KERNEL32(k_and, OP_AND)
KERNEL32(k_and_lit, OP_AND_LIT)
KERNEL32(k_and_s, OP_AND_S)
KERNEL32(k_and_inl, OP_AND_INL)
KERNEL32(k_bitop3_s, OP_BITOP3_S)
KERNEL32(k_cmp_e32, OP_CMP_E32)
KERNEL32(k_cmp_e64, OP_CMP_E64)
KERNEL32(k_shr, OP_SHR)
KERNEL32(k_shrv, OP_SHRV)
KERNEL32(k_alignbit, OP_ALIGN)
KERNEL32(k_bcnt, OP_BCNT)
KERNEL32(k_mullo, OP_MULLO)
KERNEL32(k_mulhi, OP_MULHI)
KERNEL32(k_bfe, OP_BFE)
KERNEL32(k_ffbl, OP_FFBL)
KERNEL32(k_bitop3, OP_BITOP3)
KERNEL32(k_and_or, OP_AND_OR)
KERNEL32(k_lshl_add, OP_LSHL_ADD)
KERNEL32(k_cndmask, OP_CNDMASK)
KERNEL32(k_cmp_cnd, OP_CMP_CND)
KERNEL32(k_mad24, OP_MAD24)
KERNEL32(k_perm, OP_PERM)
KERNEL32(k_xad, OP_XAD)

#define KERNEL64(NAME, OP)                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {     \
        uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7,    \
                 a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;            \
        uint64_t k = ((uint64_t)seed << 32) | seed | 1u;                            \
        uint32_t s = (seed & 7u) + 1u;                                              \
        (void)k; (void)s;                                                           \
        BODY32(OP)                                                                  \
        uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                         \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);    \
    }

#define OP_SHR64(x) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(x));
#define OP_SHL64V(x) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(s));
#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(s) : "vcc");
#define OP_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(k));
#define OP_CMP64(x) asm volatile("v_cmp_lt_u64 vcc, %1, %0\n\tv_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(k) : "vcc");
#define OP_MOV64(x) asm volatile("v_mov_b64 %0, %1" : "+v"(x) : "v"(k));
#define OP_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(k));

KERNEL32DEP(kd_add, OP_ADD)
KERNEL32DEP(kd_and, OP_AND)
KERNEL32DEP(kd_bcnt, OP_BCNT)
KERNEL32DEP(kd_mullo, OP_MULLO)
KERNEL32DEP(kd_bfe, OP_BFE)
KERNEL32DEP(kd_cmp_cnd, OP_CMP_CND)
KERNEL64(k_shr64, OP_SHR64)
KERNEL64(k_shl64v, OP_SHL64V)
KERNEL64(k_mad64, OP_MAD64)
KERNEL64(k_lshladd64, OP_LSHLADD64)
KERNEL64(k_cmp64_cnd, OP_CMP64)
KERNEL64(k_mov64, OP_MOV64)
KERNEL64DEP(kd_shr64, OP_SHR64)
KERNEL64DEP(kd_mad64, OP_MAD64)

// LDS byte look-up with random addresses (the action-select table)
__global__ void __launch_bounds__(256) k_lds_u8(uint32_t* out, uint32_t seed) {
    __shared__ uint8_t lut[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lut[i] = (uint8_t)(i * 37 + seed);
    __syncthreads();
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = a0 * 9 + lut[a0 & 2047]; a1 = a1 * 9 + lut[a1 & 2047]; a2 = a2 * 9 + lut[a2 & 2047]; a3 = a3 * 9 + lut[a3 & 2047];
            a4 = a4 * 9 + lut[a4 & 2047]; a5 = a5 * 9 + lut[a5 & 2047]; a6 = a6 * 9 + lut[a6 & 2047]; a7 = a7 * 9 + lut[a7 & 2047];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
    struct Case { const char* name; kern_t fn; double insts_per_iter; };
    std::vector<Case> cases = {
        {"v_add_u32", k_add, 32}, {"v_and_b32", k_and, 32}, {"v_and_b32 literal", k_and_lit, 32}, {"v_and_b32 sgpr", k_and_s, 32}, {"v_and_b32 inline const", k_and_inl, 32}, {"v_bitop3_b32 sgpr", k_bitop3_s, 32}, {"v_cmp_lt_u32_e32", k_cmp_e32, 32}, {"v_cmp_lt_u32_e64", k_cmp_e64, 32}, {"v_lshrrev_b32 const", k_shr, 32}, {"v_lshrrev_b32 var", k_shrv, 32},
        {"v_alignbit_b32", k_alignbit, 32}, {"v_bcnt_u32_b32", k_bcnt, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_mul_hi_u32", k_mulhi, 32},
        {"v_bfe_u32", k_bfe, 32}, {"v_ffbl_b32", k_ffbl, 32}, {"v_bitop3_b32", k_bitop3, 32}, {"v_and_or_b32", k_and_or, 32},
        {"v_lshl_add_u32", k_lshl_add, 32}, {"v_cndmask_b32", k_cndmask, 32}, {"v_cmp_lt_u32+v_cndmask", k_cmp_cnd, 32},
        {"v_mad_u32_u24", k_mad24, 32}, {"v_perm_b32", k_perm, 32}, {"v_xad_u32", k_xad, 32},
        {"v_lshrrev_b64 const", k_shr64, 32}, {"v_lshlrev_b64 var", k_shl64v, 32}, {"v_mad_u64_u32", k_mad64, 32},
        {"v_lshl_add_u64", k_lshladd64, 32}, {"v_cmp_lt_u64+v_lshl_add_u64", k_cmp64_cnd, 32}, {"v_mov_b64", k_mov64, 32},
        {"lds u8 lookup (+mad)", k_lds_u8, 32},
        {"DEP v_add_u32", kd_add, 32}, {"DEP v_and_b32", kd_and, 32}, {"DEP v_bcnt_u32_b32", kd_bcnt, 32},
        {"DEP v_mul_lo_u32", kd_mullo, 32}, {"DEP v_bfe_u32", kd_bfe, 32}, {"DEP v_cmp+v_cndmask", kd_cmp_cnd, 32},
        {"DEP v_lshrrev_b64", kd_shr64, 32}, {"DEP v_mad_u64_u32", kd_mad64, 32},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        printf("---- %d wave(s) per SIMD (256 CUs x %d blocks of 256 threads)\n", waves_per_simd, waves_per_simd);
        for (auto& c : cases) {
            dim3 grid(256 * waves_per_simd), block(256);
            hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, 12345u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, 12345u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per SIMD = waves_per_simd * ITER * insts_per_iter
            double insts = (double)waves_per_simd * ITER * c.insts_per_iter;
            double ns_per = ms * 1e6 / insts;
            printf("%-28s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles at 2.4 GHz)\n", c.name, ms, ns_per, ns_per * 2.4);
        }
    }
    return 0;
}
