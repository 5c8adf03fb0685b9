// ubench_pairs.hip -- which VALU instructions of the rollout kernels share a quad-cycle on gfx950 ("VALU2", the second
// instruction a SIMD can start per quad-cycle)?  tools/ubench.hip times one opcode at a time; this one times MIXES: two
// opcodes alternating inside a wave, and two kinds of waves on the same SIMD (even waves run one opcode, odd waves the
// other).  Reported: cycles per wave-instruction per SIMD (2.4 GHz nominal), 8 independent register sets per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_pairs.hip -o tools/ubench_pairs ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITER 2048
#define A_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_SHR(x) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
#define A_AND64(x) asm volatile("v_and_b32_e64 %0, %0, %1" : "+v"(x) : "v"(k));
#define B_BITOP(x) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x80" : "+v"(x) : "v"(k));
#define B_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define B_BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define B_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(k));
#define B_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(k));

#define C_SHR64(x) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(x##q));
#define C_SHL64V(x) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x##q) : "v"(k & 15u));
#define C_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x##q) : "v"(x), "v"(k) : "vcc");
#define B_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x));
#define B_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define B_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(k));
#define A_SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_SHLV(x) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(k));
#define B_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define B_CMPCND(x) asm volatile("v_cmp_eq_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");

#define INIT                                                                                                    \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
             a7 = a0 * 19;                                                                                       \
    uint32_t k = seed | 1u;                                                                                      \
    uint64_t a0q = a0, a1q = a1, a2q = a2, a3q = a3, a4q = a4, a5q = a5, a6q = a6, a7q = a7;
#define FINI out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(a0q ^ a1q ^ a2q ^ a3q ^ a4q ^ a5q ^ a6q ^ a7q);

// one opcode
#define PURE(NAME, OP)                                                                           \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                  \
        INIT for (int it = 0; it < ITER; ++it) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) } \
        } FINI }
// two opcodes alternating inside every wave (16 + 16 per iteration)
#define MIXED(NAME, P, Q)                                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                  \
        INIT for (int it = 0; it < ITER; ++it) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { P(a0) Q(a1) P(a2) Q(a3) P(a4) Q(a5) P(a6) Q(a7) } \
        } FINI }
// two opcodes, two instructions of the first to one of the second (the rollout's mix is about 1 : 2 the other way)
#define MIXED21(NAME, P, Q)                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                  \
        INIT for (int it = 0; it < ITER; ++it) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { P(a0) P(a1) Q(a2) P(a3) P(a4) Q(a5) P(a6) P(a7) } \
        } FINI }
// even waves run P, odd waves run Q (the branch is wave-uniform)
#define SPLIT(NAME, P, Q)                                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                  \
        INIT if ((threadIdx.x >> 6) & 1) {                                                       \
            for (int it = 0; it < ITER; ++it) {                                                  \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) { Q(a0) Q(a1) Q(a2) Q(a3) Q(a4) Q(a5) Q(a6) Q(a7) } } \
        } else {                                                                                 \
            for (int it = 0; it < ITER; ++it) {                                                  \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) { P(a0) P(a1) P(a2) P(a3) P(a4) P(a5) P(a6) P(a7) } } \
        } FINI }

PURE(p_and, A_AND) PURE(p_add, A_ADD) PURE(p_xor, A_XOR) PURE(p_shr, A_SHR) PURE(p_and64, A_AND64)
PURE(p_bitop, B_BITOP) PURE(p_andor, B_ANDOR) PURE(p_bcnt, B_BCNT) PURE(p_align, B_ALIGN) PURE(p_mulhi, B_MULHI)
MIXED(m_and_add, A_AND, A_ADD) MIXED(m_and_bitop, A_AND, B_BITOP) MIXED(m_and_bcnt, A_AND, B_BCNT) MIXED(m_and_align, A_AND, B_ALIGN)
MIXED(m_bitop_bcnt, B_BITOP, B_BCNT) MIXED(m_and_mulhi, A_AND, B_MULHI)
MIXED21(m21_and_bitop, A_AND, B_BITOP) MIXED21(m21_bitop_and, B_BITOP, A_AND)
SPLIT(s_and_add, A_AND, A_ADD) SPLIT(s_and_bitop, A_AND, B_BITOP) SPLIT(s_bitop_bcnt, B_BITOP, B_BCNT) SPLIT(s_and_mulhi, A_AND, B_MULHI)

PURE(p_shr64, C_SHR64) PURE(p_shl64v, C_SHL64V) PURE(p_mad64, C_MAD64) PURE(p_bfe, B_BFE) PURE(p_mullo, B_MULLO) PURE(p_lshladd, B_LSHLADD)
PURE(p_sub, A_SUB) PURE(p_or, A_OR) PURE(p_shlv, A_SHLV) PURE(p_or3, B_OR3) PURE(p_cmpcnd, B_CMPCND)
MIXED(m_and_shr64, A_AND, C_SHR64) MIXED(m_and_mad64, A_AND, C_MAD64) MIXED(m_bitop_shr64, B_BITOP, C_SHR64) MIXED(m_bitop_mad64, B_BITOP, C_MAD64)
MIXED(m_shr64_mad64, C_SHR64, C_MAD64) MIXED(m_and_andor, A_AND, B_ANDOR) MIXED(m_and_bfe, A_AND, B_BFE) MIXED(m_and_or3, A_AND, B_OR3)
MIXED(m_bitop_andor, B_BITOP, B_ANDOR) MIXED(m_and_cmpcnd, A_AND, B_CMPCND) MIXED(m_and_lshladd, A_AND, B_LSHLADD) MIXED(m_bitop_or3, B_BITOP, B_OR3)
MIXED(m_bitop_bitop, B_BITOP, B_BITOP) MIXED(m_and_shl64v, A_AND, C_SHL64V)

typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
    struct Case { const char* name; kern_t fn; };
    std::vector<Case> cases = {
        {"v_and_b32 (VOP2)", p_and}, {"v_add_u32 (VOP2)", p_add}, {"v_xor_b32 (VOP2)", p_xor}, {"v_lshrrev_b32 const (VOP2)", p_shr},
        {"v_and_b32_e64 (VOP3 encoding of the same)", p_and64},
        {"v_bitop3_b32", p_bitop}, {"v_and_or_b32", p_andor}, {"v_bcnt_u32_b32", p_bcnt}, {"v_alignbit_b32", p_align}, {"v_mul_hi_u32", p_mulhi},
        {"in one wave, alternating: and / add", m_and_add}, {"in one wave, alternating: and / bitop3", m_and_bitop},
        {"in one wave, alternating: and / bcnt", m_and_bcnt}, {"in one wave, alternating: and / alignbit", m_and_align},
        {"in one wave, alternating: bitop3 / bcnt", m_bitop_bcnt}, {"in one wave, alternating: and / mul_hi", m_and_mulhi},
        {"in one wave, 2 and : 1 bitop3", m21_and_bitop}, {"in one wave, 2 bitop3 : 1 and", m21_bitop_and},
        {"v_lshrrev_b64 const", p_shr64}, {"v_lshlrev_b64 var", p_shl64v}, {"v_mad_u64_u32", p_mad64}, {"v_bfe_u32", p_bfe}, {"v_mul_lo_u32", p_mullo},
        {"v_lshl_add_u32", p_lshladd}, {"v_sub_u32 (VOP2)", p_sub}, {"v_or_b32 (VOP2)", p_or}, {"v_lshlrev_b32 var (VOP2)", p_shlv}, {"v_or3_b32", p_or3},
        {"v_cmp_eq_u32 + v_cndmask_b32 (2 instr.)", p_cmpcnd},
        {"alternating: and / lshrrev_b64", m_and_shr64}, {"alternating: and / mad_u64_u32", m_and_mad64}, {"alternating: and / lshlrev_b64 var", m_and_shl64v},
        {"alternating: bitop3 / lshrrev_b64", m_bitop_shr64}, {"alternating: bitop3 / mad_u64_u32", m_bitop_mad64}, {"alternating: lshrrev_b64 / mad_u64_u32", m_shr64_mad64},
        {"alternating: and / and_or", m_and_andor}, {"alternating: and / bfe", m_and_bfe}, {"alternating: and / or3", m_and_or3}, {"alternating: and / lshl_add", m_and_lshladd},
        {"alternating: bitop3 / and_or", m_bitop_andor}, {"alternating: bitop3 / or3", m_bitop_or3}, {"alternating: and / (cmp + cndmask)", m_and_cmpcnd},
        {"even waves and, odd waves add", s_and_add}, {"even waves and, odd waves bitop3", s_and_bitop},
        {"even waves bitop3, odd waves bcnt", s_bitop_bcnt}, {"even waves and, odd waves mul_hi", s_and_mulhi},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {2, 6}) {
        printf("---- %d waves per SIMD\n", waves_per_simd);
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
            const double insts = (double)waves_per_simd * ITER * 32;
            printf("%-48s %7.3f ms  %5.2f cycles per wave-instruction per SIMD\n", c.name, ms, ms * 1e6 / insts * 2.4);
        }
    }
    return 0;
}
