// Does a wave64 VALU instruction cost less when only half of the lanes are enabled?  Times a long VALU loop under
// different exec masks.   hipcc --offload-arch=gfx950 -O3 tools/half_wave.hip -o tools/half_wave && tools/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32u : MODE == 2 ? lane >= 32u : MODE == 3 ? (lane & 1u) == 0u : lane < 16u;
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a = a * 5u + b;
                b = (b ^ c) + d;
                c = (c >> 3) | (c << 29);
                d = d + a;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int MODE>
float run(uint32_t* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 2000;
    uint32_t* out; hipMalloc(&out, blocks * 256 * 4);
    printf("all 64 lanes      : %.3f ms\n", run<0>(out, blocks, iters));
    printf("lanes 0..31       : %.3f ms\n", run<1>(out, blocks, iters));
    printf("lanes 32..63      : %.3f ms\n", run<2>(out, blocks, iters));
    printf("even lanes        : %.3f ms\n", run<3>(out, blocks, iters));
    printf("lanes 0..15       : %.3f ms\n", run<4>(out, blocks, iters));
    return 0;
}
