// tools/waitvalue_probe.hip -- round 6: the probe behind the third form of the Bounce tail queue (hipcc -O2 --offload-arch=gfx950; on the GPU box it
// printed: flag written at 67913.2 us, consumer ran at 67915.1 us and read 7).
// does hipStreamWaitValue32 hold a stream back until a kernel on ANOTHER stream has written a word of ordinary device memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void producer(uint32_t* word, unsigned long long* stamps, int spins) {
    unsigned long long t0 = __builtin_readcyclecounter();
    stamps[0] = wall_clock64();
    for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
    stamps[1] = wall_clock64();
    __hip_atomic_store(word, 7u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
    stamps[2] = wall_clock64();
    (void)t0;
}
__global__ void consumer(const uint32_t* word, unsigned long long* stamps) {
    stamps[3] = wall_clock64();
    stamps[4] = *word;
}
int main() {
    uint32_t* word; unsigned long long* stamps;
    hipMalloc(&word, 256); hipMalloc(&stamps, 64);
    hipMemset(word, 0, 256); hipMemset(stamps, 0, 64);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipDeviceSynchronize();
    hipError_t e = hipStreamWaitValue32(b, word, 7u, hipStreamWaitValueGte, 0xFFFFFFFFu);
    printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, word, stamps);
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, word, stamps, 20000);
    e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    unsigned long long h[8]; hipMemcpy(h, stamps, 64, hipMemcpyDeviceToHost);
    printf("producer start 0, flag written at %.1f us, producer end %.1f us, consumer ran at %.1f us and read %llu\n",
           (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0, h[4]);
    return 0;
}
