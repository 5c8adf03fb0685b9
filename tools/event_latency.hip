// How long after a kernel has finished does the host learn of it?  The kernel burns ~50 us, then its last instruction
// stores a flag into page-locked host memory mapped into the device; the host measures when (a) it sees the flag,
// (b) hipEventQuery on an event recorded behind the kernel first succeeds, (c) hipEventSynchronize returns.
//   hipcc --offload-arch=gfx950 -O3 tools/event_latency.hip -o tools/event_latency && tools/event_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <immintrin.h>

__global__ void burn(volatile uint32_t* flag, uint32_t seq, int iters, uint32_t* sink) {
    uint32_t a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1664525u + 1013904223u;
    if (a == 12345u) *sink = a;
    __threadfence_system();
    if (threadIdx.x == 0) *flag = seq;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    uint32_t* host_flag; uint32_t* dev_flag; uint32_t* sink;
    CK(hipHostMalloc((void**)&host_flag, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&dev_flag, host_flag, 0));
    CK(hipMalloc(&sink, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        double sum_flag = 0, sum_ev = 0; int n = 0;
        for (uint32_t rep = 1; rep <= 60; ++rep) {
            *host_flag = 0;
            const double t0 = now_us();
            hipLaunchKernelGGL(burn, dim3(1), dim3(64), 0, s, dev_flag, rep, iters, sink);
            CK(hipEventRecord(ev, s));
            double t_flag = 0, t_ev = 0;
            if (mode == 0) {          // poll both
                while (t_flag == 0 || t_ev == 0) {
                    if (t_flag == 0 && *(volatile uint32_t*)host_flag == rep) t_flag = now_us();
                    if (t_ev == 0 && hipEventQuery(ev) == hipSuccess) t_ev = now_us();
                }
            } else if (mode == 1) {   // flag by polling, event by synchronize (after the flag)
                while (*(volatile uint32_t*)host_flag != rep) _mm_pause();
                t_flag = now_us();
                CK(hipEventSynchronize(ev));
                t_ev = now_us();
            } else {                  // synchronize only; the flag is read afterwards
                CK(hipEventSynchronize(ev));
                t_ev = now_us();
                t_flag = t_ev;
            }
            if (rep > 10) { sum_flag += t_flag - t0; sum_ev += t_ev - t0; ++n; }
        }
        const char* names[3] = {"poll flag + hipEventQuery", "poll flag, then hipEventSynchronize", "hipEventSynchronize alone"};
        printf("%-40s launch->flag %.1f us   launch->event %.1f us   (event - flag = %.1f us)\n", names[mode], sum_flag / n,
               sum_ev / n, (sum_ev - sum_flag) / n);
    }
    return 0;
}
