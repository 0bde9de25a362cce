// Measurement aid: does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950, and what does a
// cross-stream event dependency cost?   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s, s2; hipStreamCreate(&s); hipStreamCreate(&s2);
    hipEvent_t e1, e2; hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    const long long cyc = 5000;   // wall clock is 100 MHz: 50 us
    const int reps = 200;
    for (int mode = 0; mode < 4; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(s); hipStreamSynchronize(s2);
            double t0 = now();
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) {            // serial pair
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
                } else if (mode == 1) {     // second of the pair in any order
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
                    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, (int*)nullptr);
                } else if (mode == 2) {     // fork/join over two streams
                    hipEventRecord(e1, s); hipStreamWaitEvent(s2, e1, 0);
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, cyc, (int*)nullptr);
                    hipEventRecord(e2, s2); hipStreamWaitEvent(s, e2, 0);
                } else {                    // one short kernel alone (launch floor)
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1LL, (int*)nullptr);
                }
            }
            hipStreamSynchronize(s); hipStreamSynchronize(s2);
            double dt = (now() - t0) / reps * 1e6;
            if (warm) printf("mode %d: %.1f us per iteration\n", mode, dt);
        }
    }
    return 0;
}
