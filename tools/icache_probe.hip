// Measurement aid: is kernel code kept in the instruction cache from one launch to the next?
//   hipcc --offload-arch=gfx950 -O2 tools/icache_probe.hip -o tools/icache_probe
// Two kernels with ~N_INSTR distinct straight-line instructions each (KB-sized bodies).  Pattern AAAA keeps
// re-running the same code, ABAB alternates two code ranges (2 x body), ABCD... four.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int I>
__device__ __forceinline__ float chunk(float x, float m, float c) {
#pragma clang loop unroll(full)
    for (int i = 0; i < 64; ++i) x = fmaf(x, m, c + (float)(I * 64 + i));      // distinct literal per instruction
    return x;
}
template <int I, int END>
__device__ __forceinline__ float chain(float x, float m, float c) {
    if constexpr (I < END) return chain<I + 1, END>(chunk<I>(x, m, c), m, c);
    else return x;
}
template <int SEED, int N>
__global__ void big(float* out, float a) {
    float x = a + threadIdx.x;
    x = chain<SEED * 1000, SEED * 1000 + N / 64>(x, 1.0001f, 0.5f);
    if (x == 12345.678f) out[0] = x;
}
__global__ void rolled(float* out, float a, int n) {
    float x = a + threadIdx.x, m = 1.0001f, c = 0.5f;
#pragma nounroll
    for (int i = 0; i < n; ++i) { x = fmaf(x, m, c); m += 1e-6f; }
    if (x == 12345.678f) out[0] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int N>
void run(hipStream_t s, float* out, int grid) {
    const int reps = 400;
    for (int pattern = 0; pattern < 3; ++pattern) {
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(s);
            double t0 = now();
            for (int r = 0; r < reps; ++r) {
                const int k = pattern == 0 ? 0 : pattern == 1 ? (r & 1) : (r & 3);
                if (k == 0) hipLaunchKernelGGL((big<1, N>), dim3(grid), dim3(256), 0, s, out, 1.0f);
                if (k == 1) hipLaunchKernelGGL((big<2, N>), dim3(grid), dim3(256), 0, s, out, 1.0f);
                if (k == 2) hipLaunchKernelGGL((big<3, N>), dim3(grid), dim3(256), 0, s, out, 1.0f);
                if (k == 3) hipLaunchKernelGGL((big<4, N>), dim3(grid), dim3(256), 0, s, out, 1.0f);
            }
            hipStreamSynchronize(s);
            if (warm) printf("N=%5d grid=%4d pattern %s: %.2f us per launch\n", N, grid,
                             pattern == 0 ? "AAAA" : pattern == 1 ? "ABAB" : "ABCD", (now() - t0) / reps * 1e6);
        }
    }
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    float* out; (void)hipMalloc(&out, 64);
    for (int n : {1024, 4096, 65536}) {
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(s);
            double t0 = now();
            for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(rolled, dim3(112), dim3(256), 0, s, out, 1.0f, n);
            hipStreamSynchronize(s);
            if (warm) printf("rolled n=%d: %.2f us per launch\n", n, (now() - t0) / 100 * 1e6);
        }
    }
    run<64>(s, out, 112);
    run<1024>(s, out, 112);
    run<4096>(s, out, 112);
    run<4096>(s, out, 1024);
    return 0;
}
