// Measurement aid: back-to-back dependent launches against the size of a by-value argument block.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blk { double v[N]; int n; };
template <int N> __global__ void k(Blk<N> a, double* out) { if (threadIdx.x == 0) out[blockIdx.x] = a.v[blockIdx.x % N] + a.v[N - 1] + a.n; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int N> void run(hipStream_t s, double* out) {
    Blk<N> h; for (int i = 0; i < N; ++i) h.v[i] = i; h.n = 1;
    const int reps = 2000;
    for (int warm = 0; warm < 2; ++warm) {
        hipStreamSynchronize(s);
        double t0 = now();
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<N>, dim3(448), dim3(256), 0, s, h, out);
        hipStreamSynchronize(s);
        if (warm) printf("%5zu B by value: %.2f us per launch\n", sizeof(Blk<N>), (now() - t0) / reps * 1e6);
    }
}
int main() {
    double* out; hipMalloc(&out, 8 * 4096);
    hipStream_t s; hipStreamCreate(&s);
    run<2>(s, out); run<16>(s, out); run<32>(s, out); run<64>(s, out); run<96>(s, out); run<128>(s, out); run<176>(s, out); run<224>(s, out); run<400>(s, out);
    return 0;
}
