// Measurement aid: in-kernel shader clock (s_memtime / s_memrealtime) and the cost of dependent
// f64 FMA / LDS / rsqrt chains for small grids.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(double* out, unsigned long long* stamps, int iters, int mode) {
    __shared__ double lds[1024];
    lds[threadIdx.x & 1023] = threadIdx.x * 1e-3;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
    if (mode == 0) { for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9); }
    else if (mode == 1) { for (int i = 0; i < iters; ++i) a = fma(a, lds[(i + threadIdx.x) & 1023], 1e-9); }
    else if (mode == 2) { for (int i = 0; i < iters; ++i) a = rsqrt(a + 1.0); }
    else if (mode == 3) { for (int i = 0; i < iters; ++i) a = exp(-a * 1e-3); }
    else if (mode == 4) { for (int i = 0; i < iters; ++i) { a = fma(a, b, 1e-9); __syncthreads(); } }
    else if (mode == 5) { for (int i = 0; i < iters; ++i) a = 1.0 / (a + 1.5); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    double* out; unsigned long long* st;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&st, 4096 * 16);
    const char* names[] = {"f64 fma chain", "f64 fma + lds read", "rsqrt f64 chain", "exp f64 chain", "fma + __syncthreads", "f64 div chain"};
    for (int blocks : {7, 256}) for (int threads : {64, 256, 1024}) for (int mode = 0; mode < 6; ++mode) {
        int iters = 2000;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, out, st, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = (double)h[0] / iters, ns = (double)h[1] * 10.0 / iters;   // memrealtime ticks at 100 MHz
        printf("blocks %3d threads %4d %-22s %7.1f cyc/iter %7.2f ns/iter  clock %.2f GHz\n", blocks, threads, names[mode], cyc, ns, cyc / ns);
    }
    return 0;
}
