// Measurement aid: how the global -> LDS staging time of ONE workgroup depends on its wave count and on where the
// data is (written by the previous kernel = the step's situation, or pushed out to HBM).  224 workgroups of 52 KB,
// like the reverse pass of a one-problem step.
//   hipcc --offload-arch=gfx950 -O3 tools/glds_scale_probe.hip -o tools/glds_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

// MASK: lanes with (lane & MASK) == MASK skip their request (MASK = 0: none skipped... use 64 for "none")
template <int NT, int MASK = 64>
__global__ __launch_bounds__(NT) void stage_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, long long* t) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const float* s = src + (size_t)blockIdx.x * n;
    __syncthreads();
    long long t0 = wall_clock64();
    for (int c = (tid & ~63); c < n / 4; c += NT)
        if (c + lane < n / 4 && (MASK == 64 || (lane & MASK) != MASK))
            __builtin_amdgcn_global_load_lds((gvoid*)(s + 4 * (size_t)(c + lane)), (lvoid*)(sm + 4 * c), 16, 0, 0);
    long long t1 = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    long long t2 = wall_clock64();
    float acc = 0.f;
    for (int e = tid; e < n; e += NT) acc += sm[e];
    dst[(size_t)blockIdx.x * NT + tid] = acc;
    if (tid == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = t2 - t0; }
}
__global__ void produce(float* p, size_t n);
// the same copy through REGISTERS: every 16-byte load of the thread issued first (UNITS per thread), then the LDS writes
template <int NT, int UNITS>
__global__ __launch_bounds__(NT) void stage_regs_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, long long* t) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)blockIdx.x * n);
    __syncthreads();
    long long t0 = wall_clock64();
    float4 v[UNITS];
#pragma unroll
    for (int u = 0; u < UNITS; ++u) v[u] = s[min(tid + u * NT, n / 4 - 1)];
    long long t1 = wall_clock64();
#pragma unroll
    for (int u = 0; u < UNITS; ++u)
        if (tid + u * NT < n / 4) reinterpret_cast<float4*>(sm)[tid + u * NT] = v[u];
    __syncthreads();
    long long t2 = wall_clock64();
    float acc = 0.f;
    for (int e = tid; e < n; e += NT) acc += sm[e];
    dst[(size_t)blockIdx.x * NT + tid] = acc;
    if (tid == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = t2 - t0; }
}
template <int NT, int UNITS>
void run_regs(float* src, float* dst, long long* t, float* junk, int n, int blocks, bool cold) {
    hipFuncSetAttribute((const void*)stage_regs_kernel<NT, UNITS>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    double issue = 0, land = 0, mx = 0;
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, src, (size_t)n * blocks);
        if (cold) hipMemsetAsync(junk, rep, 1u << 30, 0);
        hipLaunchKernelGGL((stage_regs_kernel<NT, UNITS>), dim3(blocks), dim3(NT), n * 4, 0, src, dst, n, t);
        hipDeviceSynchronize();
        std::vector<long long> h(2 * blocks);
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0, m = 0;
        for (int i = 0; i < blocks; ++i) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i + 1] > m) m = h[2 * i + 1]; }
        if (rep) { issue += a / blocks; land += b / blocks; mx += m; }
    }
    printf("%-28s %4d threads  %3d wgs x %5.1f KB %s: issued %.2f us, landed %.2f us (mean), %.2f us (slowest wg)\n", "through registers", NT, blocks,
           n * 4 / 1024.0, cold ? "cold" : "warm", issue / (reps - 1) / 100, land / (reps - 1) / 100, mx / (reps - 1) / 100);
}
__global__ void produce(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023);
}
template <int NT, int MASK = 64>
void run(const char* label, float* src, float* dst, long long* t, float* junk, int n, int blocks, bool cold) {
    hipFuncSetAttribute((const void*)stage_kernel<NT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    double issue = 0, land = 0, mx = 0;
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, src, (size_t)n * blocks);
        if (cold) hipMemsetAsync(junk, rep, 1u << 30, 0);
        hipLaunchKernelGGL((stage_kernel<NT, MASK>), dim3(blocks), dim3(NT), n * 4, 0, src, dst, n, t);
        hipDeviceSynchronize();
        std::vector<long long> h(2 * blocks);
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0, m = 0;
        for (int i = 0; i < blocks; ++i) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i + 1] > m) m = h[2 * i + 1]; }
        if (rep) { issue += a / blocks; land += b / blocks; mx += m; }
    }
    printf("%-28s %4d threads  %3d wgs x %5.1f KB %s: issued %.2f us, landed %.2f us (mean), %.2f us (slowest wg)\n", label, NT, blocks, n * 4 / 1024.0,
           cold ? "cold" : "warm", issue / (reps - 1) / 100, land / (reps - 1) / 100, mx / (reps - 1) / 100);
}
int main() {
    const int blocks = 224;
    float *src, *dst, *junk; long long* t;
    hipMalloc(&src, (size_t)32768 * 4 * blocks); hipMalloc(&dst, (size_t)1024 * blocks * 4); hipMalloc(&t, 16 * blocks); hipMalloc(&junk, 1u << 30);
    for (int n : {6656, 13312, 26624}) {            // 26, 52, 104 KB
        for (int cold = 0; cold < 2; ++cold) {
            run<256>("", src, dst, t, junk, n, blocks, cold);
            run<512>("", src, dst, t, junk, n, blocks, cold);
            run<1024>("", src, dst, t, junk, n, blocks, cold);
        }
    }
    run_regs<256, 13>(src, dst, t, junk, 13312, blocks, false);
    run_regs<256, 13>(src, dst, t, junk, 13312, blocks, true);
    run_regs<256, 26>(src, dst, t, junk, 26624, blocks, false);
    run_regs<512, 13>(src, dst, t, junk, 26624, blocks, false);
    run<256, 63>("lane 63 of every request off", src, dst, t, junk, 13312, blocks, false);
    run<256, 32>("upper half of every request off", src, dst, t, junk, 13312, blocks, false);
    run<256, 1>("odd lanes off", src, dst, t, junk, 13312, blocks, false);
    run<256>("112 wgs", src, dst, t, junk, 13312, 112, false);
    run<256>("448 wgs", src, dst, t, junk, 13312, 448, false);
    return 0;
}
