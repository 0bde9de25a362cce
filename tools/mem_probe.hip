// Measurement aid: latency of dependent load batches in a small consumer grid that reads data a
// previous kernel just wrote (producer -> consumer across a kernel boundary).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void producer(float* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (float)i * 1e-6f;
}
// each thread does `batches` dependent rounds; each round issues `width` independent loads
template <int WIDTH>
__global__ void consumer(const float* buf, size_t stride, int batches, float* out, unsigned long long* stamps) {
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    size_t base = (size_t)blockIdx.x * 65536 + threadIdx.x;
    for (int b = 0; b < batches; ++b) {
        float v[WIDTH];
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) v[k] = buf[base + (size_t)k * stride + (size_t)((int)acc & 1)];
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) acc += v[k];
        base += 256;
    }
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) stamps[blockIdx.x] = r1 - r0;
}
int main() {
    const size_t n = 64u << 20;  // 256 MB of floats
    float *buf, *out; unsigned long long* st;
    hipMalloc(&buf, n * 4); hipMalloc(&out, 1 << 22); hipMalloc(&st, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cblocks : {7, 56, 256}) for (int fresh : {1, 0}) for (int width : {1, 8}) {
        const int batches = 10;
        float ms_sum = 0; unsigned long long ticks = 0;
        for (int rep = 0; rep < 20; ++rep) {
            if (fresh) hipLaunchKernelGGL(producer, dim3(112), dim3(256), 0, 0, buf, (size_t)cblocks * 65536 + 8 * 4240 + 4096);
            else hipLaunchKernelGGL(producer, dim3(112), dim3(256), 0, 0, buf + (32u << 20), (size_t)1 << 20);
            hipEventRecord(e0, 0);
            if (width == 1) hipLaunchKernelGGL(consumer<1>, dim3(cblocks), dim3(256), 0, 0, buf, (size_t)4240, batches, out, st);
            else hipLaunchKernelGGL(consumer<8>, dim3(cblocks), dim3(256), 0, 0, buf, (size_t)4240, batches, out, st);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h; hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
            if (rep >= 5) { ms_sum += ms; ticks += h; }
        }
        printf("consumer blocks %3d  %s  width %d: kernel(event) %6.2f us, in-kernel %6.2f us -> %5.0f ns per dependent batch\n", cblocks,
               fresh ? "just-written" : "stale(cold)  ", width, ms_sum / 15 * 1e3, ticks / 15.0 * 0.01, ticks / 15.0 * 10.0 / batches);
    }
    return 0;
}
