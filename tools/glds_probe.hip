// Measurement / validation aid: whole-workgroup global -> LDS copy with global_load_lds_dwordx4 (no registers,
// all requests in flight at once) against a rolled load/store loop.  Checks the copied image and times both.
//   hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o tools/glds_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

// n16 units of 16 bytes, src and dst 16-byte aligned; returns without waiting
__device__ __forceinline__ void stage16(const float* g, float* lds, int n16, int tid, int nt) {
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int c = wave * 64; c < n16; c += nw * 64)
        if (c + lane < n16)
            __builtin_amdgcn_global_load_lds((gvoid*)(g + 4 * (size_t)(c + lane)), (lvoid*)(lds + 4 * c), 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, long long* t) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const float* s = src + (size_t)blockIdx.x * n;
    long long t0 = wall_clock64();
    if (MODE == 0) {
        for (int e = tid; e < n / 4; e += nt) reinterpret_cast<float4*>(sm)[e] = reinterpret_cast<const float4*>(s)[e];
    } else if (MODE == 1) {
        stage16(s, sm, n / 4, tid, nt);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    } else {
        const int lane = tid & 63;
        for (int c = (tid & ~63); c < n; c += nt)
            if (c + lane < n) __builtin_amdgcn_global_load_lds((gvoid*)(s + c + lane), (lvoid*)(sm + c), 4, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();
    long long t1 = wall_clock64();
    float* d = dst + (size_t)blockIdx.x * n;
    for (int e = tid; e < n; e += nt) d[e] = sm[e] + 1.0f;
    if (tid == 0 && blockIdx.x == 0) t[MODE] = t1 - t0;
}
int main() {
    const int n = NFLOATS, blocks = 112;       // 74 KB per workgroup, like the reverse pass
    std::vector<float> h((size_t)n * blocks), o((size_t)n * blocks);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 100003);
    float *src, *dst, *junk; long long* t;
    hipMalloc(&src, h.size() * 4); hipMalloc(&dst, h.size() * 4); hipMalloc(&t, 32); hipMalloc(&junk, 512 << 20);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)copy_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    hipFuncSetAttribute((const void*)copy_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    hipFuncSetAttribute((const void*)copy_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(junk, rep, 512 << 20);        // push src out of the caches
            hipMemset(dst, 0, h.size() * 4);
            if (mode == 0) hipLaunchKernelGGL(copy_kernel<0>, dim3(blocks), dim3(256), n * 4, 0, src, dst, n, t);
            else if (mode == 1) hipLaunchKernelGGL(copy_kernel<1>, dim3(blocks), dim3(256), n * 4, 0, src, dst, n, t);
            else hipLaunchKernelGGL(copy_kernel<2>, dim3(blocks), dim3(256), n * 4, 0, src, dst, n, t);
            hipDeviceSynchronize();
        }
        hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < o.size(); ++i) bad += o[i] != h[i] + 1.0f;
        long long tt[3]; hipMemcpy(tt, t, 24, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %zu mismatches, staging %.2f us (workgroup 0, cold)\n", mode, mode == 0 ? "rolled loop" : mode == 1 ? "global_load_lds x4" : "global_load_lds x1",
               bad, tt[mode] / 100.0);
    }
    return 0;
}
