// Measurement aid: what does a producer -> consumer hand-over INSIDE one launch cost on this 8-XCD part, against the
// ~2 us of a kernel boundary?  224 producer workgroups (first in the grid) each write a 4 KB chunk, publish it
// (variants below) and bump their group's counter; 28 consumer workgroups (behind them in the grid, other CUs / XCDs)
// spin on the counter, acquire, read their group's 32 chunks and check every word.
//   mode 0: plain stores, __threadfence(), atomicAdd            | consumer: __threadfence() after the spin, plain loads
//   mode 1: nontemporal stores, release fence (agent), atomicAdd | consumer: acquire fence (agent), plain loads
//   mode 2: nontemporal stores, s_waitcnt only, atomicAdd        | consumer: loads with sc0 sc1 (bypass the caches)
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o tools/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kProd = 224, kGroups = 7, kPerGroup = kProd / kGroups, kConsPerGroup = 4, kChunk = 1024;   // floats per chunk
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 load_sc(const f4* p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void handoff(float* data, unsigned* counter, unsigned long long* stamps, unsigned* bad, int iter) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < kProd) {
        const int g = b % kGroups;
        // some work first so that consumers are resident and spinning when the data appears
        float x = (float)tid;
        for (int i = 0; i < 2000; ++i) x = fmaf(x, 1.0000001f, 1e-7f);
        f4 v = {(float)(b * 7 + iter), (float)tid, (float)iter, x * 0.f};
        f4* dst = reinterpret_cast<f4*>(data + (size_t)b * kChunk) + tid;
        if (MODE == 0) *dst = v; else __builtin_nontemporal_store(v, dst);
        if (MODE == 0) __threadfence();
        else if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid == 0) {
            atomicMax(&stamps[0], (unsigned long long)wall_clock64());           // the latest "published" time
            __hip_atomic_fetch_add(&counter[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int c = b - kProd, g = c % kGroups, part = c / kGroups;
    if (tid == 0) {
        while (__hip_atomic_load(&counter[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kPerGroup * (unsigned)(iter + 1))
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    else if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) atomicMax(&stamps[1], (unsigned long long)wall_clock64());   // the latest "seen" time
    unsigned wrong = 0;
    for (int k = part; k < kPerGroup; k += kConsPerGroup) {
        const int pb = g + kGroups * k;
        const f4* src = reinterpret_cast<const f4*>(data + (size_t)pb * kChunk) + tid;
        const f4 v = MODE == 2 ? load_sc(src) : *src;
        wrong += v.x != (float)(pb * 7 + iter) || v.y != (float)tid || v.z != (float)iter;
    }
    if (wrong) atomicAdd(bad, wrong);
    __syncthreads();
    if (tid == 0) atomicMax(&stamps[2], (unsigned long long)wall_clock64());   // the latest "data checked" time
}
__global__ void boundary_a(float* data, unsigned long long* stamps, int iter) {
    const int b = blockIdx.x, tid = threadIdx.x;
    f4 v = {(float)(b * 7 + iter), (float)tid, (float)iter, 0.f};
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(data + (size_t)b * kChunk) + tid);
    if (tid == 0) atomicMax(&stamps[0], (unsigned long long)wall_clock64());
}
__global__ void boundary_b(const float* data, unsigned long long* stamps, unsigned* bad, int iter) {
    const int c = blockIdx.x, g = c % kGroups, part = c / kGroups, tid = threadIdx.x;
    unsigned wrong = 0;
    for (int k = part; k < kPerGroup; k += kConsPerGroup) {
        const int pb = g + kGroups * k;
        const f4 v = reinterpret_cast<const f4*>(data + (size_t)pb * kChunk)[tid];
        wrong += v.x != (float)(pb * 7 + iter) || v.y != (float)tid || v.z != (float)iter;
    }
    if (wrong) atomicAdd(bad, wrong);
    __syncthreads();
    if (tid == 0) atomicMax(&stamps[2], (unsigned long long)wall_clock64());
}
template <int MODE>
void run(const char* label, float* data, unsigned* counter, unsigned long long* stamps, unsigned* bad) {
    (void)hipMemset(counter, 0, 64); (void)hipMemset(bad, 0, 4);
    double seen = 0, done = 0; const int iters = 50;
    for (int it = 0; it < iters; ++it) {
        (void)hipMemset(stamps, 0, 64);
        hipLaunchKernelGGL(handoff<MODE>, dim3(kProd + kGroups * kConsPerGroup), dim3(256), 0, 0, data, counter, stamps, bad, it);
        (void)hipDeviceSynchronize();
        unsigned long long h[3]; (void)hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost);
        if (it >= 5) { seen += (double)(h[1] - h[0]); done += (double)(h[2] - h[0]); }
    }
    unsigned hb; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-70s last publish -> last consumer saw it %.2f us, -> data checked %.2f us, stale words %u\n", label, seen / (iters - 5) / 100,
           done / (iters - 5) / 100, hb);
}
int main() {
    float* data; unsigned *counter, *bad; unsigned long long* stamps;
    (void)hipMalloc(&data, (size_t)kProd * kChunk * 4); (void)hipMalloc(&counter, 64); (void)hipMalloc(&bad, 4); (void)hipMalloc(&stamps, 64);
    run<0>("in-kernel: plain stores + __threadfence both sides", data, counter, stamps, bad);
    run<1>("in-kernel: nt stores + release / acquire fences (agent)", data, counter, stamps, bad);
    run<2>("in-kernel: nt stores + waitcnt, consumer loads sc0 sc1", data, counter, stamps, bad);
    (void)hipMemset(bad, 0, 4);
    double done = 0; const int iters = 50;
    for (int it = 0; it < iters; ++it) {
        (void)hipMemset(stamps, 0, 64);
        hipLaunchKernelGGL(boundary_a, dim3(kProd), dim3(256), 0, 0, data, stamps, it);
        hipLaunchKernelGGL(boundary_b, dim3(kGroups * kConsPerGroup), dim3(256), 0, 0, data, stamps, bad, it);
        (void)hipDeviceSynchronize();
        unsigned long long h[3]; (void)hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost);
        if (it >= 5) done += (double)(h[2] - h[0]);
    }
    unsigned hb; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-70s last store -> data checked %.2f us, stale words %u\n", "two launches (kernel boundary)", done / (iters - 5) / 100, hb);
    return 0;
}
