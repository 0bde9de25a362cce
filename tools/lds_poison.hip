// Measurement / QA aid: fill the LDS of every CU with a bit pattern (quiet NaNs by default) between two launches of the library.
// LDS is not cleared between kernels: a kernel that reads an LDS word it never wrote sees whatever the last workgroup on that CU left --
// values that vary with scheduling.  Poisoned, such a read turns into NaNs in the outputs (tools/poison_probe.py --lds).
//   hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/lds_poison.hip -o tools/liblds_poison.so
#include <hip/hip_runtime.h>
#include <cstdint>
namespace {
__global__ __launch_bounds__(256) void lds_poison_kernel(uint32_t pattern, unsigned words, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    for (unsigned i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // (keep the stores: read one word back)
    if (threadIdx.x == 0 && lds[(blockIdx.x * 7919u) % words] != pattern) sink[0] = 1u;
    // stay resident for a while so that the workgroups spread over all CUs instead of reusing the first ones
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000ull) { }
}
}  // namespace
extern "C" int lds_poison(void* stream, uint32_t pattern, uint32_t* dev_sink) {
    static bool attr = false;
    const size_t bytes = 160 * 1024;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 1;
        attr = true;
    }
    // one workgroup per CU holds all of its LDS; twice the CU count in case some CUs get two in sequence
    hipLaunchKernelGGL(lds_poison_kernel, dim3(512), dim3(256), bytes, (hipStream_t)stream, pattern, (unsigned)(bytes / 4), dev_sink);
    return (int)hipGetLastError();
}
