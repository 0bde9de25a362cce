// Measurement aid: which XCD a workgroup lands on, by grid shape (1-D, 2-D) -- HW_REG_XCC_ID read in the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = (int)(x & 0xf);
}
int main() {
    int* d; hipMalloc(&d, 1 << 22);
    for (int shape = 0; shape < 2; ++shape) {
        dim3 g = shape == 0 ? dim3(200 * 64) : dim3(200, 64);
        int n = g.x * g.y;
        hipLaunchKernelGGL(k, g, dim3(64), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int match = 0; for (int i = 0; i < n; ++i) match += (h[i] == (h[0] + i) % 8);
        printf("grid (%u, %u): first 24 XCC ids:", g.x, g.y);
        for (int i = 0; i < 24; ++i) printf(" %d", h[i]);
        printf("  | ids equal to (id0 + linear index) %% 8: %d of %d\n", match, n);
    }
    return 0;
}
