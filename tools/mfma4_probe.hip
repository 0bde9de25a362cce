// Measurement aid: operand layout and A-broadcast (CBSZ / ABID) of v_mfma_f32_4x4x1_16b_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o tools/mfma4_probe && tools/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
    const int lane = threadIdx.x;
    const float a = mode == 0 ? 1.0f + lane : 1.0f;      // mode 0: D names the A lane; mode 1: the B lane
    const float b = mode == 1 ? 1.0f + lane : 1.0f;
    f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
    f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 1, 0, 0);
    f4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 1, 1, 0);
    for (int r = 0; r < 4; ++r) { out[(0 * 64 + lane) * 4 + r] = d0[r]; out[(1 * 64 + lane) * 4 + r] = d1[r]; out[(2 * 64 + lane) * 4 + r] = d2[r]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 2 * 3 * 64 * 4 * sizeof(float));
    probe<<<1, 64>>>(d, 0);
    probe<<<1, 64>>>(d + 3 * 64 * 4, 1);
    static float h[2 * 3 * 64 * 4]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[3] = {"cbsz 0", "cbsz 1 abid 0", "cbsz 1 abid 1"};
    for (int v = 0; v < 3; ++v) {
        printf("== %s: D[lane][reg] = A-lane x B-lane\n", nm[v]);
        for (int lane = 0; lane < 12; ++lane) {
            printf("lane %2d:", lane);
            for (int r = 0; r < 4; ++r)
                printf("  r%d: A%2d*B%2d", r, (int)h[(v * 64 + lane) * 4 + r] - 1, (int)h[3 * 64 * 4 + (v * 64 + lane) * 4 + r] - 1);
            printf("\n");
        }
    }
    return 0;
}
