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
typedef float f4v __attribute__((ext_vector_type(4)));
// cycles per MFMA of one wave: a chain on one accumulator / four accumulators in turn
template <int KIND, int NACC>
__global__ void rate(float* out, long long* cyc, int n) {
    f4v acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + threadIdx.x, b = 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
            else if (u & 1) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % NACC], 1, 1, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % NACC], 1, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sacc = 0.f;
    for (int i = 0; i < NACC; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x + blockDim.x * blockIdx.x] = sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int NACC>
static void time_rate(const char* what, int waves) {
    float* o; long long* c; (void)hipMalloc(&o, 64 * 16 * sizeof(float)); (void)hipMalloc(&c, 8);
    const int n = 4096;
    rate<KIND, NACC><<<1, 64 * waves>>>(o, c, n);
    rate<KIND, NACC><<<1, 64 * waves>>>(o, c, n);
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%s, %d accumulator(s), %d wave(s) per SIMD: %.1f counter ticks per MFMA of a wave (one workgroup; the chip-wide figures above are the ones to use)\n",
           what, NACC, waves / 4 > 0 ? waves / 4 : 1, (double)h / (8.0 * n));
}
// chip-wide: 1024 workgroups of 4 waves, wall time by events -> core cycles (2.4 GHz) per MFMA per SIMD
template <int KIND, int NACC>
static void time_chip(const char* what) {
    float* o; long long* c; (void)hipMalloc(&o, 1024 * 256 * sizeof(float)); (void)hipMalloc(&c, 8);
    const int n = 8192;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<KIND, NACC><<<1024, 256>>>(o, c, n);
    (void)hipEventRecord(e0);
    rate<KIND, NACC><<<1024, 256>>>(o, c, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 1024.0 * 4 * 8.0 * n / 1024.0;      // MFMAs per SIMD (256 CUs x 4)
    printf("%s, %d accumulator(s), 4 waves per SIMD chip-wide: %.2f ms, %.1f cycles at 2.4 GHz per MFMA per SIMD\n", what, NACC, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    time_chip<0, 1>("16x16x4 f32"); time_chip<0, 4>("16x16x4 f32"); time_chip<1, 1>("4x4x1 16b f32"); time_chip<1, 4>("4x4x1 16b f32");
    time_rate<0, 1>("16x16x4 f32", 4); time_rate<0, 4>("16x16x4 f32", 4); time_rate<0, 4>("16x16x4 f32", 16);
    time_rate<1, 1>("4x4x1 16b f32", 4); time_rate<1, 4>("4x4x1 16b f32", 4); time_rate<1, 4>("4x4x1 16b f32", 16);
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
