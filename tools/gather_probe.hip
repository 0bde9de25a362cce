// Measurement aid: what the memory system delivers for 16-byte gathers from a table of a given size -- the
// access shape of the likelihood kernel's voxel lookup (one float4 record per sphere query).  For every table
// size and every "coherence" (how far apart in the table the 64 lanes of one gather instruction land) it prints
// gathers/ns chip-wide and the GB/s that is at 16 useful bytes, at one 64-byte sector and at one 128-byte line per
// gather.  This is the ceiling the SDF roofline fraction has to be read against.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Each lane does `rounds` rounds of U independent gathers.  mode 0: every lane uniformly random over the table.
// mode 1: the 64 lanes of a wave walk a random 3-D line through a bricked 512^3 table (stride `step` voxels per
// lane): the shape of one sphere followed over 64 consecutive time steps.
template <int U>
__global__ __launch_bounds__(64) void gather(const float4* __restrict__ table, uint64_t nrec, int rounds, int mode,
                                              float step, int bricked, float* __restrict__ out) {
    const uint32_t wave = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t rec;
            if (mode == 0) {
                const uint32_t h = mix((wave * 64u + lane) * 2654435761u + (uint32_t)(r * U + u) * 40503u + 17u);
                const uint32_t h2 = mix(h + 0x9E3779B9u);
                rec = (((uint64_t)h << 32) | h2) % nrec;
            } else {
                const uint32_t h = mix(wave * 2654435761u + (uint32_t)(r * U + u) * 40503u + 17u);
                const uint32_t n = 512;
                const float x0 = (float)(h & 511u), y0 = (float)((h >> 9) & 511u), z0 = (float)((h >> 18) & 511u);
                const uint32_t d = mix(h + 1u);
                float dx = (float)(d & 1023u) - 511.5f, dy = (float)((d >> 10) & 1023u) - 511.5f, dz = (float)((d >> 20) & 1023u) - 511.5f;
                const float inv = step * rsqrtf(dx * dx + dy * dy + dz * dz);
                const int ix = min(max((int)(x0 + dx * inv * lane), 0), (int)n - 1);
                const int iy = min(max((int)(y0 + dy * inv * lane), 0), (int)n - 1);
                const int iz = min(max((int)(z0 + dz * inv * lane), 0), (int)n - 1);
                if (bricked) {
                    const uint64_t b = ((uint64_t)(ix >> 2) * (n / 4) + (iy >> 2)) * (n / 4) + (iz >> 2);
                    const int m = (iz & 1) | ((iy & 1) << 1) | ((ix & 1) << 2) | ((iz & 2) << 2) | ((iy & 2) << 3) | ((ix & 2) << 4);
                    rec = b * 64 + m;
                } else {
                    rec = ((uint64_t)ix * n + iy) * n + iz;
                }
            }
            v[u] = table[rec];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void fill(float4* t, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = make_float4((float)(i & 1023), 1.f, 2.f, 3.f);
}

int main(int argc, char** argv) {
    const bool calib = argc > 1;      // `gather_probe calib`: only the 2 GiB uniform case (for a --pmc FETCH_SIZE pass)
    const uint64_t max_rec = (2ull << 30) / 16;
    float4* table; float* out;
    if (hipMalloc(&table, max_rec * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, table, max_rec);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* label, uint64_t nrec, int mode, float step, int bricked, int waves_per_cu, int U) {
        const int rounds = 64 / U * 4, blocks = 256 * waves_per_cu * 8;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            if (U == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(64), 0, 0, table, nrec, rounds, mode, step, bricked, out);
            else hipLaunchKernelGGL(gather<8>, dim3(blocks), dim3(64), 0, 0, table, nrec, rounds, mode, step, bricked, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double g = (double)blocks * 64 * rounds * U;
        printf("%-44s U=%d: %7.1f us  %6.2f gathers/ns = %5.2f TB/s @16B  %5.2f @64B  %5.2f @128B\n", label, U, best * 1e3,
               g / (best * 1e6), g * 16 / (best * 1e9), g * 64 / (best * 1e9), g * 128 / (best * 1e9));
    };
    char label[128];
    if (calib) {
        run("uniform random, 2048 MiB table (calibration)", max_rec, 0, 0.f, 0, 8, 4);
        printf("gathers per launch: %llu\n", (unsigned long long)(256ull * 8 * 8 * 64 * (64 / 4 * 4) * 4));
        return 0;
    }
    for (uint64_t mb : {16ull, 64ull, 192ull, 512ull, 2048ull})
        for (int U : {4, 8}) {
            snprintf(label, sizeof label, "uniform random, %4llu MiB table", (unsigned long long)mb);
            run(label, (mb << 20) / 16, 0, 0.f, 0, 8, U);
        }
    for (int bricked : {0, 1})
        for (float step : {0.5f, 1.f, 2.f, 4.f, 8.f}) {
            snprintf(label, sizeof label, "64-lane line, %.1f voxels/lane, %s 2 GiB", step, bricked ? "bricked" : "linear ");
            run(label, max_rec, 1, step, bricked, 8, 4);
        }
    return 0;
}
