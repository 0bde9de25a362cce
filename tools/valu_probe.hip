// Measurement aid (not part of the product): issue cost of the instruction classes the fused prior kernel's generation
// phase is made of -- plain and packed f32 FMA, 32-bit integer multiplies (Philox), transcendentals, f32 <-> f16
// conversions -- and how much of an f16 / f32 MFMA's time vector instructions of the same wave (or of a second wave on
// the SIMD) can use.  One 256- or 512-thread workgroup per CU; cycles by s_memtime around the loop, per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && tools/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
enum { M_FMA, M_PKFMA, M_MULLO_HI, M_MAD64, M_COS, M_CVT_F16, M_CVT_PK_F16, M_XOR, M_MFMA16, M_MFMA16_V4, M_MFMA16_V8, M_MFMA16_V16,
       M_MFMA32F, M_MFMA32F_V4, M_MFMA16_PK4, M_LOG, M_FRACT, M_COUNT };
static const char* kNames[] = {"v_fma_f32 x8", "v_pk_fma_f32 x8", "v_mul_lo_u32 + v_mul_hi_u32 (x4 each)", "v_mad_u64_u32 x8", "v_cos_f32 x8",
                               "v_cvt_f16_f32 x8", "v_cvt_pk_f16_f32 (2 floats -> b32) x8", "v_xor_b32 x8", "mfma_f32_16x16x32_f16 x8",
                               "mfma16 x8 + 4 v_fma each", "mfma16 x8 + 8 v_fma each", "mfma16 x8 + 16 v_fma each", "mfma_f32_16x16x4_f32 x8",
                               "mfma f32 x8 + 4 v_fma each", "mfma16 x8 + 4 v_pk_fma each", "v_log_f32 x8", "v_fract_f32 x8"};

template <int MODE>
__global__ void probe(float* out, unsigned long long* stamps, int iters) {
    float a[8], b = 1.0001f, c = 1e-6f;
    v2f p[8];
    uint32_t u[8], hi[8];
    unsigned long long w[8];
    v4f acc[8];
    v8h fa, fb;
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0f + threadIdx.x * 1e-6f + i; p[i] = (v2f){a[i], a[i] + 0.5f}; u[i] = threadIdx.x * 2654435761u + i; hi[i] = u[i] ^ 77u; w[i] = u[i];
        acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (threadIdx.x & 7)); fb[i] = (_Float16)(0.02f * i); }
    v2f b2 = (v2f){b, b}, c2 = (v2f){c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == M_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == M_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
            if (MODE == M_MULLO_HI) {
                if (i & 1) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi[i]) : "v"(u[i]), "v"(0xD2511F53u));
                else asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(hi[i]) : "v"(u[i]), "v"(0xD2511F53u));
            }
            if (MODE == M_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(u[i]), "v"(0xD2511F53u) : "vcc");
            if (MODE == M_COS) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_CVT_F16) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
            if (MODE == M_CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b));
            if (MODE == M_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(hi[i]));
            if (MODE == M_MFMA16 || MODE == M_MFMA16_V4 || MODE == M_MFMA16_V8 || MODE == M_MFMA16_V16 || MODE == M_MFMA16_PK4) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
                constexpr int NV = MODE == M_MFMA16_V4 ? 4 : MODE == M_MFMA16_V8 ? 8 : MODE == M_MFMA16_V16 ? 16 : 0;
#pragma unroll
                for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
                if (MODE == M_MFMA16_PK4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(b2), "v"(c2));
                }
            }
            if (MODE == M_MFMA32F || MODE == M_MFMA32F_V4) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[i], 0, 0, 0);
                if (MODE == M_MFMA32F_V4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)(u[i] ^ hi[i]) + (float)(uint32_t)(w[i] >> 32) + acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(float* out, unsigned long long* st, int threads) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 16 + w] / iters / 8.0);
    std::sort(c.begin(), c.end());
    printf("%-44s %4d threads/CU: %7.2f cycles per group (median), %7.2f max\n", kNames[MODE], threads, c[c.size() / 2], c.back());
}

template <int MODE>
void run_all(float* out, unsigned long long* st) {
    for (int threads : {256, 512, 1024}) run<MODE>(out, st, threads);
}

int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&st, 256 * 16 * 8);
    printf("cycles per 'group' = one instruction of the x8 modes (or one MFMA plus its vector instructions), per wave; 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD\n");
    run_all<M_FMA>(out, st); run_all<M_PKFMA>(out, st); run_all<M_XOR>(out, st); run_all<M_MULLO_HI>(out, st); run_all<M_MAD64>(out, st);
    run_all<M_COS>(out, st); run_all<M_LOG>(out, st); run_all<M_FRACT>(out, st); run_all<M_CVT_F16>(out, st); run_all<M_CVT_PK_F16>(out, st);
    run_all<M_MFMA16>(out, st); run_all<M_MFMA16_V4>(out, st); run_all<M_MFMA16_V8>(out, st); run_all<M_MFMA16_V16>(out, st); run_all<M_MFMA16_PK4>(out, st);
    run_all<M_MFMA32F>(out, st); run_all<M_MFMA32F_V4>(out, st);
    return 0;
}
