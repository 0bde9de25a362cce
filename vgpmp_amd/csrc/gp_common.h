// Block size, schedule thresholds and the float64 dot / reduction helpers shared by the GP kernels.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once
#include "gp_wtable.h"

namespace {

constexpr int kBlock = 256;
// shared launches (the few-problem schedule) up to this many (problem, latent) pairs: measured on config 2's shape (S = 128): they win
// up to 4 problems x 7 latents, one launch per kernel from 5 (5 problems 147 vs 143 us, 8: 198 vs 150); with few samples (config 3's
// shape, S = 7) up to 64 pairs (5 problems 71 vs 86 us per step, 8: 80 vs 86; 16 problems 117 vs 101)
#ifndef VG_FUSE_MAX_PL
#define VG_FUSE_MAX_PL 32
#endif
constexpr int kFuseMaxPL = VG_FUSE_MAX_PL, kFuseMaxPLFewSamples = 64;
constexpr int kGemmF16MinSamples = 512;      // few problems of this many samples: stage 2's GEMM role in its f16-split form
__host__ __device__ inline int vg_fuse_max_pl(int S) { return S <= 32 ? kFuseMaxPLFewSamples : kFuseMaxPL; }
__device__ __forceinline__ double matern52_dell(double t1, double t2, double ell, double var) {
    double r = fabs(t1 - t2) / ell;
    return var * exp(-kSqrt5 * r) * (5.0 * r * r / (3.0 * ell)) * (1.0 + kSqrt5 * r);
}

// e / n for 0 <= e < 2^21 via a float reciprocal (exact for n <= 4096, checked exhaustively): a 32-bit
// integer division expands to ~30 instructions on the critical path of every indexing loop
__device__ __forceinline__ int vg_div(int e, float inv_n) { return (int)(((float)e + 0.5f) * inv_n); }

// strided dot product with four independent accumulators (a dependent f64 FMA costs ~40 cycles)
__device__ __forceinline__ double dot4(const double* a, int sa, const double* b, int sb, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 < n; k += 4) {
        s0 = fma(a[k * sa], b[k * sb], s0);
        s1 = fma(a[(k + 1) * sa], b[(k + 1) * sb], s1);
        s2 = fma(a[(k + 2) * sa], b[(k + 2) * sb], s2);
        s3 = fma(a[(k + 3) * sa], b[(k + 3) * sb], s3);
    }
    for (; k < n; ++k) s0 = fma(a[k * sa], b[k * sb], s0);
    return (s0 + s1) + (s2 + s3);
}

// the same dot product on EIGHT adjacent lanes (sub = lane & 7): each takes every eighth term -- four loads per
// operand in flight per 32 terms, issued unconditionally on clamped indices and masked afterwards -- then three
// butterfly steps; every lane of the group returns the sum.  A 32-term row costs ~0.25 us instead of ~0.9.
__device__ __forceinline__ double dot8(const double* a, int sa, const double* b, int sb, int n, int sub) {
    double s0 = 0.0, s1 = 0.0;
    for (int k0 = 0; k0 < n; k0 += 32) {
        double av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(k0 + sub + 8 * j, n - 1);
            av[j] = a[k * sa]; bv[j] = b[k * sb];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double t = k0 + sub + 8 * j < n ? av[j] : 0.0;
            if (j & 1) s1 = fma(t, bv[j], s1); else s0 = fma(t, bv[j], s0);
        }
    }
    double s = s0 + s1;
    s += __shfl_xor(s, 1, VG_WAVE); s += __shfl_xor(s, 2, VG_WAVE); s += __shfl_xor(s, 4, VG_WAVE);
    return s;
}

// (tid, nt: the thread's index in and the size of the GROUP that sums -- the workgroup, or a half of one whose two halves run the same
//  role side by side and pass the same barriers: prior_split_cov_b_kernel)
__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nt) {
    v = vg_wave_sum(v);
    __syncthreads();
    if ((tid & (VG_WAVE - 1)) == 0) red[tid / VG_WAVE] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < nt / VG_WAVE; ++k) t += red[k];
    return t;
}
__device__ __forceinline__ double block_sum(double v, double* red) { return block_sum(v, red, (int)threadIdx.x, (int)blockDim.x); }

// ---- float32 = f16 hi + f16 lo (the operands of the f16-split products: gp_prior_split.h, the stage-2 GEMM role of many samples)
typedef float vg_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vg_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 vg_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 vg_h8 __attribute__((ext_vector_type(8)));
typedef float vg_f2 __attribute__((ext_vector_type(2)));
// No packed-FP32 VALU instruction in this library (profiles/r06/flake.md: on MI355X / ROCm 7.0.2 v_pk_fma / v_pk_mul / v_pk_add_f32 that take
// source 1's high register for both results read 0.0 there in lanes 48-63 while another wave of the compute unit runs a wide f16 matrix
// instruction; the build passes -fno-slp-vectorize and tests/test_capi_load.py scans the code objects).  Arithmetic on float vectors is therefore written per component: the compiler turns `v * s` on an
// ext_vector_type into v_pk_mul_f32 whatever the vectoriser is told.
template <typename V> __device__ __forceinline__ V vg_scale4(V v, float s) { return V{v[0] * s, v[1] * s, v[2] * s, v[3] * s}; }

// (x0, x1) = hi + lo, both halves rounded to nearest: v_cvt_pk_f16_f32, the residuals x - hi by v_fma_mix_f32 (an f16
// operand read straight from the packed pair), v_cvt_pk_f16_f32 again
__device__ __forceinline__ void vg_split2(float x0, float x1, vg_h2& hi, vg_h2& lo) {
    hi = __builtin_convertvector((vg_f2){x0, x1}, vg_h2);
    const uint32_t hb = __builtin_bit_cast(uint32_t, hi);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(x1));
    lo = __builtin_convertvector((vg_f2){l0, l1}, vg_h2);
}
__device__ __forceinline__ void vg_split4(const vg_f32x4& x, vg_h4& hi, vg_h4& lo) {
    vg_h2 h0, l0, h1, l1;
    vg_split2(x[0], x[1], h0, l0);
    vg_split2(x[2], x[3], h1, l1);
    hi = (vg_h4){h0[0], h0[1], h1[0], h1[1]};
    lo = (vg_h4){l0[0], l0[1], l1[0], l1[1]};
}



// ---- the W stream (vgpmp_device.h, "The W stream"): eight float16 weights of one Philox block, packed in pairs
// `tab`: the 8192 magnitudes -- kWTable in global memory (a 16 KB table: L1 / L2 resident) or a copy in LDS
__device__ __forceinline__ uint32_t vg_w_pair(uint32_t word, const unsigned short* __restrict__ tab) {
    const uint32_t lo = tab[word & 0x1fffu], hi = tab[(word >> 16) & 0x1fffu];
    return (lo | (hi << 16)) ^ (word & 0x80008000u);
}
__device__ __forceinline__ vg_h8 vg_w8_from(uint4 r, const unsigned short* __restrict__ tab) {
    typedef uint32_t vg_u32x4 __attribute__((ext_vector_type(4)));
    const vg_u32x4 v = {vg_w_pair(r.x, tab), vg_w_pair(r.y, tab), vg_w_pair(r.z, tab), vg_w_pair(r.w, tab)};
    return __builtin_bit_cast(vg_h8, v);
}
__device__ __forceinline__ vg_h8 vg_w8(uint32_t i, uint2 key, const unsigned short* __restrict__ tab) {
    return vg_w8_from(vg_philox(make_uint4(i, (uint32_t)VG_STREAM_W, 0u, 0u), key), tab);
}
// the same eight weights as float32 (exact: every float16 is a float32)
__device__ __forceinline__ void vg_w8_f32(uint32_t i, uint2 key, const unsigned short* __restrict__ tab, float (&z)[8]) {
    const vg_h8 h = vg_w8(i, key, tab);
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (float)h[k];
}

}  // namespace
