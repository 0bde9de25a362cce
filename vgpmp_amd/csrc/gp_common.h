// Block size, schedule thresholds and the float64 dot / reduction helpers shared by the GP kernels.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

constexpr int kBlock = 256;
// shared launches (the few-problem schedule) up to this many (problem, latent) pairs: measured on config 2's shape (S = 128): they win
// up to 4 problems x 7 latents, one launch per kernel from 5 (5 problems 147 vs 143 us, 8: 198 vs 150); with few samples (config 3's
// shape, S = 7) up to 64 pairs (5 problems 71 vs 86 us per step, 8: 80 vs 86; 16 problems 117 vs 101)
constexpr int kFuseMaxPL = 32, kFuseMaxPLFewSamples = 64;
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

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = vg_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & (VG_WAVE - 1)) == 0) red[threadIdx.x / VG_WAVE] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < (int)(blockDim.x / VG_WAVE); ++k) t += red[k];
    return t;
}

}  // namespace
