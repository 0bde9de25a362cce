// GP half of the ELBO step for gfx950: Philox noise, float64 covariance path (Kuu/Kuf, Cholesky,
// q_sqrt assembly, KL), random-Fourier-feature prior on the f32 MFMA pipe, Matheron path assembly,
// and the full reverse pass down to the Adam update of the unconstrained variables.
//
// Reference path: models/vgpmp.py:200-218,265-289; kullback_leiblers/prior_kl.py:16-35;
// covariances/multioutput/Kuus.py:42-53, Kufs.py:26-34; kernel_conditioning/cond_kernel.py:19-22;
// [3P] gpflow_sampling random_fourier / decoupled exact update; utils/miscellaneous.py:68-84.
//
// Precision plan: everything that touches (Kuu + jitter I)^-1 (condition number ~1e7) is float64 on
// the vector pipe; A = Kfu (Kuu + jitter I)^-1 is formed once per latent in float64 and only then
// rounded, so the per-sample work (prior GEMM, path assembly) is well conditioned float32.
#include "gp_path.h"
#include <string.h>

namespace {

constexpr int kBlock = 256;
constexpr double kVarFloor = 0.1;               // models/vgpmp.py:139 positive(lower=1e-1)
constexpr double kSqrt5 = 2.2360679774997896964;

__device__ __forceinline__ double softplus_d(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ double matern52(double t1, double t2, double ell, double var) {
    double r = fabs(t1 - t2) / ell;
    r = sqrt(fmax(r * r, 1e-36));
    return var * (1.0 + kSqrt5 * r + (5.0 / 3.0) * r * r) * exp(-kSqrt5 * r);
}
__device__ __forceinline__ double matern52_dell(double t1, double t2, double ell, double var) {
    double r = fabs(t1 - t2) / ell;
    return var * exp(-kSqrt5 * r) * (5.0 * r * r / (3.0 * ell)) * (1.0 + kSqrt5 * r);
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

// =================================================================================================
// RNG
// =================================================================================================
// omega [P,L,B,D] (Student-t, nu = 5: N(0,1) * rsqrt(chi2_5 / 5)) and beta [P,L,B]
__global__ __launch_bounds__(kBlock) void rng_basis_kernel(int L, int B, int D, float* __restrict__ omega,
                                                            float* __restrict__ beta, uint32_t seed,
                                                            uint32_t problem_base, uint32_t step,
                                                            const uint32_t* __restrict__ ctr) {
    const int p = blockIdx.y;
    if (ctr) step = *ctr;
    const uint32_t lb = blockIdx.x * kBlock + threadIdx.x;
    if (lb >= (uint32_t)(L * B)) return;
    const uint2 key = vg_key(seed, problem_base + p, step);
    float4 c0 = vg_normal4(2u * lb, VG_STREAM_CHI, key);
    float4 c1 = vg_normal4(2u * lb + 1u, VG_STREAM_CHI, key);
    float gam = (c0.x * c0.x + c0.y * c0.y + c0.z * c0.z + c0.w * c0.w + c1.x * c1.x) * 0.2f;
    float sc = 1.0f / sqrtf(gam);
    const uint32_t e0 = lb * (uint32_t)D;
    uint32_t cur = 0xFFFFFFFFu;
    float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d = 0; d < D; ++d) {
        uint32_t e = e0 + d;
        if ((e >> 2) != cur) { cur = e >> 2; nv = vg_normal4(cur, VG_STREAM_OMEGA, key); }
        omega[((size_t)p * L * B + lb) * D + d] = vg_lane(nv, (int)(e & 3u)) * sc;
    }
    uint4 r = vg_philox(make_uint4(lb >> 2, VG_STREAM_BETA, 0u, 0u), key);
    uint32_t rb = (lb & 3u) == 0 ? r.x : (lb & 3u) == 1 ? r.y : (lb & 3u) == 2 ? r.z : r.w;
    beta[(size_t)p * L * B + lb] = 6.283185307179586f * vg_u01(rb);
}

// w [P, nW], eps [P, nE], eps2 [P, nE]: counter i of a stream yields elements 4i..4i+3
__global__ __launch_bounds__(kBlock) void rng_normals_kernel(uint32_t nW, uint32_t nE, float* __restrict__ w,
                                                              float* __restrict__ eps, float* __restrict__ eps2,
                                                              uint32_t seed, uint32_t problem_base, uint32_t step,
                                                              const uint32_t* __restrict__ ctr) {
    const int p = blockIdx.y;
    if (ctr) step = *ctr;
    const uint32_t cW = (nW + 3u) >> 2, cE = (nE + 3u) >> 2;
    uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= cW + 2u * cE) return;
    const uint2 key = vg_key(seed, problem_base + p, step);
    float* dst; uint32_t n, stream;
    if (c < cW) { dst = w + (size_t)p * nW; n = nW; stream = VG_STREAM_W; }
    else if (c < cW + cE) { c -= cW; dst = eps + (size_t)p * nE; n = nE; stream = VG_STREAM_EPS; }
    else { c -= cW + cE; dst = eps2 + (size_t)p * nE; n = nE; stream = VG_STREAM_EPS2; }
    float4 v = vg_normal4(c, stream, key);
    const uint32_t e = 4u * c;
    if (e + 3u < n && ((((size_t)p * n) & 3u) == 0)) {
        *reinterpret_cast<float4*>(dst + e) = v;
    } else {
        if (e < n) dst[e] = v.x;
        if (e + 1u < n) dst[e + 1] = v.y;
        if (e + 2u < n) dst[e + 2] = v.z;
        if (e + 3u < n) dst[e + 3] = v.w;
    }
}

// =================================================================================================
// Covariance path, forward (float64).  One workgroup per (latent, problem).
// =================================================================================================
struct CovArgs {
    int N, M, L, D;
    const double *X, *Zy, *y_u;
    double jitter, kl_scale;
    const double *q_mu, *q_sqrt, *raw_ell, *raw_var;
    vg_workspace ws;
};

__global__ __launch_bounds__(kBlock) void cov_fwd_kernel(CovArgs a) {
    extern __shared__ double sm[];
    const int l = blockIdx.x, p = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D, ld = Mz + 1;
    const size_t pl = (size_t)p * L + l;
    double* La = sm;                 // Cholesky factor (in place)
    double* Li = La + Mz * ld;       // Lk^-1
    double* Ki = Li + Mz * ld;       // (K + jitter I)^-1
    double* zs = Ki + Mz * ld;       // Zy[:, l]
    double* dl = zs + Mz;            // q_mu - p_mu
    __shared__ double red[kBlock / VG_WAVE];

    const double ell = softplus_d(a.raw_ell[pl]);
    const double var = kVarFloor + softplus_d(a.raw_var[pl]);
    if (tid == 0) { a.ws.ell[pl] = ell; a.ws.var[pl] = var; }
    for (int i = tid; i < Mz; i += nt) zs[i] = a.Zy[(size_t)i * D + l];
    __syncthreads();
    double* Kg = a.ws.K + pl * Mz * Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double k = matern52(zs[i], zs[j], ell, var);
        Kg[e] = k;
        La[i * ld + j] = k + (i == j ? a.jitter : 0.0);
        Li[i * ld + j] = 0.0;
    }
    double* Kufg = a.ws.Kuf + pl * Mz * N;
    for (int e = tid; e < Mz * N; e += nt) {
        int i = e / N, n = e - i * N;
        Kufg[e] = matern52(zs[i], a.X[(size_t)n * D + l], ell, var);
    }
    __syncthreads();
    // ---- Cholesky of Kuu + jitter I, right-looking (models/vgpmp.py:214-215)
    for (int k = 0; k < Mz; ++k) {
        if (tid == 0) La[k * ld + k] = sqrt(La[k * ld + k]);
        __syncthreads();
        const double piv = La[k * ld + k];
        for (int i = k + 1 + tid; i < Mz; i += nt) La[i * ld + k] /= piv;
        __syncthreads();
        const int w = Mz - k - 1;
        for (int e = tid; e < w * w; e += nt) {
            int i = k + 1 + e / w, j = k + 1 + e % w;
            if (j <= i) La[i * ld + j] -= La[i * ld + k] * La[j * ld + k];
        }
        __syncthreads();
    }
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        if (j > i) La[i * ld + j] = 0.0;
    }
    __syncthreads();
    // ---- Lk^-1: one lane per column, forward substitution
    if (tid < Mz) {
        const int j = tid;
        Li[j * ld + j] = 1.0 / La[j * ld + j];
        for (int i = j + 1; i < Mz; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += La[i * ld + k] * Li[k * ld + j];
            Li[i * ld + j] = -s / La[i * ld + i];
        }
    }
    __syncthreads();
    double* Lkg = a.ws.Lk + pl * Mz * Mz;
    double* Lig = a.ws.Linv + pl * Mz * Mz;
    double* Kig = a.ws.Kinv + pl * Mz * Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        for (int k = (i > j ? i : j); k < Mz; ++k) s += Li[k * ld + i] * Li[k * ld + j];
        Ki[i * ld + j] = s;
        Kig[e] = s;
        Lkg[e] = La[i * ld + j];
        Lig[e] = Li[i * ld + j];
    }
    __syncthreads();
    // ---- A = Kfu (Kuu + jitter I)^-1   [N, Mz]
    double* A64 = a.ws.A64 + pl * N * Mz;
    float* A32 = a.ws.A + pl * N * Mz;
    for (int e = tid; e < N * Mz; e += nt) {
        int n = e / Mz, m = e - n * Mz;
        double s = 0.0;
        for (int k = 0; k < Mz; ++k) s += Kufg[(size_t)k * N + n] * Ki[k * ld + m];
        A64[e] = s;
        A32[e] = (float)s;
    }
    // ---- q_sqrt = Lk pad(Q) + jitter diag(1,1,0..)      (models/vgpmp.py:208-218)
    const double* Q = a.q_sqrt + pl * M * M;
    float* C32 = a.ws.C + pl * Mz * Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        if (j >= 2) {
            for (int k = j; k <= i; ++k) s += La[i * ld + k] * Q[(size_t)(k - 2) * M + (j - 2)];
        }
        if (i == j && i < 2) s += a.jitter;
        C32[e] = (float)s;
    }
    // ---- q_mu (full) and the KL term  (models/vgpmp.py:200-202, prior_kl.py:16-35)
    const double y0 = a.y_u[((size_t)p * 2 + 0) * L + l], y1 = a.y_u[((size_t)p * 2 + 1) * L + l];
    const double k00 = Kg[0] + a.jitter, k01 = Kg[1], k11 = Kg[Mz + 1] + a.jitter;
    const double det = k00 * k11 - k01 * k01;
    const double c0 = (k11 * y0 - k01 * y1) / det, c1 = (k00 * y1 - k01 * y0) / det;
    __syncthreads();   // Kg written by other threads above
    for (int i = tid; i < Mz; i += nt) {
        double mi = i == 0 ? y0 : (i == 1 ? y1 : a.q_mu[pl * M + (i - 2)]);
        a.ws.m[pl * Mz + i] = (float)mi;
        double ki0 = Kg[(size_t)i * Mz + 0] + (i == 0 ? a.jitter : 0.0);
        double ki1 = Kg[(size_t)i * Mz + 1] + (i == 1 ? a.jitter : 0.0);
        dl[i] = mi - (ki0 * c0 + ki1 * c1);
    }
    __syncthreads();
    double klacc = 0.0;
    for (int i = tid; i < Mz; i += nt) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s += Li[i * ld + k] * dl[k];
        a.ws.afull[pl * Mz + i] = s;
        if (i >= 2) klacc += s * s;
    }
    for (int e = tid; e < M * M; e += nt) {
        int r = e / M, c = e - r * M;
        if (c <= r) {
            double q = Q[e];
            klacc += q * q;
            if (c == r) klacc -= log(q * q);
        }
    }
    double kl = block_sum(klacc, red);
    if (tid == 0) {
        a.ws.kl_l[pl] = 0.5 * (kl - (double)M);
        a.ws.cvec[pl * 2] = c0;
        a.ws.cvec[pl * 2 + 1] = c1;
    }
}

// =================================================================================================
// Random Fourier features  Phi[l, j, b] = sqrt(2 var / B) cos(x_j . omega_lb / ell + beta_lb)
// and dPhi/dell.  Points j < N are rows of X, the rest rows of Zy.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void features_kernel(int N, int Mz, int L, int D, int B,
                                                           const double* __restrict__ X,
                                                           const double* __restrict__ Zy,
                                                           const double* __restrict__ raw_ell,
                                                           const double* __restrict__ raw_var,
                                                           const float* __restrict__ omega,
                                                           const float* __restrict__ beta, float* __restrict__ Phi,
                                                           float* __restrict__ dPhi, uint32_t* __restrict__ tick) {
    if (tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *tick += 1u;
    const int b = blockIdx.x * kBlock + threadIdx.x;
    const int j = blockIdx.y;
    const int l = blockIdx.z % L, p = blockIdx.z / L;
    if (b >= B) return;
    const int J = N + Mz;
    const size_t pl = (size_t)p * L + l;
    const float ell = (float)softplus_d(raw_ell[pl]);
    const float var = (float)(kVarFloor + softplus_d(raw_var[pl]));
    const double* pt = j < N ? X + (size_t)j * D : Zy + (size_t)(j - N) * D;
    const float* om = omega + (pl * B + b) * D;
    float proj = 0.f;
    for (int d = 0; d < D; ++d) proj = fmaf((float)pt[d], om[d], proj);
    float sn, cs;
    sincosf(proj / ell + beta[pl * B + b], &sn, &cs);
    const float c = sqrtf(2.0f * var / (float)B);
    const size_t o = (pl * J + j) * B + b;
    Phi[o] = c * cs;
    if (dPhi) dPhi[o] = c * sn * proj / (ell * ell);
}

// =================================================================================================
// Prior draws  F0[s, l, j] = sum_b w[s, l, b] Phi[l, j, b]   (and H with dPhi) on the f32 MFMA pipe.
// v_mfma_f32_16x16x4_f32: lane -> A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15];
// each lane loads 4 consecutive k (16 B) per operand, so one load pair feeds 4 MFMAs (k = 4g + c).
// =================================================================================================
typedef float vg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kNT = 3;     // 16-column tiles per wave

__global__ __launch_bounds__(kBlock) void prior_gemm_kernel(int S, int L, int J, int B, int SK, int nsel,
                                                             const float* __restrict__ W,
                                                             const float* __restrict__ Phi,
                                                             const float* __restrict__ dPhi, float* __restrict__ F0,
                                                             float* __restrict__ H, size_t slab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int z = blockIdx.z;
    const int sel = z % nsel; z /= nsel;
    const int sk = z % SK; z /= SK;
    const int l = z % L, p = z / L;
    const int s0 = (blockIdx.y * 4 + wave) * 16;
    const int j0 = blockIdx.x * (16 * kNT);
    if (s0 >= S) return;
    const float* Bm = sel == 0 ? Phi : dPhi;
    float* Out = (sel == 0 ? F0 : H) + (size_t)sk * slab;
    const int kchunk = B / SK, kbeg = sk * kchunk, kend = kbeg + kchunk;
    const int r = lane & 15, g = lane >> 4;
    const int srow = min(s0 + r, S - 1);
    const float* ap = W + (((size_t)p * S + srow) * L + l) * B + 4 * g;
    const float* bp[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        int jc = min(j0 + 16 * t + r, J - 1);
        bp[t] = Bm + (((size_t)p * L + l) * J + jc) * B + 4 * g;
    }
    vg_f32x4 acc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc[t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f};
    float4 a_cur = *reinterpret_cast<const float4*>(ap + kbeg);
    float4 b_cur[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) b_cur[t] = *reinterpret_cast<const float4*>(bp[t] + kbeg);
    for (int k = kbeg; k < kend; k += 16) {
        const int kn = (k + 16 < kend) ? k + 16 : k;      // prefetch next k-step while the MFMAs run
        float4 a_nxt = *reinterpret_cast<const float4*>(ap + kn);
        float4 b_nxt[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) b_nxt[t] = *reinterpret_cast<const float4*>(bp[t] + kn);
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.x, b_cur[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.y, b_cur[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.z, b_cur[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.w, b_cur[t].w, acc[t], 0, 0, 0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < kNT; ++t) b_cur[t] = b_nxt[t];
    }
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        const int jc = j0 + 16 * t + r;
        if (jc >= J) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = s0 + g * 4 + q;
            if (s < S) Out[(((size_t)p * S + s) * L + l) * J + jc] = acc[t][q];
        }
    }
}

// =================================================================================================
// Path assembly  (decoupled / Matheron update, vgpmp.py:281-282):
//   u = m + C eps;  r = u - F0(Z) - sqrt(jitter) eps2;  f = F0(X) + A r
// One workgroup per (sample chunk, latent, problem).
// =================================================================================================
struct PathArgs {
    int S, N, Mz, L, SK;
    size_t slab;
    float sqrt_jitter;
    const float *A, *C, *m, *F0, *H, *eps, *eps2;
    float *R, *f;
    const float* G;
    float* part;
    size_t part_len;
    int NC;
};

__device__ __forceinline__ float read_slabs(const float* base, size_t off, int SK, size_t slab) {
    float v = base[off];
    for (int k = 1; k < SK; ++k) v += base[off + (size_t)k * slab];
    return v;
}

__global__ __launch_bounds__(kBlock) void paths_fwd_kernel(PathArgs a) {
    extern __shared__ float smf[];
    const int ch = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L, J = N + Mz, ld = Mz + 1;
    const size_t pl = (size_t)p * L + l;
    float* Cs = smf;                   // [Mz][ld]
    float* As = Cs + Mz * ld;          // [N][ld]
    float* rs = As + N * ld;           // [SC][ld]
    for (int e = tid; e < Mz * Mz; e += nt) Cs[(e / Mz) * ld + e % Mz] = a.C[pl * Mz * Mz + e];
    for (int e = tid; e < N * Mz; e += nt) As[(e / Mz) * ld + e % Mz] = a.A[pl * N * Mz + e];
    __syncthreads();
    const int s_base = ch * VG_SC;
    for (int e = tid; e < VG_SC * Mz; e += nt) {
        const int sl = e / Mz, mi = e - sl * Mz, s = s_base + sl;
        float r = 0.f;
        if (s < S) {
            const float* ep = a.eps + ((size_t)p * S + s) * Mz * L + l;
            float u = a.m[pl * Mz + mi];
            for (int k = 0; k <= mi; ++k) u = fmaf(Cs[mi * ld + k], ep[(size_t)k * L], u);
            const size_t fo = (((size_t)p * S + s) * L + l) * J + N + mi;
            r = u - read_slabs(a.F0, fo, a.SK, a.slab) - a.sqrt_jitter * a.eps2[(((size_t)p * S + s) * Mz + mi) * L + l];
            a.R[(((size_t)p * S + s) * L + l) * Mz + mi] = r;
        }
        rs[sl * ld + mi] = r;
    }
    __syncthreads();
    for (int e = tid; e < VG_SC * N; e += nt) {
        const int sl = e / N, n = e - sl * N, s = s_base + sl;
        if (s >= S) continue;
        float v = read_slabs(a.F0, (((size_t)p * S + s) * L + l) * J + n, a.SK, a.slab);
        for (int k = 0; k < Mz; ++k) v = fmaf(As[n * ld + k], rs[sl * ld + k], v);
        a.f[(((size_t)p * S + s) * L + l) * N + n] = v;
    }
}

// Reverse of the path assembly, reduced over one chunk of samples:
//   dR = G A;  dm = sum_s dR;  dC = dR^T eps;  dA = G^T R;
//   s_var = <G, F0X> - <dR, F0Z>;  s_ell = <G, H_X> - <dR, H_Z>
__global__ __launch_bounds__(kBlock) void paths_bwd_kernel(PathArgs a) {
    extern __shared__ float smf[];
    __shared__ float red[2][kBlock / VG_WAVE];
    const int ch = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L, J = N + Mz, ld = Mz + 1, ldn = N + 1;
    const size_t pl = (size_t)p * L + l;
    float* As = smf;                     // [N][ld]
    float* Gs = As + N * ld;             // [SC][ldn]
    float* Rs = Gs + VG_SC * ldn;        // [SC][ld]
    float* Es = Rs + VG_SC * ld;         // [SC][ld]
    float* dRs = Es + VG_SC * ld;        // [SC][ld]
    const int s_base = ch * VG_SC;
    for (int e = tid; e < N * Mz; e += nt) As[(e / Mz) * ld + e % Mz] = a.A[pl * N * Mz + e];
    for (int e = tid; e < VG_SC * N; e += nt) {
        const int sl = e / N, n = e - sl * N, s = s_base + sl;
        Gs[sl * ldn + n] = s < S ? a.G[(((size_t)p * S + s) * L + l) * N + n] : 0.f;
    }
    for (int e = tid; e < VG_SC * Mz; e += nt) {
        const int sl = e / Mz, mi = e - sl * Mz, s = s_base + sl;
        Rs[sl * ld + mi] = s < S ? a.R[(((size_t)p * S + s) * L + l) * Mz + mi] : 0.f;
        Es[sl * ld + mi] = s < S ? a.eps[(((size_t)p * S + s) * Mz + mi) * L + l] : 0.f;
    }
    __syncthreads();
    float sv = 0.f, se = 0.f;
    for (int e = tid; e < VG_SC * Mz; e += nt) {
        const int sl = e / Mz, mi = e - sl * Mz, s = s_base + sl;
        float d = 0.f;
        for (int n = 0; n < N; ++n) d = fmaf(Gs[sl * ldn + n], As[n * ld + mi], d);
        dRs[sl * ld + mi] = d;
        if (s < S) {
            const size_t fo = (((size_t)p * S + s) * L + l) * J + N + mi;
            sv -= d * read_slabs(a.F0, fo, a.SK, a.slab);
            if (a.H) se -= d * read_slabs(a.H, fo, a.SK, a.slab);
        }
    }
    for (int e = tid; e < VG_SC * N; e += nt) {
        const int sl = e / N, n = e - sl * N, s = s_base + sl;
        if (s >= S) continue;
        const size_t fo = (((size_t)p * S + s) * L + l) * J + n;
        const float gv = Gs[sl * ldn + n];
        sv = fmaf(gv, read_slabs(a.F0, fo, a.SK, a.slab), sv);
        if (a.H) se = fmaf(gv, read_slabs(a.H, fo, a.SK, a.slab), se);
    }
    __syncthreads();
    float* out = a.part + (pl * a.NC + ch) * a.part_len;
    for (int mi = tid; mi < Mz; mi += nt) {
        float t = 0.f;
        for (int sl = 0; sl < VG_SC; ++sl) t += dRs[sl * ld + mi];
        out[mi] = t;
    }
    float* oC = out + Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int mi = e / Mz, k = e - mi * Mz;
        float t = 0.f;
        for (int sl = 0; sl < VG_SC; ++sl) t = fmaf(dRs[sl * ld + mi], Es[sl * ld + k], t);
        oC[e] = t;
    }
    float* oA = oC + Mz * Mz;
    for (int e = tid; e < N * Mz; e += nt) {
        const int n = e / Mz, mi = e - n * Mz;
        float t = 0.f;
        for (int sl = 0; sl < VG_SC; ++sl) t = fmaf(Gs[sl * ldn + n], Rs[sl * ld + mi], t);
        oA[e] = t;
    }
    sv = vg_wave_sum(sv); se = vg_wave_sum(se);
    if ((tid & 63) == 0) { red[0][tid >> 6] = sv; red[1][tid >> 6] = se; }
    __syncthreads();
    if (tid == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int k = 0; k < (int)(nt >> 6); ++k) { t0 += red[0][k]; t1 += red[1][k]; }
        float* os = oA + (size_t)N * Mz;
        os[0] = t0; os[1] = t1; os[2] = 0.f; os[3] = 0.f;
    }
}

// =================================================================================================
// Covariance path, reverse (float64) + Adam.  One workgroup per (latent, problem).
// =================================================================================================
struct CovBwdArgs {
    CovArgs c;
    int NC;
    size_t part_len;
    int nblk;                 // likelihood partial sums per problem
    double lik_scale;         // alpha / S_total
    double *out_lik, *out_kl;
    double *g_qmu, *g_qsqrt, *g_ell, *g_var;
    int do_adam, trainable, want_dell;
    double lr_t;              // lr * sqrt(1 - b2^t) / (1 - b1^t)
    double lr;
    const uint32_t* ctr;      // device step counter (1-based Adam step after the tick) or null
    double *mq_mu, *mq_sqrt, *m_ell, *m_var;      // Adam first moments (alias of params layout)
    double *vq_mu, *vq_sqrt, *v_ell, *v_var;
    double *pq_mu, *pq_sqrt, *p_ell, *p_var;      // parameters (updated in place)
};

__device__ __forceinline__ void adam_update(double* x, double* m, double* v, double g, double lr_t) {
    // Keras Adam (TF 2.12): beta1 = 0.8, beta2 = 0.95 (models/vgpmp.py:77), epsilon 1e-7
    double mm = *m + (g - *m) * (1.0 - 0.8);
    double vv = *v + (g * g - *v) * (1.0 - 0.95);
    *m = mm; *v = vv;
    *x -= lr_t * mm / (sqrt(vv) + 1e-7);
}

__global__ __launch_bounds__(kBlock) void cov_bwd_kernel(CovBwdArgs b) {
    extern __shared__ double sm[];
    __shared__ double red[kBlock / VG_WAVE];
    const CovArgs a = b.c;
    const int l = blockIdx.x, p = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D, ld = Mz + 1;
    const size_t pl = (size_t)p * L + l;
    double* M0 = sm;                 // dC, later U
    double* M1 = M0 + Mz * ld;       // T1, later P
    double* M2 = M1 + Mz * ld;       // dKj
    double* M3 = M2 + Mz * ld;       // dLk, later S
    double* dmv = M3 + Mz * ld;      // [Mz]
    double* ddv = dmv + Mz;          // [Mz]
    double* zs = ddv + Mz;           // [Mz]
    double* sc = zs + Mz;            // [8] scalars
    const double ell = a.ws.ell[pl], var = a.ws.var[pl];
    const double kls = a.kl_scale;
    const double* Kg = a.ws.K + pl * Mz * Mz;
    const double* Lkg = a.ws.Lk + pl * Mz * Mz;
    const double* Lig = a.ws.Linv + pl * Mz * Mz;
    const double* Kig = a.ws.Kinv + pl * Mz * Mz;
    const double* Kufg = a.ws.Kuf + pl * Mz * N;
    const double* A64 = a.ws.A64 + pl * N * Mz;
    const double* af = a.ws.afull + pl * Mz;
    double* dA = a.ws.dA64 + pl * N * Mz;
    const double* Q = a.q_sqrt + pl * M * M;
    const float* part = a.ws.part + pl * b.NC * b.part_len;

    // ---- 1. sum the per-chunk partial reductions (float32 -> float64)
    for (int i = tid; i < Mz; i += nt) {
        double s = 0.0;
        for (int c = 0; c < b.NC; ++c) s += (double)part[c * b.part_len + i];
        dmv[i] = s;
        zs[i] = a.Zy[(size_t)i * D + l];
    }
    for (int e = tid; e < Mz * Mz; e += nt) {
        double s = 0.0;
        for (int c = 0; c < b.NC; ++c) s += (double)part[c * b.part_len + Mz + e];
        M0[(e / Mz) * ld + e % Mz] = s;
    }
    for (int e = tid; e < N * Mz; e += nt) {
        double s = 0.0;
        for (int c = 0; c < b.NC; ++c) s += (double)part[c * b.part_len + Mz + Mz * Mz + e];
        dA[e] = s;
    }
    if (tid < 2) {
        double s = 0.0;
        for (int c = 0; c < b.NC; ++c) s += (double)part[c * b.part_len + Mz + Mz * Mz + (size_t)N * Mz + tid];
        sc[tid] = s;      // sc[0] = s_var, sc[1] = s_ell
    }
    __syncthreads();
    // ---- 2. T1 = A^T dA
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += A64[(size_t)n * Mz + i] * dA[(size_t)n * Mz + j];
        M1[i * ld + j] = s;
    }
    __syncthreads();
    // ---- 3. dKj = -T1 Kinv ; dLk = dC Qp^T (lower) ; dQ = tril(Lk^T dC)[2:,2:] ; dKfu contraction
    double acc_var = 0.0, acc_ell = 0.0;
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        for (int k = 0; k < Mz; ++k) s += M1[i * ld + k] * Kig[(size_t)k * Mz + j];
        M2[i * ld + j] = -s;
        double t = 0.0;
        if (j <= i && j >= 2) {
            // dLk[i][j] = sum_c dC[i][c] Qp[j][c],  Qp[j][c] = Q[j-2][c-2] for 2 <= c <= j
            for (int c = 2; c <= j; ++c) t += M0[i * ld + c] * Q[(size_t)(j - 2) * M + (c - 2)];
        }
        M3[i * ld + j] = t;
    }
    double* gQ = b.g_qsqrt + pl * M * M;
    for (int e = tid; e < M * M; e += nt) {
        int r = e / M, c = e - r * M;
        double s = 0.0;
        if (c <= r) {
            for (int i = r + 2; i < Mz; ++i) s += Lkg[(size_t)i * Mz + (r + 2)] * M0[i * ld + (c + 2)];
            double q = Q[e];
            s += kls * (q - (c == r ? 1.0 / q : 0.0));
        }
        gQ[e] = s;
    }
    for (int e = tid; e < N * Mz; e += nt) {
        int n = e / Mz, mi = e - n * Mz;
        double s = 0.0;
        for (int k = 0; k < Mz; ++k) s += dA[(size_t)n * Mz + k] * Kig[(size_t)k * Mz + mi];
        const double xn = a.X[(size_t)n * D + l];
        acc_var += s * Kufg[(size_t)mi * N + n];
        acc_ell += s * matern52_dell(xn, zs[mi], ell, var);
    }
    // ---- 4. KL reverse: ddelta = Lk^-T [0,0,a]
    for (int k = tid; k < Mz; k += nt) {
        double s = 0.0;
        for (int i = (k > 2 ? k : 2); i < Mz; ++i) s += Lig[(size_t)i * Mz + k] * af[i];
        ddv[k] = s;
    }
    __syncthreads();
    const double c0 = a.ws.cvec[pl * 2], c1 = a.ws.cvec[pl * 2 + 1];
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        if (j <= i) M3[i * ld + j] -= kls * ddv[i] * af[j];
        if (j < 2) M2[i * ld + j] += kls * (-ddv[i]) * (j == 0 ? c0 : c1);
    }
    __syncthreads();
    if (tid == 0) {
        double d0 = 0.0, d1 = 0.0;
        for (int i = 0; i < Mz; ++i) {
            d0 += (Kg[(size_t)i * Mz + 0] + (i == 0 ? a.jitter : 0.0)) * (-ddv[i]);
            d1 += (Kg[(size_t)i * Mz + 1] + (i == 1 ? a.jitter : 0.0)) * (-ddv[i]);
        }
        const double k00 = Kg[0] + a.jitter, k01 = Kg[1], k11 = Kg[Mz + 1] + a.jitter;
        const double det = k00 * k11 - k01 * k01;
        const double w0 = (k11 * d0 - k01 * d1) / det, w1 = (k00 * d1 - k01 * d0) / det;
        M2[0 * ld + 0] -= kls * w0 * c0; M2[0 * ld + 1] -= kls * w0 * c1;
        M2[1 * ld + 0] -= kls * w1 * c0; M2[1 * ld + 1] -= kls * w1 * c1;
    }
    __syncthreads();
    // ---- 5. P = Phi(Lk^T tril(dLk))
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        if (j <= i) {
            for (int k = i; k < Mz; ++k) s += Lkg[(size_t)k * Mz + i] * M3[k * ld + j];
            if (i == j) s *= 0.5;
        }
        M1[i * ld + j] = s;
    }
    __syncthreads();
    // ---- 6. U = P Lk^-1 (lower)
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        if (j <= i)
            for (int k = j; k <= i; ++k) s += M1[i * ld + k] * Lig[(size_t)k * Mz + j];
        M0[i * ld + j] = s;
    }
    __syncthreads();
    // ---- 7. Sm = Lk^-T U
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
        for (int k = (i > j ? i : j); k < Mz; ++k) s += Lig[(size_t)k * Mz + i] * M0[k * ld + j];
        M3[i * ld + j] = s;
    }
    __syncthreads();
    // ---- 8. dKj += sym(Sm); contract with K and dK/dell
    for (int e = tid; e < Mz * Mz; e += nt) {
        int i = e / Mz, j = e - i * Mz;
        double dk = M2[i * ld + j] + 0.5 * (M3[i * ld + j] + M3[j * ld + i]);
        acc_var += dk * Kg[e];
        acc_ell += dk * matern52_dell(zs[i], zs[j], ell, var);
    }
    acc_var = block_sum(acc_var, red);
    acc_ell = block_sum(acc_ell, red);
    // ---- 9. outputs
    const double g_var = (acc_var / var + sc[0] / (2.0 * var)) * sigmoid_d(a.raw_var[pl]);
    const double g_ell = (acc_ell + (b.want_dell ? sc[1] : 0.0)) * sigmoid_d(a.raw_ell[pl]);
    double* gm = b.g_qmu + pl * M;
    for (int i = tid; i < M; i += nt) gm[i] = dmv[i + 2] + kls * ddv[i + 2];
    if (tid == 0) {
        b.g_ell[pl] = g_ell;
        b.g_var[pl] = g_var;
    }
    if (l == 0 && tid < 64) {
        // ELBO pieces of this problem: alpha/S * sum logp, and KL summed over the latents
        double s = 0.0;
        for (int k = tid; k < b.nblk; k += 64) s += (double)a.ws.lik_partial[(size_t)p * b.nblk + k];
        s = vg_wave_sum(s);
        double kk = 0.0;
        for (int k = tid; k < L; k += 64) kk += a.ws.kl_l[(size_t)p * L + k];
        kk = vg_wave_sum(kk);
        if (tid == 0) { b.out_lik[p] = b.lik_scale * s; b.out_kl[p] = kls * kk; }
    }
    if (!b.do_adam) return;
    __syncthreads();
    if (b.ctr) {
        const double t = (double)*b.ctr;
        b.lr_t = b.lr * sqrt(1.0 - pow(0.95, t)) / (1.0 - pow(0.8, t));
    }
    if (b.trainable & VGPMP_TRAIN_Q_MU)
        for (int i = tid; i < M; i += nt)
            adam_update(b.pq_mu + pl * M + i, b.mq_mu + pl * M + i, b.vq_mu + pl * M + i, gm[i], b.lr_t);
    if (b.trainable & VGPMP_TRAIN_Q_SQRT)
        for (int e = tid; e < M * M; e += nt) {
            int r = e / M, c = e - r * M;
            if (c <= r)
                adam_update(b.pq_sqrt + pl * M * M + e, b.mq_sqrt + pl * M * M + e, b.vq_sqrt + pl * M * M + e, gQ[e],
                            b.lr_t);
        }
    if (tid == 0) {
        if (b.trainable & VGPMP_TRAIN_LENGTHSCALES) adam_update(b.p_ell + pl, b.m_ell + pl, b.v_ell + pl, g_ell, b.lr_t);
        if (b.trainable & VGPMP_TRAIN_KERNEL_VARIANCE) adam_update(b.p_var + pl, b.m_var + pl, b.v_var + pl, g_var, b.lr_t);
    }
}

// forward-only epilogue: ELBO pieces without the reverse pass
__global__ void elbo_pieces_kernel(int L, int nblk, const float* __restrict__ lik_partial,
                                   const double* __restrict__ kl_l, double lik_scale, double kls,
                                   double* __restrict__ out_lik, double* __restrict__ out_kl) {
    const int p = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int k = tid; k < nblk; k += 64) s += (double)lik_partial[(size_t)p * nblk + k];
    s = vg_wave_sum(s);
    double kk = 0.0;
    for (int k = tid; k < L; k += 64) kk += kl_l[(size_t)p * L + k];
    kk = vg_wave_sum(kk);
    if (tid == 0) { out_lik[p] = lik_scale * s; out_kl[p] = kls * kk; }
}

// stand-alone Adam over the packed variables (sample-sharded mode, after the gradient all-reduce)
__global__ __launch_bounds__(kBlock) void adam_kernel(size_t n, double* __restrict__ x, const double* __restrict__ g,
                                                       double* __restrict__ m, double* __restrict__ v, double lr_t,
                                                       int tril_M) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (tril_M > 0) {
        int e = (int)(i % ((size_t)tril_M * tril_M));
        if (e % tril_M > e / tril_M) return;
    }
    adam_update(x + i, m + i, v + i, g[i], lr_t);
}

template <typename T>
T* carve(char*& cur, size_t count, bool real) {
    uintptr_t v = (uintptr_t)cur;
    v = (v + 255) & ~(uintptr_t)255;
    T* out = real ? (T*)v : nullptr;
    cur = (char*)(v + count * sizeof(T));
    return out;
}

}  // namespace

int vg_check_dims(const vgpmp_dims* d) {
    if (d->num_problems < 1 || d->S < 1 || d->S_total < d->S || d->N < 1 || d->M < 1 || d->L < 1 || d->B < 16)
        return VGPMP_E_SHAPE;
    if (d->M + 2 > VGPMP_MAX_MZ || d->L > VGPMP_MAX_DOF || (d->B % 16) != 0) return VGPMP_E_SHAPE;
    if (d->split_k != 1 && d->split_k != 2 && d->split_k != 4 && d->split_k != 8) return VGPMP_E_SHAPE;
    if ((d->B / d->split_k) % 16 != 0) return VGPMP_E_SHAPE;
    if (d->N > 1024) return VGPMP_E_SHAPE;
    return 0;
}

size_t vg_layout_workspace(const vgpmp_dims* d, void* base, vg_workspace* ws) {
    const bool real = base != nullptr;
    char* cur = real ? (char*)base : (char*)(uintptr_t)256;
    char* start = cur;
    const size_t P = d->num_problems, L = d->L, N = d->N, Mz = vg_mz(d), J = vg_j(d), S = d->S, B = d->B;
    const size_t PL = P * L;
    ws->ell = carve<double>(cur, PL, real);
    ws->var = carve<double>(cur, PL, real);
    ws->K = carve<double>(cur, PL * Mz * Mz, real);
    ws->Lk = carve<double>(cur, PL * Mz * Mz, real);
    ws->Linv = carve<double>(cur, PL * Mz * Mz, real);
    ws->Kinv = carve<double>(cur, PL * Mz * Mz, real);
    ws->Kuf = carve<double>(cur, PL * Mz * N, real);
    ws->A64 = carve<double>(cur, PL * N * Mz, real);
    ws->afull = carve<double>(cur, PL * Mz, real);
    ws->cvec = carve<double>(cur, PL * 2, real);
    ws->kl_l = carve<double>(cur, PL, real);
    ws->dA64 = carve<double>(cur, PL * N * Mz, real);
    ws->A = carve<float>(cur, PL * N * Mz, real);
    ws->C = carve<float>(cur, PL * Mz * Mz, real);
    ws->m = carve<float>(cur, PL * Mz, real);
    ws->Phi = carve<float>(cur, PL * J * B, real);
    ws->dPhi = carve<float>(cur, PL * J * B, real);
    ws->F0 = carve<float>(cur, (size_t)d->split_k * P * S * L * J, real);
    ws->H = carve<float>(cur, (size_t)d->split_k * P * S * L * J, real);
    ws->R = carve<float>(cur, P * S * L * Mz, real);
    ws->G = carve<float>(cur, P * S * L * N, real);
    ws->lik_partial = carve<float>(cur, P * (size_t)vg_loglik_blocks_per_problem(d->S, d->N), real);
    ws->part = carve<float>(cur, PL * vg_chunks(d) * vg_part_len(d), real);
    return (size_t)(cur - start) + 256;
}

int vg_workspace_lookup(const vgpmp_dims* d, const vg_workspace* ws, const char* name, void** ptr, size_t* count,
                        int32_t* is_double) {
    const size_t P = d->num_problems, L = d->L, N = d->N, Mz = vg_mz(d), J = vg_j(d), S = d->S, B = d->B;
    *is_double = 0;
    if (!strcmp(name, "A")) { *ptr = ws->A; *count = P * L * N * Mz; }
    else if (!strcmp(name, "C")) { *ptr = ws->C; *count = P * L * Mz * Mz; }
    else if (!strcmp(name, "m")) { *ptr = ws->m; *count = P * L * Mz; }
    else if (!strcmp(name, "Phi")) { *ptr = ws->Phi; *count = P * L * J * B; }
    else if (!strcmp(name, "F0")) { *ptr = ws->F0; *count = (size_t)d->split_k * P * S * L * J; }
    else if (!strcmp(name, "H")) { *ptr = ws->H; *count = (size_t)d->split_k * P * S * L * J; }
    else if (!strcmp(name, "R")) { *ptr = ws->R; *count = P * S * L * Mz; }
    else if (!strcmp(name, "G")) { *ptr = ws->G; *count = P * S * L * N; }
    else if (!strcmp(name, "kl_l")) { *ptr = ws->kl_l; *count = P * L; *is_double = 1; }
    else if (!strcmp(name, "Kinv")) { *ptr = ws->Kinv; *count = P * L * Mz * Mz; *is_double = 1; }
    else if (!strcmp(name, "Lk")) { *ptr = ws->Lk; *count = P * L * Mz * Mz; *is_double = 1; }
    else return VGPMP_E_ARG;
    return 0;
}

int vg_launch_rng(const vgpmp_dims* d, const vgpmp_noise* nz, uint32_t seed, uint32_t problem_base, uint32_t step,
                  const uint32_t* ctr, hipStream_t st) {
    const int P = d->num_problems, L = d->L, B = d->B, D = d->L, Mz = vg_mz(d);
    hipLaunchKernelGGL(rng_basis_kernel, dim3((L * B + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, L, B, D, nz->omega,
                       nz->beta, seed, problem_base, step, ctr);
    const uint32_t nW = (uint32_t)d->S * L * B, nE = (uint32_t)d->S * Mz * L;
    const uint32_t nctr = ((nW + 3) >> 2) + 2 * ((nE + 3) >> 2);
    hipLaunchKernelGGL(rng_normals_kernel, dim3((nctr + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, nW, nE, nz->w,
                       nz->eps, nz->eps2, seed, problem_base, step, ctr);
    return (int)hipGetLastError();
}

static double adam_lr_t(double lr, int t) { return lr * sqrt(1.0 - pow(0.95, t)) / (1.0 - pow(0.8, t)); }

int vg_launch_adam(const vgpmp_dims* d, const vgpmp_params* x, const vgpmp_params* g, const vgpmp_params* am,
                   const vgpmp_params* av, int trainable, double lr, int t, hipStream_t st) {
    const size_t P = d->num_problems, L = d->L, M = d->M;
    const double lr_t = adam_lr_t(lr, t);
    auto go = [&](size_t n, double* xx, const double* gg, double* mm, double* vv, int tril) {
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, xx, gg, mm,
                           vv, lr_t, tril);
    };
    if (trainable & VGPMP_TRAIN_Q_MU) go(P * L * M, x->q_mu, g->q_mu, am->q_mu, av->q_mu, 0);
    if (trainable & VGPMP_TRAIN_Q_SQRT) go(P * L * M * M, x->q_sqrt, g->q_sqrt, am->q_sqrt, av->q_sqrt, (int)M);
    if (trainable & VGPMP_TRAIN_LENGTHSCALES) go(P * L, x->raw_ell, g->raw_ell, am->raw_ell, av->raw_ell, 0);
    if (trainable & VGPMP_TRAIN_KERNEL_VARIANCE) go(P * L, x->raw_var, g->raw_var, am->raw_var, av->raw_var, 0);
    return (int)hipGetLastError();
}

static int set_dyn_lds(const void* fn, size_t bytes) {
    if (bytes > 160 * 1024) return VGPMP_E_SHAPE;
    if (bytes > 48 * 1024) VG_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}


int vg_elbo_step(const vgpmp_dims* d, const vgpmp_robot* rb, const vgpmp_sdf* sdf, const vgpmp_problem* pb,
                 const vgpmp_params* params, const vgpmp_params* am, const vgpmp_params* av, const vgpmp_noise* nz,
                 const vgpmp_outputs* out, const vg_workspace* ws, int what, int trainable, double lr, int adam_t,
                 uint32_t seed, uint32_t problem_base, uint32_t step, hipStream_t st, hipEvent_t* ev) {
    const int P = d->num_problems, S = d->S, N = d->N, M = d->M, L = d->L, B = d->B, Mz = M + 2, J = N + Mz;
    int evi = 0;
    auto mark = [&]() { if (ev) (void)hipEventRecord(ev[evi++], st); };
    uint32_t* ctr = pb->step_counter;
    mark();
    const int SK = d->split_k, NC = vg_chunks(d);
    const bool backward = (what & VGPMP_DO_BACKWARD) != 0;
    const bool want_dell = backward && (trainable & VGPMP_TRAIN_LENGTHSCALES);
    int rc;
    if (what & VGPMP_GEN_NOISE) {
        rc = vg_launch_rng(d, nz, seed, problem_base, step, ctr, st);
        if (rc) return rc;
    }
    mark();
    // ---- covariance path (float64)
    CovArgs ca;
    ca.N = N; ca.M = M; ca.L = L; ca.D = L;
    ca.X = pb->X; ca.Zy = pb->Zy; ca.y_u = pb->y_u;
    ca.jitter = pb->jitter; ca.kl_scale = pb->kl_scale;
    ca.q_mu = params->q_mu; ca.q_sqrt = params->q_sqrt; ca.raw_ell = params->raw_ell; ca.raw_var = params->raw_var;
    ca.ws = *ws;
    const size_t lds_cov = ((size_t)3 * Mz * (Mz + 1) + 2 * Mz) * sizeof(double);
    rc = set_dyn_lds((const void*)cov_fwd_kernel, lds_cov);
    if (rc) return rc;
    hipLaunchKernelGGL(cov_fwd_kernel, dim3(L, P), dim3(kBlock), lds_cov, st, ca);
    mark();
    // ---- features and prior GEMM
    hipLaunchKernelGGL(features_kernel, dim3((B + kBlock - 1) / kBlock, J, P * L), dim3(kBlock), 0, st, N, Mz, L, L, B,
                       pb->X, pb->Zy, params->raw_ell, params->raw_var, nz->omega, nz->beta, ws->Phi,
                       want_dell ? ws->dPhi : (float*)nullptr, (what & VGPMP_DO_ADAM) ? ctr : (uint32_t*)nullptr);
    mark();
    const size_t slab = (size_t)P * S * L * J;
    const int nsel = want_dell ? 2 : 1;
    hipLaunchKernelGGL(prior_gemm_kernel, dim3((J + 16 * kNT - 1) / (16 * kNT), (S + 63) / 64, P * L * SK * nsel),
                       dim3(kBlock), 0, st, S, L, J, B, SK, nsel, nz->w, ws->Phi, ws->dPhi, ws->F0, ws->H, slab);
    mark();
    // ---- path assembly
    PathArgs pa;
    pa.S = S; pa.N = N; pa.Mz = Mz; pa.L = L; pa.SK = SK; pa.slab = slab;
    pa.sqrt_jitter = (float)sqrt(pb->jitter);
    pa.A = ws->A; pa.C = ws->C; pa.m = ws->m; pa.F0 = ws->F0; pa.H = want_dell ? ws->H : nullptr;
    pa.eps = nz->eps; pa.eps2 = nz->eps2; pa.R = ws->R; pa.f = out->f; pa.G = ws->G; pa.part = ws->part;
    pa.part_len = vg_part_len(d); pa.NC = NC;
    const size_t lds_pf = ((size_t)Mz * (Mz + 1) + (size_t)N * (Mz + 1) + (size_t)VG_SC * (Mz + 1)) * sizeof(float);
    rc = set_dyn_lds((const void*)paths_fwd_kernel, lds_pf);
    if (rc) return rc;
    hipLaunchKernelGGL(paths_fwd_kernel, dim3(NC, L, P), dim3(kBlock), lds_pf, st, pa);
    mark();
    // ---- likelihood forward + reverse (fk_sdf.hip)
    const double lik_scale = pb->alpha / (double)d->S_total;
    int nblk = 0;
    rc = vg_launch_loglik_paths(rb, sdf, out->f, P, S, L, N, (float)(-lik_scale), ws->G, out->logp, ws->lik_partial,
                                &nblk, st);
    if (rc) return rc;
    mark();
    if (!backward) {
        hipLaunchKernelGGL(elbo_pieces_kernel, dim3(P), dim3(64), 0, st, L, nblk, ws->lik_partial, ws->kl_l, lik_scale,
                           pb->kl_scale, out->lik, out->kl);
        return (int)hipGetLastError();
    }
    // ---- reverse of the path assembly, then of the covariance path (+ Adam)
    const size_t lds_pb = ((size_t)N * (Mz + 1) + (size_t)VG_SC * (N + 1) + (size_t)3 * VG_SC * (Mz + 1)) * sizeof(float);
    rc = set_dyn_lds((const void*)paths_bwd_kernel, lds_pb);
    if (rc) return rc;
    hipLaunchKernelGGL(paths_bwd_kernel, dim3(NC, L, P), dim3(kBlock), lds_pb, st, pa);
    mark();
    CovBwdArgs cb;
    cb.c = ca; cb.NC = NC; cb.part_len = vg_part_len(d); cb.nblk = nblk; cb.lik_scale = lik_scale;
    cb.out_lik = out->lik; cb.out_kl = out->kl;
    cb.g_qmu = out->grad.q_mu; cb.g_qsqrt = out->grad.q_sqrt; cb.g_ell = out->grad.raw_ell; cb.g_var = out->grad.raw_var;
    cb.do_adam = (what & VGPMP_DO_ADAM) ? 1 : 0; cb.trainable = trainable; cb.want_dell = want_dell ? 1 : 0;
    cb.lr_t = cb.do_adam ? adam_lr_t(lr, adam_t) : 0.0;
    cb.lr = lr; cb.ctr = ctr;
    cb.mq_mu = am ? am->q_mu : nullptr; cb.mq_sqrt = am ? am->q_sqrt : nullptr;
    cb.m_ell = am ? am->raw_ell : nullptr; cb.m_var = am ? am->raw_var : nullptr;
    cb.vq_mu = av ? av->q_mu : nullptr; cb.vq_sqrt = av ? av->q_sqrt : nullptr;
    cb.v_ell = av ? av->raw_ell : nullptr; cb.v_var = av ? av->raw_var : nullptr;
    cb.pq_mu = params->q_mu; cb.pq_sqrt = params->q_sqrt; cb.p_ell = params->raw_ell; cb.p_var = params->raw_var;
    const size_t lds_cb = ((size_t)4 * Mz * (Mz + 1) + 3 * Mz + 8) * sizeof(double);
    rc = set_dyn_lds((const void*)cov_bwd_kernel, lds_cb);
    if (rc) return rc;
    hipLaunchKernelGGL(cov_bwd_kernel, dim3(L, P), dim3(kBlock), lds_cb, st, cb);
    mark();
    return (int)hipGetLastError();
}
