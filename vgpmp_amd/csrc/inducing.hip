// Reverse pass wrt the inducing locations (trainable_params.inducing_variable, utils/miscellaneous.py:338;
// Z = 0.09 + 0.82 sigmoid(raw_Z), models/vgpmp.py:29-42).  The ELBO step itself carries the two kernel hyper-parameters
// through the Cholesky factorisation in FORWARD mode; M inducing times per latent are too many tangents, so this path is
// the reverse form the oracle states (oracle/vgpmp_oracle.py::elbo_backward, want_z):
//   z_reduce   per (sample chunk, latent): dR = G A, dA += G^T R, dC += dR^T eps          (float32, like the step's sums)
//   z_rff      per (64 bases, latent): T = dR^T W, then coef T sin(arg) omega  -> d loss / d Zy[:, :] through the prior draw
//   z_cov      per latent, float64 in LDS: dKfu = dA Kinv, dKj = -A^T dA Kinv + KL terms + Cholesky adjoint of
//              dLk = dC pad(Q)^T - ddelta a^T, then d k / d z                               -> d loss / d Zy[:, l]
//   z_update   per problem: sum of the parts, chain rule through the bijector, Adam, next Zy
// One launch per kernel: the mode is off by default in the reference, nothing here is on the tuned path.
#include "vgpmp_device.h"
#include "gp_path.h"
#include "gp_math.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSC = 8;        // samples per chunk (vg_sc)

template <typename T>
T* carve(char*& cur, size_t count, bool real) {
    uintptr_t v = (uintptr_t)cur;
    v = (v + 255) & ~(uintptr_t)255;
    T* out = real ? (T*)v : nullptr;
    cur = (char*)(v + count * sizeof(T));
    return out;
}

// grid (P): Zy[p] = [0; 1; Z]  (inducing_variables.py:73-82)
__global__ __launch_bounds__(kBlock) void z_build_kernel(const double* __restrict__ raw_Z, int M, int L, double* __restrict__ Zy) {
    const int p = blockIdx.x, Mz = M + 2;
    for (int e = threadIdx.x; e < Mz * L; e += kBlock) {
        const int i = e / L, l = e - i * L;
        Zy[(size_t)p * Mz * L + e] = i == 0 ? 0.0 : i == 1 ? 1.0 : kZLow + (kZHigh - kZLow) * sigmoid_d(raw_Z[((size_t)p * M + i - 2) * L + l]);
    }
}

struct ReduceArgs {
    int S, N, Mz, L, NC;
    const float4* A4;      // [P, L, N, Mz]  .x = A
    const float *G, *R, *eps;
    vg_ind_scratch sc;
};
// grid (NC, L, P).  LDS: A [N][Mz], G rows [kSC][N], R / eps / dR rows [kSC][Mz]
__global__ __launch_bounds__(kBlock) void z_reduce_kernel(ReduceArgs a) {
    extern __shared__ float zlds[];
    const int ch = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L;
    const size_t pl = (size_t)p * L + l;
    float* As = zlds;                       // [N][Mz]
    float* Gs = As + N * Mz;                // [kSC][N]
    float* Rs = Gs + kSC * N;               // [kSC][Mz]
    float* Es = Rs + kSC * Mz;              // [kSC][Mz]
    float* Ds = Es + kSC * Mz;              // [kSC][Mz]
    const int s0 = ch * kSC;
    for (int e = tid; e < N * Mz; e += kBlock) As[e] = a.A4[pl * N * Mz + e].x;
    for (int e = tid; e < kSC * N; e += kBlock) {
        const int sl = e / N, n = e - sl * N, s = s0 + sl;
        Gs[e] = s < S ? a.G[(((size_t)p * S + s) * L + l) * N + n] : 0.f;
    }
    for (int e = tid; e < kSC * Mz; e += kBlock) {
        const int sl = e / Mz, m = e - sl * Mz, s = s0 + sl;
        Rs[e] = s < S ? a.R[(((size_t)p * S + s) * L + l) * Mz + m] : 0.f;
        Es[e] = s < S ? a.eps[(((size_t)p * S + s) * Mz + m) * L + l] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < kSC * Mz; e += kBlock) {              // dR = G A
        const int sl = e / Mz, m = e - sl * Mz, s = s0 + sl;
        float t = 0.f;
#pragma unroll 8
        for (int n = 0; n < N; ++n) t = fmaf(Gs[sl * N + n], As[n * Mz + m], t);
        Ds[e] = t;
        if (s < S) a.sc.dR[(((size_t)p * S + s) * L + l) * Mz + m] = t;
    }
    __syncthreads();
    float* dA = a.sc.dA + (pl * a.NC + ch) * (size_t)N * Mz;
    for (int e = tid; e < N * Mz; e += kBlock) {                // dA = G^T R over the chunk
        const int n = e / Mz, m = e - n * Mz;
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < kSC; ++sl) t = fmaf(Gs[sl * N + n], Rs[sl * Mz + m], t);
        dA[e] = t;
    }
    float* dC = a.sc.dC + (pl * a.NC + ch) * (size_t)Mz * Mz;
    for (int e = tid; e < Mz * Mz; e += kBlock) {               // dC = dR^T eps over the chunk
        const int i = e / Mz, k = e - i * Mz;
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < kSC; ++sl) t = fmaf(Ds[sl * Mz + i], Es[sl * Mz + k], t);
        dC[e] = t;
    }
}

struct RffArgs {
    int S, Mz, L, D, B;
    const float *W, *omega, *beta;
    const double *Zy, *ell, *var;
    vg_ind_scratch sc;
};
// grid (B / kIndBChunk, L, P), 256 threads = 4 waves: lane = basis of the chunk, wave w owns rows w, w + 4, ... of Zy
constexpr int kRffS = 64, kRffRows = (VGPMP_MAX_MZ + 3) / 4;
__global__ __launch_bounds__(kBlock) void z_rff_kernel(RffArgs a) {
    __shared__ __attribute__((aligned(16))) float dRs[kRffS * VGPMP_MAX_MZ];
    __shared__ float zys[VGPMP_MAX_MZ * VGPMP_MAX_DOF];
    __shared__ __attribute__((aligned(16))) float Ws[kRffS * kIndBChunk];
    const int cb = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int S = a.S, Mz = a.Mz, L = a.L, D = a.D, B = a.B;
    const size_t pl = (size_t)p * L + l;
    const int b = cb * kIndBChunk + lane;
    for (int e = tid; e < Mz * D; e += kBlock) zys[e] = (float)a.Zy[(size_t)p * Mz * D + e];
    const float ell = (float)a.ell[pl], inv_ell = 1.0f / ell;
    const float coef = __builtin_amdgcn_sqrtf(2.0f * (float)a.var[pl] / (float)B) * inv_ell;
    float om[VGPMP_MAX_DOF];
#pragma unroll
    for (int d = 0; d < VGPMP_MAX_DOF; ++d) {          // (unconditional loads on a clamped index: a conditional load is a branch
        const float v = a.omega[(pl * B + b) * D + min(d, D - 1)];      //  and a round trip of its own)
        om[d] = d < D ? v : 0.f;
    }
    const float bt = a.beta[pl * B + b];
    float acc[kRffRows];
    int mrow[kRffRows];
#pragma unroll
    for (int k = 0; k < kRffRows; ++k) { acc[k] = 0.f; mrow[k] = min(wv + 4 * k, Mz - 1); }
    for (int sb = 0; sb < S; sb += kRffS) {                     // T[m, b] = sum_s dR[s, m] w[s, b]
        __syncthreads();
        // the batch's weights [kRffS][64 bases] and dR rows by DMA, all requests in flight together (a load per sample and lane
        // inside the product loop left S round trips exposed: 94 us per launch at config 2)
        vg_stage_rows(Ws, kRffS, kIndBChunk, tid, kBlock, [&](int r) -> const float* {
            const int s = sb + r;
            return s < S ? a.W + (((size_t)p * S + s) * L + l) * B + (size_t)cb * kIndBChunk : nullptr;
        });
        vg_stage_rows(dRs, kRffS, Mz, tid, kBlock, [&](int r) -> const float* {
            const int s = sb + r;
            return s < S ? a.sc.dR + (((size_t)p * S + s) * L + l) * Mz : nullptr;      // zero rows beyond S
        });
        vg_dma_wait();
        __syncthreads();
        // (rows beyond S are zeros in both tiles: a constant trip count, eight samples' reads in flight)
#pragma unroll 8
        for (int sl = 0; sl < kRffS; ++sl) {
            const float w = Ws[sl * kIndBChunk + lane];
            // (no test of m against Mz here: a branch per row put each LDS read and its wait in a block of its own -- a thousand
            //  serial LDS round trips per wave, two thirds of the kernel; rows beyond Mz read a clamped row into sums nobody uses)
#pragma unroll
            for (int k = 0; k < kRffRows; ++k) acc[k] = fmaf(dRs[sl * Mz + mrow[k]], w, acc[k]);
        }
    }
    // d / d Zy[m, d] = sum over the chunk's bases of ts[m, b] om[b, d]: ts and om through LDS, one output per thread with its
    // 64 terms in basis order (a wave-wide butterfly per output -- 56 of them, six LDS-crossbar steps each -- took 16 us)
    __syncthreads();
    constexpr int kTsLd = kIndBChunk + 1;
    float* tss = Ws;                                             // [Mz][kTsLd]
    float* oms = dRs;                                            // [64][D]
    static_assert(VGPMP_MAX_MZ * kTsLd <= kRffS * kIndBChunk && kIndBChunk * VGPMP_MAX_DOF <= kRffS * VGPMP_MAX_MZ, "tail tiles fit");
#pragma unroll
    for (int k = 0; k < kRffRows; ++k) {
        const int m = wv + 4 * k;
        if (m >= Mz) break;                                      // (uniform over the wave)
        float proj = 0.f;
#pragma unroll
        for (int d = 0; d < VGPMP_MAX_DOF; ++d)
            if (d < D) proj = fmaf(zys[m * D + d], om[d], proj);
        // the features' own evaluation (features_body): hardware sin of the argument in revolutions
        const float rev = __builtin_amdgcn_fractf((proj * inv_ell + bt) * 0.15915494309189535f);
        tss[m * kTsLd + lane] = acc[k] * __builtin_amdgcn_sinf(rev) * coef;
    }
    if (wv == 0) {
#pragma unroll
        for (int d = 0; d < VGPMP_MAX_DOF; ++d)
            if (d < D) oms[lane * D + d] = om[d];
    }
    __syncthreads();
    float* out = a.sc.rff + (pl * (B / kIndBChunk) + cb) * (size_t)Mz * D;
    for (int e = tid; e < Mz * D; e += kBlock) {
        const int m = e / D, d = e - m * D;
        float t = 0.f;
#pragma unroll 16
        for (int b2 = 0; b2 < kIndBChunk; ++b2) t = fmaf(tss[m * kTsLd + b2], oms[b2 * D + d], t);
        out[e] = t;
    }
}

struct CovArgs {
    int N, M, L, D, NC;
    const double *X, *Zy, *y_u, *q_mu, *q_sqrt, *ell, *var;
    const double *Kinv, *Lk, *Li, *K;
    double jitter;
    vg_ind_scratch sc;
};
constexpr int kTN = kIndRowTile;       // rows of X per workgroup of the rows kernel
// ---- the part of the covariance reverse pass that runs over the time points, a tile of kTN of them per workgroup (it was a
// serial loop inside z_cov_kernel: seven passes of four barriers on ONE workgroup per latent, 187 us per launch at config 2):
// A rows = Kfu Kinv, dKfu rows = dA Kinv, the tile's A^T dA and its part of d / d Zy through Kfu.  grid (tiles, L, P)
__global__ __launch_bounds__(kBlock) void z_cov_rows_kernel(CovArgs a) {
    extern __shared__ double cl[];
    const int t = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x;
    const int N = a.N, M = a.M, Mz = M + 2, L = a.L, D = a.D, ld = Mz + 1, tiles = gridDim.x;
    const size_t pl = (size_t)p * L + l, mm = (size_t)Mz * Mz;
    double* Kinv = cl;                    // (K + jI)^-1
    double* kf = Kinv + Mz * ld;          // [kTN][Mz] Kfu rows, then dKfu rows
    double* at = kf + kTN * Mz;           // [kTN][Mz] A rows
    double* da = at + kTN * Mz;           // [kTN][Mz] dA rows
    double* zs = da + kTN * Mz;           // [Mz]
    double* xs = zs + Mz;                 // [kTN]
    const double ell = a.ell[pl], var = a.var[pl];
    const int n0 = t * kTN, nr = min(kTN, N - n0);
    for (int e = tid; e < Mz * Mz; e += kBlock) Kinv[(e / Mz) * ld + e % Mz] = a.Kinv[pl * mm + e];
    for (int i = tid; i < Mz; i += kBlock) zs[i] = a.Zy[((size_t)p * Mz + i) * D + l];
    for (int r = tid; r < nr; r += kBlock) xs[r] = a.X[(size_t)(n0 + r) * D + l];
    for (int e = tid; e < nr * Mz; e += kBlock) {              // dA: the chunks in order, eight requests at a time
        const float* src = a.sc.dA + ((pl * a.NC) * N + n0) * (size_t)Mz + e;
        double c = 0.0;
        for (int ch0 = 0; ch0 < a.NC; ch0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(ch0 + u, a.NC - 1) * N * Mz];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (ch0 + u < a.NC) c += (double)v[u];
        }
        da[e] = c;
    }
    __syncthreads();
    for (int e = tid; e < nr * Mz; e += kBlock) kf[e] = matern52(xs[e / Mz], zs[e % Mz], ell, var);
    __syncthreads();
    for (int e = tid; e < nr * Mz; e += kBlock) {               // A rows = Kfu Kinv
        const int r = e / Mz, i = e - r * Mz;
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < Mz; ++k) s = fma(kf[r * Mz + k], Kinv[k * ld + i], s);
        at[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < nr * Mz; e += kBlock) {               // dKfu rows = dA Kinv (into kf)
        const int r = e / Mz, i = e - r * Mz;
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < Mz; ++k) s = fma(da[r * Mz + k], Kinv[k * ld + i], s);
        kf[e] = s;
    }
    double* mt = a.sc.mt_part + (pl * tiles + t) * mm;
    for (int e = tid; e < Mz * Mz; e += kBlock) {               // the tile's A^T dA
        const int i = e / Mz, j = e - i * Mz;
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < nr; ++r) s = fma(at[r * Mz + i], da[r * Mz + j], s);
        mt[e] = s;
    }
    __syncthreads();
    for (int i = tid; i < Mz; i += kBlock) {                    // Kfu[n, i] = k(x_n, z_i): d / d z_i
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < nr; ++r) s = fma(kf[r * Mz + i], matern52_d1(zs[i], xs[r], ell, var), s);
        a.sc.gz_part[(pl * tiles + t) * Mz + i] = s;
    }
}

// grid (L, P), float64 matrices [Mz][ld] in LDS: the tiles' sums, then the KL and Cholesky reverse pass
__global__ __launch_bounds__(kBlock) void z_cov_kernel(CovArgs a) {
    extern __shared__ double cl[];
    const int l = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
    const int N = a.N, M = a.M, Mz = M + 2, L = a.L, D = a.D, ld = Mz + 1, tiles = (N + kTN - 1) / kTN;
    const size_t pl = (size_t)p * L + l, mm = (size_t)Mz * Mz;
    double* Kinv = cl;                    // (K + jI)^-1
    double* Lk = Kinv + Mz * ld;          // chol(K + jI)
    double* Li = Lk + Mz * ld;            // Lk^-1
    double* dKj = Li + Mz * ld;
    double* dLk = dKj + Mz * ld;          // also: dC, then tril(dLk), then Pm
    double* T1 = dLk + Mz * ld;           // scratch product
    double* Mt = T1 + Mz * ld;            // A^T dA
    double* zs = Mt + Mz * ld;            // [Mz]
    double* mv = zs + Mz;                 // [Mz] m = [y_u; q_mu]
    double* af = mv + Mz;                 // [Mz] a_full
    double* dd = af + Mz;                 // [Mz] ddelta
    double* gz = dd + Mz;                 // [Mz]
    double* kc0 = gz + Mz;                // [Mz] columns 0 and 1 of K + jI (read from memory inside the loops below they were
    double* kc1 = kc0 + Mz;               //      a round trip per element, most of them on one thread)
    __shared__ double cv[2];
    const double ell = a.ell[pl], var = a.var[pl], jit = a.jitter;
    auto Kj = [&](int i, int j) { return j == 0 ? kc0[i] : kc1[i]; };      // K + jI, columns 0 and 1
    for (int e = tid; e < Mz * Mz; e += kBlock) {
        const int i = e / Mz, j = e - i * Mz;
        Kinv[i * ld + j] = a.Kinv[pl * mm + e];
        Lk[i * ld + j] = a.Lk[pl * mm + e];
        Li[i * ld + j] = a.Li[pl * mm + e];
        double mt = 0.0;                                         // A^T dA: the tiles in order
#pragma unroll 8
        for (int t = 0; t < tiles; ++t) mt += a.sc.mt_part[(pl * tiles + t) * mm + e];
        Mt[i * ld + j] = mt;
        const float* src = a.sc.dC + (pl * a.NC) * mm + e;       // dC: the chunks in order, eight requests at a time
        double c = 0.0;
        for (int ch0 = 0; ch0 < a.NC; ch0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(ch0 + u, a.NC - 1) * mm];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (ch0 + u < a.NC) c += (double)v[u];
        }
        T1[i * ld + j] = c;
    }
    // q_sqrt into the space dKj takes later (read from memory inside the product below it cost a round trip per term)
    for (int e = tid; e < M * M; e += kBlock) dKj[(e / M) * ld + e % M] = a.q_sqrt[pl * M * M + e];
    for (int i = tid; i < Mz; i += kBlock) {
        zs[i] = a.Zy[((size_t)p * Mz + i) * D + l];
        mv[i] = i < 2 ? a.y_u[((size_t)p * 2 + i) * L + l] : a.q_mu[pl * M + i - 2];
        double g = 0.0;
#pragma unroll 8
        for (int t = 0; t < tiles; ++t) g += a.sc.gz_part[(pl * tiles + t) * Mz + i];
        gz[i] = g;
        kc0[i] = a.K[pl * mm + (size_t)i * Mz] + (i == 0 ? jit : 0.0);
        kc1[i] = a.K[pl * mm + (size_t)i * Mz + 1] + (i == 1 ? jit : 0.0);
    }
    __syncthreads();
    // dLk = dC pad(Q)^T: column j < 2 of pad(Q)^T rows is zero;  dLk[i, j] = sum_{k >= 2} dC[i, k] Q[j-2, k-2] (k <= j)
    for (int e = tid; e < Mz * Mz; e += kBlock) {
        const int i = e / Mz, j = e - i * Mz;
        double t = 0.0;
        if (j >= 2)
#pragma unroll 8
            for (int k = 2; k <= j; ++k) t = fma(T1[i * ld + k], dKj[(j - 2) * ld + (k - 2)], t);      // q_sqrt from its LDS copy
        dLk[i * ld + j] = t;
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += kBlock) {               // dKj = -(A^T dA) Kinv
        const int i = e / Mz, j = e - i * Mz;
        double t = 0.0;
#pragma unroll 8
        for (int k = 0; k < Mz; ++k) t = fma(Mt[i * ld + k], Kinv[k * ld + j], t);
        dKj[i * ld + j] = -t;
    }
    // ---- KL (kullback_leiblers/prior_kl.py:16-35): p_mu = Kj[:, :2] c, c = Kj[:2, :2]^-1 y;  a = Lk^-1 (m - p_mu)
    if (tid == 0) {
        const double a00 = Kj(0, 0), a01 = Kj(0, 1), a10 = Kj(1, 0), a11 = Kj(1, 1), det = a00 * a11 - a01 * a10;
        cv[0] = (a11 * mv[0] - a01 * mv[1]) / det;
        cv[1] = (a00 * mv[1] - a10 * mv[0]) / det;
    }
    __syncthreads();
    const double c0 = cv[0], c1 = cv[1];
    for (int i = tid; i < Mz; i += kBlock) dd[i] = mv[i] - (Kj(i, 0) * c0 + Kj(i, 1) * c1);      // delta
    __syncthreads();
    for (int i = tid; i < Mz; i += kBlock) {
        double t = 0.0;
#pragma unroll 8
        for (int k = 0; k <= i; ++k) t = fma(Li[i * ld + k], dd[k], t);
        af[i] = t;
    }
    __syncthreads();
    for (int i = tid; i < Mz; i += kBlock) {                    // ddelta = Lk^-T abar, abar = a with its first two zeroed
        double t = 0.0;
#pragma unroll 8
        for (int k = max(i, 2); k < Mz; ++k) t = fma(Li[k * ld + i], af[k], t);
        dd[i] = t;
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += kBlock) {
        const int i = e / Mz, j = e - i * Mz;
        dLk[i * ld + j] -= dd[i] * af[j];
        if (j < 2) dKj[i * ld + j] -= dd[i] * (j == 0 ? c0 : c1);      // d p_mu = -ddelta
    }
    __syncthreads();
    if (tid == 0) {
        double dc0 = 0.0, dc1 = 0.0;                            // dc = Kj[:, :2]^T dp_mu
        for (int i = 0; i < Mz; ++i) { dc0 -= Kj(i, 0) * dd[i]; dc1 -= Kj(i, 1) * dd[i]; }
        const double a00 = Kj(0, 0), a01 = Kj(0, 1), a10 = Kj(1, 0), a11 = Kj(1, 1), det = a00 * a11 - a01 * a10;
        const double t0 = (a11 * dc0 - a10 * dc1) / det, t1 = (a00 * dc1 - a01 * dc0) / det;   // Kj[:2, :2]^-T dc
        dKj[0] -= t0 * c0; dKj[1] -= t0 * c1; dKj[ld] -= t1 * c0; dKj[ld + 1] -= t1 * c1;
    }
    __syncthreads();
    // ---- Cholesky adjoint (Murray 2016): Pm = tril(Lk^T tril(dLk)), diagonal halved;  dKj += sym(Lk^-T Pm Lk^-1)
    for (int e = tid; e < Mz * Mz; e += kBlock) {
        const int i = e / Mz, j = e - i * Mz;
        double t = 0.0;
        if (j <= i) {
#pragma unroll 8
            for (int k = i; k < Mz; ++k) t = fma(Lk[k * ld + i], dLk[k * ld + j], t);      // rows k >= i (Lk lower), k >= j holds
            if (i == j) t *= 0.5;
        }
        T1[i * ld + j] = t;                                      // Pm
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += kBlock) {               // Mt = Pm Li   (Pm lower: k <= i; Li lower: k >= j)
        const int i = e / Mz, j = e - i * Mz;
        double t = 0.0;
#pragma unroll 8
        for (int k = j; k <= i; ++k) t = fma(T1[i * ld + k], Li[k * ld + j], t);
        Mt[i * ld + j] = t;
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += kBlock) {               // Sm = Li^T Mt
        const int i = e / Mz, j = e - i * Mz;
        double t = 0.0;
#pragma unroll 8
        for (int k = i; k < Mz; ++k) t = fma(Li[k * ld + i], Mt[k * ld + j], t);
        dLk[i * ld + j] = t;
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += kBlock) {
        const int i = e / Mz, j = e - i * Mz;
        T1[i * ld + j] = dKj[i * ld + j] + 0.5 * (dLk[i * ld + j] + dLk[j * ld + i]);
    }
    __syncthreads();
    // ---- Kuu[i, j] = k(z_i, z_j): z_i sits in row i and in column i
    for (int i = tid; i < Mz; i += kBlock) {
        double t = gz[i];
#pragma unroll 8
        for (int j = 0; j < Mz; ++j) t = fma(T1[i * ld + j] + T1[j * ld + i], matern52_d1(zs[i], zs[j], ell, var), t);
        a.sc.cov[pl * Mz + i] = t;
    }
}

struct UpdArgs {
    int M, L, NB;
    double *raw_Z, *m_Z, *v_Z, *g_Z;
    vg_ind_scratch sc;
    int do_adam;
    const uint32_t* ctr;
    double lr, lr_t;
};
// grid (P): d loss / d raw_Z, Adam (the next evaluation rebuilds Zy from raw_Z)
__global__ __launch_bounds__(kBlock) void z_update_kernel(UpdArgs a) {
    const int p = blockIdx.x, M = a.M, L = a.L, Mz = M + 2;
    const double lr_t = a.ctr ? adam_step_size(a.lr, (double)*a.ctr) : a.lr_t;
    for (int e = threadIdx.x; e < M * L; e += kBlock) {
        const int m = e / L, d = e - m * L, i = m + 2;
        double g = a.sc.cov[((size_t)p * L + d) * Mz + i];
        for (int l = 0; l < L; ++l)
            for (int cb0 = 0; cb0 < a.NB; cb0 += 8) {          // eight partials requested together, added in order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = a.sc.rff[((((size_t)p * L + l) * a.NB + min(cb0 + u, a.NB - 1)) * Mz + i) * L + d];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (cb0 + u < a.NB) g += (double)v[u];
            }
        const size_t o = (size_t)p * M * L + e;
        double raw = a.raw_Z[o];
        const double sg = sigmoid_d(raw);
        g *= (kZHigh - kZLow) * sg * (1.0 - sg);
        a.g_Z[o] = g;
        if (a.do_adam) {
            adam_update(&raw, a.m_Z + o, a.v_Z + o, g, lr_t);
            a.raw_Z[o] = raw;
        }
    }
}

}  // namespace

size_t vg_layout_ind_scratch(const vgpmp_dims* d, void* base, vg_ind_scratch* out) {
    char* cur = (char*)base;
    const bool real = base != nullptr;
    const size_t P = (size_t)d->num_problems, L = d->L, N = d->N, Mz = vg_mz(d), NC = vg_chunks(d), S = d->S;
    out->dA = carve<float>(cur, P * L * NC * N * Mz, real);
    out->dC = carve<float>(cur, P * L * NC * Mz * Mz, real);
    out->dR = carve<float>(cur, P * S * L * Mz, real);
    out->rff = carve<float>(cur, P * L * (d->B / kIndBChunk) * Mz * L, real);
    out->cov = carve<double>(cur, P * L * Mz, real);
    const size_t tiles = (N + kIndRowTile - 1) / kIndRowTile;
    out->mt_part = carve<double>(cur, P * L * tiles * Mz * Mz, real);
    out->gz_part = carve<double>(cur, P * L * tiles * Mz, real);
    return (size_t)(cur - (char*)base) + 256;
}

int vg_launch_inducing_build(const vgpmp_dims* d, const vgpmp_inducing_params* ind, hipStream_t st) {
    hipLaunchKernelGGL(z_build_kernel, dim3(d->num_problems), dim3(kBlock), 0, st, ind->raw_Z, d->M, d->L, ind->Zy);
    return (int)hipGetLastError();
}

int vg_launch_inducing_backward(const vg_ind_launch& a, hipStream_t st) {
    const vgpmp_dims* d = a.d;
    const int P = d->num_problems, S = d->S, N = d->N, M = d->M, Mz = M + 2, L = d->L, B = d->B, NC = vg_chunks(d);
    vg_ind_scratch sc;
    vg_layout_ind_scratch(d, a.ind->scratch, &sc);
    ReduceArgs r;
    r.S = S; r.N = N; r.Mz = Mz; r.L = L; r.NC = NC;
    r.A4 = reinterpret_cast<const float4*>(a.ws->A4); r.G = a.ws->G; r.R = a.ws->R; r.eps = a.nz->eps; r.sc = sc;
    const size_t lds_r = ((size_t)N * Mz + (size_t)kSC * N + 3 * (size_t)kSC * Mz) * sizeof(float);
    int rc = vg_grant_dyn_lds((const void*)z_reduce_kernel, lds_r);
    if (rc) return rc;
    hipLaunchKernelGGL(z_reduce_kernel, dim3(NC, L, P), dim3(kBlock), lds_r, st, r);
    RffArgs f;
    f.S = S; f.Mz = Mz; f.L = L; f.D = L; f.B = B;
    f.W = a.nz->w; f.omega = a.nz->omega; f.beta = a.nz->beta; f.Zy = a.ind->Zy; f.ell = a.ws->ell; f.var = a.ws->var; f.sc = sc;
    hipLaunchKernelGGL(z_rff_kernel, dim3(B / kIndBChunk, L, P), dim3(kBlock), 0, st, f);
    CovArgs c;
    c.N = N; c.M = M; c.L = L; c.D = L; c.NC = NC;
    c.X = a.X; c.Zy = a.ind->Zy; c.y_u = a.y_u; c.q_mu = a.params->q_mu; c.q_sqrt = a.params->q_sqrt;
    c.ell = a.ws->ell; c.var = a.ws->var; c.Kinv = a.ws->Kinv; c.Lk = a.ws->Lk64; c.Li = a.ws->Li64; c.K = a.ws->Ks64;
    c.jitter = a.jitter; c.sc = sc;
    const size_t lds_cr = ((size_t)Mz * (Mz + 1) + (size_t)3 * kTN * Mz + (size_t)Mz + kTN) * sizeof(double);
    rc = vg_grant_dyn_lds((const void*)z_cov_rows_kernel, lds_cr);
    if (rc) return rc;
    hipLaunchKernelGGL(z_cov_rows_kernel, dim3((N + kTN - 1) / kTN, L, P), dim3(kBlock), lds_cr, st, c);
    const size_t lds_c = ((size_t)7 * Mz * (Mz + 1) + (size_t)7 * Mz) * sizeof(double);
    rc = vg_grant_dyn_lds((const void*)z_cov_kernel, lds_c);
    if (rc) return rc;
    hipLaunchKernelGGL(z_cov_kernel, dim3(L, P), dim3(kBlock), lds_c, st, c);
    UpdArgs u;
    u.M = M; u.L = L; u.NB = B / kIndBChunk;
    u.raw_Z = a.ind->raw_Z; u.m_Z = a.ind->m_Z; u.v_Z = a.ind->v_Z; u.g_Z = a.ind->g_Z; u.sc = sc;
    u.do_adam = (a.do_adam && (a.trainable & VGPMP_TRAIN_INDUCING)) ? 1 : 0;
    u.ctr = a.ctr; u.lr = a.lr; u.lr_t = a.lr_t;
    hipLaunchKernelGGL(z_update_kernel, dim3(P), dim3(kBlock), 0, st, u);
    return (int)hipGetLastError();
}
