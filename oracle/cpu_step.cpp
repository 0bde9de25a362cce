// oracle/cpu_step.cpp -- TEST INFRASTRUCTURE AND CPU BASELINE, never part of the product.
//
// A second, independent float64 restatement of ONE optimisation step of the reference's hot path:
// loss = -ELBO of models/vgpmp.py:265-289 with injected randomness, its gradient wrt the unconstrained variables, and the Keras
// Adam update of utils/miscellaneous.py:68-84 -- plain C++ loops with OpenMP over latents and samples.  Two uses, both outside
// the product (only tests/, __graft_entry__ and bench.py's cpu_baseline may load what this builds; vgpmp_amd/ never does):
//   * a CPU baseline a reader believes: the NumPy oracle is element-wise array code (6 steps/s at BASELINE config 2); this is what
//     a compiled, threaded CPU implementation of the same arithmetic does on the same host cores (bench.py: cpu_baseline.openmp);
//   * a cross-check of oracle/vgpmp_oracle.py: written from SURVEY Appendix A and the reference files cited below, not from the
//     NumPy code -- tests/test_oracle_cpu_step.py holds the two against each other (loss 1e-10, gradients 1e-8 of their largest
//     entry: the covariance path has a condition number of ~1e7, two float64 factorisations agree no further).
// PARITY UNPINNED for the GP half (A2-A7, A11-A13), exactly as the NumPy oracle: the reference holds no vector for it.
//
// Build: oracle/Makefile -> oracle/_build/libvgpmp_cpu_step.so   (g++ -O3 -march=native -fopenmp)
//
// Reference lines followed: kernels Matern-5/2 (GPflow 2.2, [3P]); covariances/multioutput/Kuus.py:42-53, Kufs.py:26-34;
// models/vgpmp.py:200-218 (q_mu / q_sqrt assembly), :281-283 (pathwise draw, joint sigmoid), :287 (sample mean);
// kullback_leiblers/prior_kl.py:16-35; likelihoods/likelihood.py:49-52, 86-176; utils/sampler.py:103-120, 142-168, 190-244;
// utils/sdf_utils.py:62-66, 100-136; GPflowSampling random_fourier / exact_update ([3P]); Keras Adam.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

struct vgo_scene {
    int32_t D, P, craig, nx, ny, nz;
    const double* dh;             // [D, 3]  d, a, alpha          (data/robots/*/config.yaml dh_parameters)
    const double* twist;          // [D]
    const double* base;           // [4, 4]  base pose
    const int32_t* sphere_frame;  // [P]     frame (0..D) of every sphere, non-decreasing
    const double* sphere_off;     // [P, 3]
    const double* radii;          // [P]
    const double* low;            // [D]
    const double* high;           // [D]
    const double* sigma_obs;      // [P]     (used un-squared: likelihood.py:37-41, 99)
    const double* grid;           // [nx, ny, nz] data[x, y, z]
    double origin[3], offset[3];  // grid origin; scene position subtracted from the sphere centres (likelihood.py:160)
    double delta, epsilon;
};

struct vgo_dims { int32_t S, N, M, B; };

struct vgo_state {                // unconstrained variables and Adam moments of ONE problem (float64, updated in place)
    double *q_mu, *q_sqrt, *raw_ell, *raw_var;                      // [M, L], [L, M, M], [L], [L]
    double *m_q_mu, *m_q_sqrt, *m_raw_ell, *m_raw_var;
    double *v_q_mu, *v_q_sqrt, *v_raw_ell, *v_raw_var;
};

struct vgo_noise {                // layouts of oracle.Noise
    const double *omega, *beta, *w, *eps, *eps2;                    // [L, B, D], [L, B], [S, L, B], [S, Mz, L], [S, Mz, L]
};

}  // extern "C"

namespace {

constexpr double kSqrt5 = 2.23606797749978969640917366873128;
constexpr double kVarFloor = 0.1;      // models/vgpmp.py:139 positive(lower=1e-1)

inline double softplus(double x) { return x > 0 ? x + std::log1p(std::exp(-x)) : std::log1p(std::exp(x)); }
inline double sigmoid(double x) { return 1.0 / (1.0 + std::exp(-x)); }

inline double matern52(double t1, double t2, double ell, double var) {
    double r = std::fabs(t1 - t2) / ell;
    r = std::sqrt(std::max(r * r, 1e-36));       // GPflow clips the squared distance before the root
    return var * (1.0 + kSqrt5 * r + 5.0 / 3.0 * r * r) * std::exp(-kSqrt5 * r);
}
inline double matern52_dell(double t1, double t2, double ell, double var) {
    const double r = std::fabs(t1 - t2) / ell;
    return var * std::exp(-kSqrt5 * r) * (5.0 * r * r / (3.0 * ell)) * (1.0 + kSqrt5 * r);
}

// dense helpers on row-major n x n / n x m matrices
void cholesky(const double* a, double* l, int n) {      // lower factor; the strict upper triangle is zeroed
    std::fill(l, l + (size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = a[j * n + j];
        for (int k = 0; k < j; ++k) d -= l[j * n + k] * l[j * n + k];
        d = std::sqrt(d);
        l[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = a[i * n + j];
            for (int k = 0; k < j; ++k) s -= l[i * n + k] * l[j * n + k];
            l[i * n + j] = s / d;
        }
    }
}
void solve_lower(const double* l, double* x, int n, int m) {          // X <- L^-1 X,  X [n, m]
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < i; ++k)
            for (int c = 0; c < m; ++c) x[i * m + c] -= l[i * n + k] * x[k * m + c];
        for (int c = 0; c < m; ++c) x[i * m + c] /= l[i * n + i];
    }
}
void solve_lower_t(const double* l, double* x, int n, int m) {        // X <- L^-T X
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k)
            for (int c = 0; c < m; ++c) x[i * m + c] -= l[k * n + i] * x[k * m + c];
        for (int c = 0; c < m; ++c) x[i * m + c] /= l[i * n + i];
    }
}

struct Mat4 { double m[16]; };
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 c;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j];
            c.m[4 * i + j] = s;
        }
    return c;
}
inline Mat4 dh_link(bool craig, double theta, double d, double a, double alpha) {
    const double ct = std::cos(theta), st = std::sin(theta), ca = std::cos(alpha), sa = std::sin(alpha);
    Mat4 h;
    if (craig) {      // utils/sampler.py:190-214 (modified / Craig)
        const double v[16] = {ct, -st, 0, a, st * ca, ct * ca, -sa, -d * sa, st * sa, ct * sa, ca, d * ca, 0, 0, 0, 1};
        std::memcpy(h.m, v, sizeof(v));
    } else {          // utils/sampler.py:142-168 (classic)
        const double v[16] = {ct, -st * ca, st * sa, a * ct, st, ct * ca, -ct * sa, a * st, 0, sa, ca, d, 0, 0, 0, 1};
        std::memcpy(h.m, v, sizeof(v));
    }
    return h;
}

// log p(e | g) of one configuration and its gradient wrt the joint angles (likelihoods/likelihood.py:86-176)
double log_prob(const vgo_scene& sc, const double* g, double* dlogp_dg) {
    const int D = sc.D, P = sc.P;
    Mat4 T[17];
    std::memcpy(T[0].m, sc.base, sizeof(double) * 16);
    for (int i = 0; i < D; ++i) T[i + 1] = mul(T[i], dh_link(sc.craig != 0, g[i] + sc.twist[i], sc.dh[3 * i], sc.dh[3 * i + 1], sc.dh[3 * i + 2]));
    double F[17][3] = {}, Mo[17][3] = {};      // force / moment sums per frame
    double acc = 0.0;
    for (int p = 0; p < P; ++p) {
        const Mat4& t = T[sc.sphere_frame[p]];
        double pos[3], gp[3];
        long idx[3];
        for (int i = 0; i < 3; ++i)
            pos[i] = t.m[4 * i] * sc.sphere_off[3 * p] + t.m[4 * i + 1] * sc.sphere_off[3 * p + 1] + t.m[4 * i + 2] * sc.sphere_off[3 * p + 2] + t.m[4 * i + 3];
        const int n[3] = {sc.nx, sc.ny, sc.nz};
        for (int i = 0; i < 3; ++i) {           // utils/sdf_utils.py:62-66: clip(int64_trunc(((p - offset) - origin) / delta), 0, n - 1)
            const double q = ((pos[i] - sc.offset[i]) - sc.origin[i]) / sc.delta;
            idx[i] = std::min<long>(std::max<long>((long)std::trunc(q), 0), n[i] - 1);
        }
        auto at = [&](long x, long y, long z) { return sc.grid[((size_t)x * sc.ny + y) * sc.nz + z]; };
        const double dist = at(idx[0], idx[1], idx[2]) - sc.radii[p];
        const double c = std::max(sc.epsilon - dist, 0.0);                               // likelihood.py:131-143
        acc += c * c / sc.sigma_obs[p];                                                  // likelihood.py:99
        // clamped central difference, exact zeros -> 0.1 (utils/sdf_utils.py:100-136); d logp / d dist = +c / sigma
        for (int i = 0; i < 3; ++i) {
            long hi[3] = {idx[0], idx[1], idx[2]}, lo[3] = {idx[0], idx[1], idx[2]};
            hi[i] = std::min<long>(idx[i] + 1, n[i] - 1);
            lo[i] = std::max<long>(idx[i] - 1, 0);
            double gr = (at(hi[0], hi[1], hi[2]) - at(lo[0], lo[1], lo[2])) / (2.0 * sc.delta);
            if (gr == 0.0) gr = 0.1;
            gp[i] = c / sc.sigma_obs[p] * gr;
        }
        const int fr = sc.sphere_frame[p];
        for (int i = 0; i < 3; ++i) F[fr][i] += gp[i];
        Mo[fr][0] += pos[1] * gp[2] - pos[2] * gp[1];
        Mo[fr][1] += pos[2] * gp[0] - pos[0] * gp[2];
        Mo[fr][2] += pos[0] * gp[1] - pos[1] * gp[0];
    }
    // geometric Jacobian: joint i turns about z of frame i (Craig) or i - 1 (classic) and moves the spheres on frames >= i
    double Fs[3] = {0, 0, 0}, Ms[3] = {0, 0, 0};
    for (int i = D; i >= 1; --i) {
        for (int k = 0; k < 3; ++k) { Fs[k] += F[i][k]; Ms[k] += Mo[i][k]; }
        const Mat4& ax = T[sc.craig ? i : i - 1];
        const double z[3] = {ax.m[2], ax.m[6], ax.m[10]}, o[3] = {ax.m[3], ax.m[7], ax.m[11]};
        const double oxF[3] = {o[1] * Fs[2] - o[2] * Fs[1], o[2] * Fs[0] - o[0] * Fs[2], o[0] * Fs[1] - o[1] * Fs[0]};
        dlogp_dg[i - 1] = z[0] * (Ms[0] - oxF[0]) + z[1] * (Ms[1] - oxF[1]) + z[2] * (Ms[2] - oxF[2]);
    }
    return -0.5 * acc;
}

}  // namespace

extern "C" {

// One optimisation step.  t = Adam updates applied before this call; trainable: bit 0 q_mu, 1 q_sqrt, 2 lengthscales, 4 = bit 3
// kernel variance (VGPMP_TRAIN_* of include/vgpmp.h); do_adam = 0: loss and gradient only.  grad (optional): [M L | L M M | L | L].
// Returns the loss (-ELBO) in *loss.  threads <= 0: the OpenMP default.
int vgo_step(const vgo_scene* scp, const vgo_dims* dm, const vgo_state* st, const vgo_noise* nz, const double* X /*[N, D]*/,
             const double* Zy /*[Mz, D]*/, const double* y /*[2, D]*/, double alpha, double lr, int32_t t, int32_t trainable,
             int32_t do_adam, int32_t threads, double* loss, double* grad) {
    const vgo_scene& sc = *scp;
    const int S = dm->S, N = dm->N, M = dm->M, B = dm->B, L = sc.D, D = sc.D, Mz = M + 2, J = N + Mz;
    const double jitter = 1e-6, sj = std::sqrt(jitter);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    std::vector<double> ell(L), var(L), yu(2 * L);
    for (int l = 0; l < L; ++l) {
        ell[l] = softplus(st->raw_ell[l]);
        var[l] = kVarFloor + softplus(st->raw_var[l]);
        for (int e = 0; e < 2; ++e) {          // models/vgpmp.py:75-76: the pinned states through the inverse joint sigmoid
            const double x = (y[e * D + l] - sc.low[l]) / (sc.high[l] - sc.low[l]);
            yu[e * L + l] = std::log(x) - std::log1p(-x);
        }
    }
    const size_t mm = (size_t)Mz * Mz;
    std::vector<double> K(L * mm), Lk(L * mm), Kinv(L * mm), Kuf((size_t)L * Mz * N), A((size_t)L * N * Mz), C(L * mm), mvec((size_t)L * Mz),
        afull((size_t)L * Mz), cvec(2 * L), kl_l(L);
    // ---- covariance path per latent (A2, A3, A11)
#pragma omp parallel for schedule(dynamic)
    for (int l = 0; l < L; ++l) {
        double *Kl = &K[l * mm], *Ll = &Lk[l * mm], *Ki = &Kinv[l * mm], *Cl = &C[l * mm];
        std::vector<double> Kj(mm);
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) {
                Kl[i * Mz + j] = matern52(Zy[i * D + l], Zy[j * D + l], ell[l], var[l]);
                Kj[i * Mz + j] = Kl[i * Mz + j] + (i == j ? jitter : 0.0);
            }
        cholesky(Kj.data(), Ll, Mz);
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) Ki[i * Mz + j] = i == j ? 1.0 : 0.0;
        solve_lower(Ll, Ki, Mz, Mz);
        solve_lower_t(Ll, Ki, Mz, Mz);                                   // (Kuu + jI)^-1
        double* Kf = &Kuf[(size_t)l * Mz * N];
        for (int i = 0; i < Mz; ++i)
            for (int n = 0; n < N; ++n) Kf[i * N + n] = matern52(Zy[i * D + l], X[n * D + l], ell[l], var[l]);
        std::vector<double> sol(Kf, Kf + (size_t)Mz * N);
        solve_lower(Ll, sol.data(), Mz, N);
        solve_lower_t(Ll, sol.data(), Mz, N);
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < Mz; ++i) A[((size_t)l * N + n) * Mz + i] = sol[i * N + n];      // A = Kfu (Kuu + jI)^-1
        // q_sqrt = Lk pad(Q) + jitter diag(1, 1, 0, ...)   (models/vgpmp.py:208-218)
        const double* Q = st->q_sqrt + (size_t)l * M * M;
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) {
                double s = 0.0;
                if (j >= 2)
                    for (int k = std::max(j, 2); k <= i; ++k) s += Ll[i * Mz + k] * Q[(k - 2) * M + (j - 2)];      // Q lower: k - 2 >= j - 2
                Cl[i * Mz + j] = s + ((i == j && i < 2) ? jitter : 0.0);
            }
        double* ml = &mvec[(size_t)l * Mz];
        ml[0] = yu[l]; ml[1] = yu[L + l];
        for (int k = 0; k < M; ++k) ml[2 + k] = st->q_mu[k * L + l];
        // KL: a = (Lk^-1 (m - p_mu))[2:], p_mu = Kj[:, :2] Kj[:2, :2]^-1 y_u     (kullback_leiblers/prior_kl.py:16-35)
        const double k00 = Kj[0], k01 = Kj[1], k10 = Kj[Mz], k11 = Kj[Mz + 1], det = k00 * k11 - k01 * k10;
        const double c0 = (k11 * ml[0] - k01 * ml[1]) / det, c1 = (-k10 * ml[0] + k00 * ml[1]) / det;
        cvec[2 * l] = c0; cvec[2 * l + 1] = c1;
        double* af = &afull[(size_t)l * Mz];
        for (int i = 0; i < Mz; ++i) af[i] = ml[i] - (Kj[i * Mz] * c0 + Kj[i * Mz + 1] * c1);
        solve_lower(Ll, af, Mz, 1);
        double aa = 0.0, ld = 0.0, qq = 0.0;
        for (int i = 2; i < Mz; ++i) aa += af[i] * af[i];
        for (int i = 0; i < M; ++i) {
            ld += std::log(Q[i * M + i] * Q[i * M + i]);
            for (int j = 0; j <= i; ++j) qq += Q[i * M + j] * Q[i * M + j];
        }
        kl_l[l] = 0.5 * (aa - M - ld + qq);
    }
    // ---- random-feature prior (A5): Phi, dPhi/dell [L, J, B];  F0 = W Phi^T, H = W dPhi^T  [S, L, J]
    std::vector<double> Phi((size_t)L * J * B), dPhi((size_t)L * J * B), F0((size_t)S * L * J), H((size_t)S * L * J);
#pragma omp parallel for collapse(2) schedule(static)
    for (int l = 0; l < L; ++l)
        for (int j = 0; j < J; ++j) {
            const double* pt = j < N ? X + (size_t)j * D : Zy + (size_t)(j - N) * D;
            const double c = std::sqrt(2.0 * var[l] / B);
            double* ph = &Phi[((size_t)l * J + j) * B];
            double* dp = &dPhi[((size_t)l * J + j) * B];
            for (int b = 0; b < B; ++b) {
                const double* om = nz->omega + ((size_t)l * B + b) * D;
                double proj = 0.0;
                for (int d = 0; d < D; ++d) proj += pt[d] * om[d];
                const double arg = proj / ell[l] + nz->beta[(size_t)l * B + b];
                ph[b] = c * std::cos(arg);
                dp[b] = c * std::sin(arg) * proj / (ell[l] * ell[l]);
            }
        }
    // (a feature row stays in L1 while every sample's weights stream past it; the dot products as SIMD reductions)
#pragma omp parallel for collapse(2) schedule(static)
    for (int l = 0; l < L; ++l)
        for (int j = 0; j < J; ++j) {
            const double* ph = &Phi[((size_t)l * J + j) * B];
            const double* dp = &dPhi[((size_t)l * J + j) * B];
            for (int s = 0; s < S; ++s) {
                const double* w = nz->w + ((size_t)s * L + l) * B;
                double a = 0.0, h = 0.0;
#pragma omp simd reduction(+ : a, h)
                for (int b = 0; b < B; ++b) { a += w[b] * ph[b]; h += w[b] * dp[b]; }
                F0[((size_t)s * L + l) * J + j] = a;
                H[((size_t)s * L + l) * J + j] = h;
            }
        }
    // ---- Matheron update (A6, A7), likelihood (A8-A10)
    std::vector<double> R((size_t)S * L * Mz), G((size_t)S * L * N), logp((size_t)S * N);
#pragma omp parallel for schedule(static)
    for (int s = 0; s < S; ++s) {
        std::vector<double> f((size_t)L * N);
        for (int l = 0; l < L; ++l) {
            const double *Cl = &C[l * mm], *ml = &mvec[(size_t)l * Mz], *Al = &A[(size_t)l * N * Mz];
            double* r = &R[((size_t)s * L + l) * Mz];
            const double* f0 = &F0[((size_t)s * L + l) * J];
            for (int i = 0; i < Mz; ++i) {
                double u = ml[i];
                for (int k = 0; k < Mz; ++k) u += Cl[i * Mz + k] * nz->eps[((size_t)s * Mz + k) * L + l];
                r[i] = u - f0[N + i] - sj * nz->eps2[((size_t)s * Mz + i) * L + l];
            }
            for (int n = 0; n < N; ++n) {
                double v = f0[n];
                for (int i = 0; i < Mz; ++i) v += Al[n * Mz + i] * r[i];
                f[(size_t)l * N + n] = v;
            }
        }
        for (int n = 0; n < N; ++n) {
            double g[16], dl[16], dgdf[16];
            for (int l = 0; l < L; ++l) {
                const double sg = sigmoid(f[(size_t)l * N + n]), span = sc.high[l] - sc.low[l];
                g[l] = sc.low[l] + span * sg;                                                // likelihood.py:49-52
                dgdf[l] = span * sg * (1.0 - sg);
            }
            logp[(size_t)s * N + n] = log_prob(sc, g, dl);
            for (int l = 0; l < L; ++l) G[((size_t)s * L + l) * N + n] = -(alpha / S) * dl[l] * dgdf[l];     // d loss / d f
        }
    }
    double lik = 0.0, kl = 0.0;
    for (size_t i = 0; i < logp.size(); ++i) lik += logp[i];
    lik *= alpha / S;                                                                        // models/vgpmp.py:287
    for (int l = 0; l < L; ++l) kl += kl_l[l];
    *loss = -(lik - kl);
    // ---- reverse pass per latent
    const size_t n_qmu = (size_t)M * L, n_qs = (size_t)L * M * M;
    std::vector<double> gbuf(n_qmu + n_qs + 2 * L, 0.0);
    double *g_qmu = gbuf.data(), *g_qs = g_qmu + n_qmu, *g_ell = g_qs + n_qs, *g_var = g_ell + L;
#pragma omp parallel for schedule(dynamic)
    for (int l = 0; l < L; ++l) {
        const double *Kl = &K[l * mm], *Ll = &Lk[l * mm], *Ki = &Kinv[l * mm], *Al = &A[(size_t)l * N * Mz], *Kf = &Kuf[(size_t)l * Mz * N];
        const double* Q = st->q_sqrt + (size_t)l * M * M;
        std::vector<double> dR((size_t)S * Mz), dC(mm, 0.0), dA((size_t)N * Mz, 0.0), dmv(Mz, 0.0);
        double gv = 0.0, ge = 0.0;
        for (int s = 0; s < S; ++s) {
            const double* Gl = &G[((size_t)s * L + l) * N];
            const double* r = &R[((size_t)s * L + l) * Mz];
            const double *f0 = &F0[((size_t)s * L + l) * J], *h = &H[((size_t)s * L + l) * J];
            double* dr = &dR[(size_t)s * Mz];
            for (int i = 0; i < Mz; ++i) {
                double v = 0.0;
#pragma omp simd reduction(+ : v)
                for (int n = 0; n < N; ++n) v += Gl[n] * Al[n * Mz + i];
                dr[i] = v;
                dmv[i] += v;
            }
            for (int i = 0; i < Mz; ++i)
                for (int k = 0; k < Mz; ++k) dC[i * Mz + k] += dr[i] * nz->eps[((size_t)s * Mz + k) * L + l];
            for (int n = 0; n < N; ++n)
                for (int i = 0; i < Mz; ++i) dA[n * Mz + i] += Gl[n] * r[i];
            // the prior draws are linear in sqrt(var); H = d F0 / d ell
            for (int n = 0; n < N; ++n) { gv += Gl[n] * f0[n]; ge += Gl[n] * h[n]; }
            for (int i = 0; i < Mz; ++i) { gv -= dr[i] * f0[N + i]; ge -= dr[i] * h[N + i]; }
        }
        gv /= 2.0 * var[l];
        // A = Kfu Kj^-1:  dKfu = dA Kj^-1,  dKj = -(A^T dA) Kj^-1
        std::vector<double> dKfu((size_t)N * Mz), AtdA(mm, 0.0), dKj(mm), dLk(mm, 0.0);
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < Mz; ++i) {
                double v = 0.0;
                for (int k = 0; k < Mz; ++k) v += dA[n * Mz + k] * Ki[k * Mz + i];
                dKfu[n * Mz + i] = v;
            }
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < Mz; ++i)
                for (int k = 0; k < Mz; ++k) AtdA[i * Mz + k] += Al[n * Mz + i] * dA[n * Mz + k];
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) {
                double v = 0.0;
                for (int k = 0; k < Mz; ++k) v += AtdA[i * Mz + k] * Ki[k * Mz + j];
                dKj[i * Mz + j] = -v;
            }
        // C = Lk pad(Q) + ...:  dLk = dC pad(Q)^T,  d Q = tril(Lk^T dC)[2:, 2:]
        for (int i = 0; i < Mz; ++i)
            for (int k = 2; k < Mz; ++k) {
                double v = 0.0;
                for (int j = 2; j <= k; ++j) v += dC[i * Mz + j] * Q[(k - 2) * M + (j - 2)];
                dLk[i * Mz + k] = v;
            }
        double* gq = g_qs + (size_t)l * M * M;
        for (int r = 0; r < M; ++r)
            for (int c = 0; c <= r; ++c) {
                double v = 0.0;
                for (int i = 0; i < Mz; ++i) v += Ll[i * Mz + (r + 2)] * dC[i * Mz + (c + 2)];
                gq[r * M + c] = v + Q[r * M + c] - (r == c ? 1.0 / Q[r * M + r] : 0.0);      // + KL: Q - diag(1 / diag Q)
            }
        // KL through a = Lk^-1 (m - p_mu)
        const double* af = &afull[(size_t)l * Mz];
        std::vector<double> dd(Mz);
        for (int i = 0; i < Mz; ++i) dd[i] = i < 2 ? 0.0 : af[i];
        solve_lower_t(Ll, dd.data(), Mz, 1);                                                   // d / d (m - p_mu)
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) dLk[i * Mz + j] -= dd[i] * af[j];
        const double c0 = cvec[2 * l], c1 = cvec[2 * l + 1];
        for (int i = 0; i < Mz; ++i) { dKj[i * Mz] += -dd[i] * c0; dKj[i * Mz + 1] += -dd[i] * c1; }
        {
            double dc[2] = {0.0, 0.0};
            for (int i = 0; i < Mz; ++i) {
                const double kj0 = Kl[i * Mz] + (i == 0 ? jitter : 0.0), kj1 = Kl[i * Mz + 1] + (i == 1 ? jitter : 0.0);
                dc[0] += kj0 * -dd[i]; dc[1] += kj1 * -dd[i];
            }
            const double k00 = Kl[0] + jitter, k01 = Kl[1], k10 = Kl[Mz], k11 = Kl[Mz + 1] + jitter, det = k00 * k11 - k01 * k10;
            // solve Kyy^T x = dc
            const double x0 = (k11 * dc[0] - k10 * dc[1]) / det, x1 = (-k01 * dc[0] + k00 * dc[1]) / det;
            dKj[0] -= x0 * c0; dKj[1] -= x0 * c1; dKj[Mz] -= x1 * c0; dKj[Mz + 1] -= x1 * c1;
        }
        // Cholesky adjoint (Murray 2016): P = Phi(Lk^T tril(dLk)), S = Lk^-T P Lk^-1, dK = (S + S^T) / 2
        {
            std::vector<double> Pm(mm, 0.0);
            for (int i = 0; i < Mz; ++i)
                for (int j = 0; j <= i; ++j) {
                    double v = 0.0;
                    for (int k = i; k < Mz; ++k) v += Ll[k * Mz + i] * dLk[k * Mz + j];      // (tril(dLk))[k, j]: k >= j holds as k >= i >= j
                    Pm[i * Mz + j] = i == j ? 0.5 * v : v;
                }
            solve_lower_t(Ll, Pm.data(), Mz, Mz);                                              // Lk^-T P
            // (X Lk^-1): solve on the transpose
            std::vector<double> Xt(mm);
            for (int i = 0; i < Mz; ++i)
                for (int j = 0; j < Mz; ++j) Xt[j * Mz + i] = Pm[i * Mz + j];
            solve_lower_t(Ll, Xt.data(), Mz, Mz);                                              // (Lk^-T) X^T = (X Lk^-1)^T
            for (int i = 0; i < Mz; ++i)
                for (int j = 0; j < Mz; ++j) dKj[i * Mz + j] += 0.5 * (Xt[j * Mz + i] + Xt[i * Mz + j]);
        }
        for (int k = 0; k < M; ++k) g_qmu[k * L + l] = dmv[2 + k] + dd[2 + k];
        // hyper-parameters: d K / d var = K / var;  d K / d ell = matern52_dell
        double gvk = 0.0, gek = 0.0;
        for (int i = 0; i < Mz; ++i)
            for (int j = 0; j < Mz; ++j) {
                gvk += dKj[i * Mz + j] * Kl[i * Mz + j];
                gek += dKj[i * Mz + j] * matern52_dell(Zy[i * D + l], Zy[j * D + l], ell[l], var[l]);
            }
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < Mz; ++i) {
                gvk += dKfu[n * Mz + i] * Kf[i * N + n];
                gek += dKfu[n * Mz + i] * matern52_dell(X[n * D + l], Zy[i * D + l], ell[l], var[l]);
            }
        g_ell[l] = (ge + gek) * sigmoid(st->raw_ell[l]);
        g_var[l] = (gv + gvk / var[l]) * sigmoid(st->raw_var[l]);
    }
    if (grad) std::memcpy(grad, gbuf.data(), gbuf.size() * sizeof(double));
    if (!do_adam) return 0;
    // ---- Keras Adam(lr, 0.8, 0.95, epsilon 1e-7) on the unconstrained variables (models/vgpmp.py:77)
    const double b1 = 0.8, b2 = 0.95, eps = 1e-7, tt = (double)(t + 1);
    const double lr_t = lr * std::sqrt(1.0 - std::pow(b2, tt)) / (1.0 - std::pow(b1, tt));
    auto adam = [&](double* x, double* m, double* v, const double* g, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            m[i] += (g[i] - m[i]) * (1.0 - b1);
            v[i] += (g[i] * g[i] - v[i]) * (1.0 - b2);
            x[i] -= lr_t * m[i] / (std::sqrt(v[i]) + eps);
        }
    };
    if (trainable & 1) adam(st->q_mu, st->m_q_mu, st->v_q_mu, g_qmu, n_qmu);
    if (trainable & 2) adam(st->q_sqrt, st->m_q_sqrt, st->v_q_sqrt, g_qs, n_qs);
    if (trainable & 4) adam(st->raw_ell, st->m_raw_ell, st->v_raw_ell, g_ell, L);
    if (trainable & 8) adam(st->raw_var, st->m_raw_var, st->v_raw_var, g_var, L);
    return 0;
}

// The randomness of one step (the layouts of vgo_noise) for the TIMED baseline: the reference redraws it every step
// (models/vgpmp.py:281), so a CPU figure that leaves the draw out would flatter the CPU.  std::mt19937_64 per chunk of 4096 values
// (seeded by (seed, array, chunk): the same numbers whatever the thread count); omega = normal / sqrt(chi^2_5 / 5) (Student-t
// spectral draw of the Matern-5/2 features), beta ~ U(0, 2 pi).  Not the device's Philox streams -- parity tests inject those.
int vgo_draw_noise(uint64_t seed, int32_t S, int32_t L, int32_t D, int32_t B, int32_t Mz, double* omega, double* beta, double* w,
                   double* eps, double* eps2, int32_t threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    struct Job { double* dst; size_t n; int kind; };      // kind 0 normal, 1 uniform phase, 2 Student-t rows of D
    const Job jobs[5] = {{omega, (size_t)L * B, 2}, {beta, (size_t)L * B, 1}, {w, (size_t)S * L * B, 0}, {eps, (size_t)S * Mz * L, 0},
                         {eps2, (size_t)S * Mz * L, 0}};
    constexpr size_t kChunk = 4096;
    for (int a = 0; a < 5; ++a) {
        const Job jb = jobs[a];
        const long nchunks = (long)((jb.n + kChunk - 1) / kChunk);
#pragma omp parallel for schedule(static)
        for (long c = 0; c < nchunks; ++c) {
            std::seed_seq sq{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)a, (uint32_t)c};
            std::mt19937_64 gen(sq);
            std::normal_distribution<double> nrm(0.0, 1.0);
            std::uniform_real_distribution<double> uni(0.0, 6.283185307179586476925286766559);
            const size_t lo = (size_t)c * kChunk, hi = std::min(jb.n, lo + kChunk);
            for (size_t i = lo; i < hi; ++i) {
                if (jb.kind == 0) jb.dst[i] = nrm(gen);
                else if (jb.kind == 1) jb.dst[i] = uni(gen);
                else {
                    double chi = 0.0;
                    for (int k = 0; k < 5; ++k) { const double z = nrm(gen); chi += z * z; }
                    const double sc = 1.0 / std::sqrt(chi / 5.0);
                    for (int d = 0; d < D; ++d) jb.dst[i * D + d] = nrm(gen) * sc;
                }
            }
        }
    }
    return 0;
}

int vgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
