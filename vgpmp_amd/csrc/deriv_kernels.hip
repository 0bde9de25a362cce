// Velocity-constrained kernel variant: Kuu / Kuf of FirstOrderKernelDerivativeSeparateIndependent
// (gpflow_vgpmp/covariances/multioutput/Kuus.py:17-39, Kufs.py:14-23) with the derivative kernels of
// gpflow_vgpmp/derivatives/first_order.py:14-29 and second_order.py:27-58, float64, one lane per matrix entry.
// The variant is unreachable from VGPMP.initialize in the reference; it is provided so that the plugin surface is
// whole.  A few thousand exponentials per call: nothing to tune.
#include "vgpmp_device.h"

namespace {

constexpr int kBlock = 256;
constexpr double kSqrt5 = 2.23606797749978969641, kFiveThirds = 5.0 / 3.0, kDefaultJitter = 1e-6;

struct Kern { int kind; double ell, var; };   // kind 0 Matern-5/2, 1 squared exponential

__device__ __forceinline__ double k_val(const Kern& k, double x, double y) {
    const double d = (x - y) / k.ell;
    if (k.kind == 1) return k.var * exp(-0.5 * d * d);
    const double r = fabs(d);
    return k.var * (1.0 + kSqrt5 * r + kFiveThirds * r * r) * exp(-kSqrt5 * r);
}
// d k(x, y) / d y   (first_order.py:14-29)
__device__ __forceinline__ double k_grad(const Kern& k, double x, double y) {
    const double diff = x - y;
    if (k.kind == 1) return diff / (k.ell * k.ell) * k_val(k, x, y);
    const double s5r = kSqrt5 * fabs(diff) / k.ell;
    return kFiveThirds * (1.0 + s5r) * exp(-s5r) * diff / (k.ell * k.ell) * k.var;
}
// d^2 k(x, y) / dx dy   (second_order.py:27-58); exact zeros are replaced by 5/3 / ell^2 as the reference does (:45)
__device__ __forceinline__ double k_grad_grad(const Kern& k, double x, double y) {
    const double diff = x - y;
    if (k.kind == 1) return (k.ell * k.ell - diff * diff) / (k.ell * k.ell * k.ell * k.ell) * k_val(k, x, y);
    const double r = fabs(diff) / k.ell, s5r = kSqrt5 * r;
    const double dr_dx = r != 0.0 ? diff / (r * k.ell * k.ell) : 0.0;
    const double res = k.var * kFiveThirds * (5.0 * r * r - s5r - 1.0) * exp(-s5r) * dr_dx * (-dr_dx);
    return res == 0.0 ? kFiveThirds / (k.ell * k.ell) : res;
}

// entry (r, c) of [Kuu | Kuf] of latent l: rows / columns 0, 1 = derivative observations at the two conditioned
// times ny = Zy[:2], the others = the points of Zy (then, for Kuf, the times X)
__global__ __launch_bounds__(kBlock) void velocity_kuu_kuf_kernel(int kind, const double* __restrict__ Zy,
                                                                   const double* __restrict__ X, int Mz, int N, int L, int D,
                                                                   const double* __restrict__ ell,
                                                                   const double* __restrict__ var, double jitter,
                                                                   double* __restrict__ Kuu, double* __restrict__ Kuf) {
    const int Me = Mz + 2, W = Me + N;
    const int e = blockIdx.x * kBlock + threadIdx.x, l = blockIdx.y;
    if (e >= Me * W) return;
    const int r = e / W, c = e - r * W;
    const Kern k{kind, ell[l], var[l]};
    const bool rd = r < 2;                                  // derivative row
    const double xr = Zy[(size_t)(rd ? r : r - 2) * D + l];
    if (c < Me) {
        const bool cd = c < 2;
        const double xc = Zy[(size_t)(cd ? c : c - 2) * D + l];
        double v = rd ? (cd ? k_grad_grad(k, xr, xc) + (r == c ? kDefaultJitter : 0.0) : k_grad(k, xr, xc))
                      : (cd ? k_grad(k, xr, xc) : k_val(k, xr, xc));
        if (r == c) v += jitter;
        Kuu[((size_t)l * Me + r) * Me + c] = v;
    } else {
        const int n = c - Me;
        const double xn = X[(size_t)n * D + l];
        Kuf[((size_t)l * Me + r) * N + n] = rd ? k_grad(k, xr, xn) : k_val(k, xr, xn);
    }
}

// order 0 / 1 / 2: k, dk/dy, d2k/dxdy of every pair (x_i, y_j)   (the K_grad / K_grad_grad dispatchers on plain arrays)
__global__ __launch_bounds__(kBlock) void kernel_derivative_kernel(int kind, int order, const double* __restrict__ x, int n,
                                                                    const double* __restrict__ y, int m, double ell, double var,
                                                                    double* __restrict__ out) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n * m) return;
    const int i = e / m, j = e - i * m;
    const Kern k{kind, ell, var};
    out[e] = order == 0 ? k_val(k, x[i], y[j]) : order == 1 ? k_grad(k, x[i], y[j]) : k_grad_grad(k, x[i], y[j]);
}

// K_conditioned (kernel_conditioning/multioutput/cond_kernel.py:17-25): K[l, i, j] = k_l(Z[i, l], X[j, l]) (+ jitter on i == j:
// Kuu of covariances/multioutput/Kuus.py:42-53), every latent in one launch
__global__ __launch_bounds__(kBlock) void cov_matrices_kernel(int kind, const double* __restrict__ Z, int nz,
                                                               const double* __restrict__ X, int nx, int D,
                                                               const double* __restrict__ ell, const double* __restrict__ var,
                                                               double jitter, double* __restrict__ out) {
    const int e = blockIdx.x * kBlock + threadIdx.x, l = blockIdx.y;
    if (e >= nz * nx) return;
    const int i = e / nx, j = e - i * nx;
    const Kern k{kind, ell[l], var[l]};
    out[(size_t)l * nz * nx + e] = k_val(k, Z[(size_t)i * D + l], X[(size_t)j * D + l]) + (i == j ? jitter : 0.0);
}

}  // namespace

int vg_launch_cov_matrices(int kind, const double* Z, int nz, const double* X, int nx, int L, const double* ell,
                           const double* var, double jitter, double* out, hipStream_t st) {
    if (nz * nx == 0) return 0;
    hipLaunchKernelGGL(cov_matrices_kernel, dim3((nz * nx + kBlock - 1) / kBlock, L), dim3(kBlock), 0, st, kind, Z, nz, X, nx, L,
                       ell, var, jitter, out);
    return (int)hipGetLastError();
}

int vg_launch_kernel_derivative(int kind, int order, const double* x, int n, const double* y, int m, double ell, double var,
                                double* out, hipStream_t st) {
    if (n * m == 0) return 0;
    hipLaunchKernelGGL(kernel_derivative_kernel, dim3((n * m + kBlock - 1) / kBlock), dim3(kBlock), 0, st, kind, order, x, n, y,
                       m, ell, var, out);
    return (int)hipGetLastError();
}

int vg_launch_velocity_kuu_kuf(int kind, const double* Zy, const double* X, int Mz, int N, int L, int D, const double* ell,
                               const double* var, double jitter, double* Kuu, double* Kuf, hipStream_t st) {
    const int Me = Mz + 2, total = Me * (Me + N);
    hipLaunchKernelGGL(velocity_kuu_kuf_kernel, dim3((total + kBlock - 1) / kBlock, L), dim3(kBlock), 0, st, kind, Zy, X, Mz, N,
                       L, D, ell, var, jitter, Kuu, Kuf);
    return (int)hipGetLastError();
}
