// Scalar float64 helpers shared by the GP kernels: bijectors, the Matern-5/2 kernel, Keras Adam.
#pragma once
#include "vgpmp_device.h"

constexpr double kVarFloor = 0.1;               // models/vgpmp.py:139 positive(lower=1e-1)
constexpr double kSqrt5 = 2.2360679774997896964;
constexpr double kZLow = 0.09, kZHigh = 0.91;   // models/vgpmp.py:41 bounded_Z: tfb.Sigmoid(0.09, 0.91)

// (one evaluation for both signs: max(x, 0) + log1p(exp(-|x|)) is x + log1p(exp(-x)) for x > 0 and log1p(exp(x)) otherwise,
// bit for bit, without the two code paths)
__device__ __forceinline__ double softplus_d(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
// softplus and logistic of the same argument from ONE exponential (the logistic differs from sigmoid_d in the last bit
// for x < 0: e / (1 + e) with e = exp(x) instead of 1 / (1 + exp(-x)))
// log(1 + e) for 0 < e <= 1 from the float32 logarithm and ONE float64 exponential: y0 = logf(1 + e) is good to ~1e-7, and
// with t = (1 + e) exp(-y0) = 1 + d the correction log(t) = d - d^2 / 2 + O(d^3) leaves ~1e-21.  (The library log1p is a
// long routine of its own; here the only long routine on the chain is exp, which the caller has just run.)
__device__ __forceinline__ double vg_log1p_unit(double e) {
    const double y0 = (double)__logf((float)(1.0 + e));
    const double d = (1.0 + e) * exp(-y0) - 1.0;
    return y0 + (d - 0.5 * d * d);
}
__device__ __forceinline__ void softplus_sigmoid_d(double x, double* sp, double* sg) {
    const double e = exp(-fabs(x)), r = 1.0 / (1.0 + e);
    *sp = fmax(x, 0.0) + vg_log1p_unit(e);
    *sg = x >= 0.0 ? r : e * r;
}
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ double matern52(double t1, double t2, double ell, double var) {
    double r = fabs(t1 - t2) / ell;
    r = sqrt(fmax(r * r, 1e-36));
    return var * (1.0 + kSqrt5 * r + (5.0 / 3.0) * r * r) * exp(-kSqrt5 * r);
}
// d k(t1, t2) / d t1
__device__ __forceinline__ double matern52_d1(double t1, double t2, double ell, double var) {
    const double d = t1 - t2, r = fabs(d) / ell;
    return -var * (5.0 / 3.0) * d / (ell * ell) * (1.0 + kSqrt5 * r) * exp(-kSqrt5 * r);
}

__device__ __forceinline__ void adam_update(double* x, double* m, double* v, double g, double lr_t) {
#pragma clang fp contract(off)      // the same roundings wherever this is inlined (contraction depends on the surrounding code)
    // Keras Adam (TF 2.12): beta1 = 0.8, beta2 = 0.95 (models/vgpmp.py:77), epsilon 1e-7
    double mm = *m + (g - *m) * (1.0 - 0.8);
    double vv = *v + (g * g - *v) * (1.0 - 0.95);
    *m = mm; *v = vv;
    *x -= lr_t * mm / (sqrt(vv) + 1e-7);
}

// Bias-corrected Adam step size of the update with 1-based count t (Keras: lr sqrt(1 - b2^t) / (1 - b1^t))
__device__ __forceinline__ double adam_step_size(double lr, double t) {
    return lr * sqrt(1.0 - exp(t * -0.05129329438755058)) / (1.0 - exp(t * -0.2231435513142098));
}
