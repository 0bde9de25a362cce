// Measurement aid: accuracy of v_rcp_f64 and of one / two Newton steps against IEEE division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p = x[i];
    double r0 = __builtin_amdgcn_rcp(p);
    double r1 = fma(fma(-p, r0, 1.0), r0, r0);
    double r2 = fma(fma(-p, r1, 1.0), r1, r1);
    const double t = 1.0 / p;
    out[3 * i] = fabs(r0 - t) / fabs(t); out[3 * i + 1] = fabs(r1 - t) / fabs(t); out[3 * i + 2] = fabs(r2 - t) / fabs(t);
}
int main() {
    const int n = 1 << 20;
    double *hx = new double[n], *ho = new double[3 * n], *dx, *dout;
    for (int i = 0; i < n; ++i) hx[i] = ldexp(0.5 + 0.5 * drand48(), (int)(60 * drand48()) - 40) * (drand48() < 0.5 ? -1 : 1);
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) if (ho[3 * i + j] > m[j]) m[j] = ho[3 * i + j];
    printf("max relative error vs 1/x: rcp %.3e (2^%.1f)  +1 Newton %.3e (2^%.1f)  +2 Newton %.3e (2^%.1f)\n", m[0], log2(m[0]), m[1],
           m[1] > 0 ? log2(m[1]) : -99.0, m[2], m[2] > 0 ? log2(m[2]) : -99.0);
    return 0;
}
