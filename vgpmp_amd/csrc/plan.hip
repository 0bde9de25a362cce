// Plan extraction on the device (models/vgpmp.py:312-339): posterior mean, the 150 pathwise samples as joint angles,
// get_best_sample's arg-max of the summed log-likelihood, the best path, and -- compute_uncertainty=True -- the variance
// over the samples of the end-effector position (models/vgpmp.py:322-327).  Consumes what a forward-only
// vgpmp_elbo_step at Xnew left behind: A = Kfu (Kuu + jI)^-1 and q_mu in the workspace, f and logp in the outputs.
#include "vgpmp_device.h"
#include "fk_chain.h"
#include "gp_path.h"

#pragma clang fp contract(off)

namespace {

constexpr int kPlanBlock = 256;

__device__ __forceinline__ float joint_angle(const vgpmp_robot* __restrict__ rb, int j, float x) {
    const float sg = 1.0f / (1.0f + __expf(-x));                   // likelihoods/likelihood.py:49-52, as the likelihood kernels
    return fmaf(rb->joint_tab[j][7], sg, rb->joint_tab[j][5]);
}

// grid (P): scores, first arg-max, best path
__global__ __launch_bounds__(kPlanBlock) void plan_pick_kernel(const vgpmp_robot* __restrict__ rb, const float* __restrict__ f,
                                                                const float* __restrict__ logp, int S, int L, int N,
                                                                int32_t* __restrict__ best, float* __restrict__ best_path) {
    __shared__ double sv[kPlanBlock];
    __shared__ int si[kPlanBlock];
    const int p = blockIdx.x, tid = threadIdx.x;
    double bv = -__builtin_huge_val();
    int bi = 0x7fffffff;
    for (int s = tid; s < S; s += kPlanBlock) {                    // ascending s per thread: strict > keeps the first maximum
        const float* row = logp + ((size_t)p * S + s) * N;
        double t = 0.0;
        for (int n = 0; n < N; ++n) t += (double)row[n];          // models/vgpmp.py:337: sum over time
        if (t != t) t = -__builtin_huge_val();                     // a diverged sample (NaN score) never wins
        if (t > bv) { bv = t; bi = s; }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int o = kPlanBlock / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const double a = sv[tid], b = sv[tid + o];
            const int ia = si[tid], ib = si[tid + o];
            if (b > a || (b == a && ib < ia)) { sv[tid] = b; si[tid] = ib; }     // tf.math.argmax: first maximum
        }
        __syncthreads();
    }
    const int b = si[0] < S ? si[0] : 0;                            // every score NaN / -inf (or S == 0): a valid index, as tf.argmax
    if (tid == 0) best[p] = b;
    if (S <= 0) return;
    const float* fb = f + ((size_t)p * S + b) * L * N;
    for (int e = tid; e < N * L; e += kPlanBlock) {
        const int n = e / L, l = e - n * L;
        best_path[(size_t)p * N * L + e] = joint_angle(rb, l, fb[(size_t)l * N + n]);
    }
}

// one thread per (p, n, l): joint_sigmoid(sum_m A[p,l,n,m] m[p,l,m])   (gpflow conditional, whiten=False: vgpmp.py:316-317)
__global__ __launch_bounds__(kPlanBlock) void plan_mean_kernel(const vgpmp_robot* __restrict__ rb, const float4* __restrict__ A4,
                                                                const float* __restrict__ m, int P, int L, int N, int Mz,
                                                                float* __restrict__ mean) {
    const size_t i = (size_t)blockIdx.x * kPlanBlock + threadIdx.x;
    if (i >= (size_t)P * N * L) return;
    const int l = (int)(i % L), n = (int)((i / L) % N), p = (int)(i / ((size_t)L * N));
    const float4* a = A4 + (((size_t)p * L + l) * N + n) * Mz;
    const float* mm = m + ((size_t)p * L + l) * Mz;
    float t = 0.f;
    for (int k = 0; k < Mz; ++k) t = fmaf(a[k].x, mm[k], t);
    mean[i] = joint_angle(rb, l, t);
}

// one thread per element of samples [P, S, N, L]
__global__ __launch_bounds__(kPlanBlock) void plan_samples_kernel(const vgpmp_robot* __restrict__ rb, const float* __restrict__ f,
                                                                   size_t PS, int L, int N, float* __restrict__ samples) {
    const size_t i = (size_t)blockIdx.x * kPlanBlock + threadIdx.x;
    if (i >= PS * N * L) return;
    const int l = (int)(i % L), n = (int)((i / L) % N);
    const size_t ps = i / ((size_t)L * N);
    samples[i] = joint_angle(rb, l, f[(ps * L + l) * N + n]);
}

// one thread per (p, n): population variance over the samples of the last frame's origin (tfp.stats.variance, axis 0)
__global__ __launch_bounds__(64) void plan_ee_variance_kernel(const vgpmp_robot* __restrict__ rb, const float* __restrict__ f,
                                                              int P, int S, int L, int N, float* __restrict__ var) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= P * N) return;
    const int p = i / N, n = i - p * N;
    auto ee = [&](int s) {
        Frame T = base_frame(rb);
        for (int j = 0; j < L; ++j) {
            float st, ct;
            sincosf(joint_angle(rb, j, f[(((size_t)p * S + s) * L + j) * N + n]) + rb->joint_tab[j][4], &st, &ct);
            dh_apply(rb, j, st, ct, T);
        }
        return T.t;
    };
    double mx = 0.0, my = 0.0, mz = 0.0;
    for (int s = 0; s < S; ++s) { const vg_float3 t = ee(s); mx += t.x; my += t.y; mz += t.z; }
    mx /= S; my /= S; mz /= S;
    double vx = 0.0, vy = 0.0, vz = 0.0;
    for (int s = 0; s < S; ++s) {
        const vg_float3 t = ee(s);
        vx += (t.x - mx) * (t.x - mx); vy += (t.y - my) * (t.y - my); vz += (t.z - mz) * (t.z - mz);
    }
    var[3 * (size_t)i] = (float)(vx / S); var[3 * (size_t)i + 1] = (float)(vy / S); var[3 * (size_t)i + 2] = (float)(vz / S);
}

}  // namespace

int vg_launch_sample_paths(const vgpmp_dims* d, const vgpmp_robot* rb, const vg_workspace* ws, const float* f, const float* logp,
                           float* mean, int32_t* best, float* best_path, float* samples, float* ee_var, hipStream_t st) {
    const int P = d->num_problems, S = d->S, L = d->L, N = d->N, Mz = vg_mz(d);
    if (mean) {
        const size_t n = (size_t)P * N * L;
        hipLaunchKernelGGL(plan_mean_kernel, dim3((unsigned)((n + kPlanBlock - 1) / kPlanBlock)), dim3(kPlanBlock), 0, st, rb,
                           reinterpret_cast<const float4*>(ws->A4), ws->m, P, L, N, Mz, mean);
    }
    if (best && best_path)
        hipLaunchKernelGGL(plan_pick_kernel, dim3(P), dim3(kPlanBlock), 0, st, rb, f, logp, S, L, N, best, best_path);
    if (samples) {
        const size_t n = (size_t)P * S * N * L;
        hipLaunchKernelGGL(plan_samples_kernel, dim3((unsigned)((n + kPlanBlock - 1) / kPlanBlock)), dim3(kPlanBlock), 0, st, rb,
                           f, (size_t)P * S, L, N, samples);
    }
    if (ee_var)
        hipLaunchKernelGGL(plan_ee_variance_kernel, dim3((P * N + 63) / 64), dim3(64), 0, st, rb, f, P, S, L, N, ee_var);
    return (int)hipGetLastError();
}
