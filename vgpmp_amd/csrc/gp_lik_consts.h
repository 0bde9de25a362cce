// Likelihood constants (sigma_obs, alpha) as trainable variables.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

// ---- likelihood constants as trainable variables (vgpmp_lik_params) -----------------------------------------
constexpr double kAlphaFloor = 1e-4, kSigmaFloor = 1e-5;      // models/vgpmp.py:82, likelihoods/likelihood.py:31,41

struct LikConstArgs {
    const double *raw_alpha, *raw_sigma;
    vg_lik_scratch sc;
    double inv_s;          // 1 / S_total
};
// effective constants from the raw variables (start of every call): one wave per problem, one lane per sphere
__global__ __launch_bounds__(VGPMP_MAX_SPHERES) void lik_consts_kernel(LikConstArgs a) {
    const int p = blockIdx.x, q = threadIdx.x;
    if (q == 0) {
        const double al = (kAlphaFloor + softplus_d(a.raw_alpha[p])) * a.inv_s;
        a.sc.alpha_fin[p] = al;
        a.sc.alpha_eff[p] = (float)al;
    }
    a.sc.sigma_eff[(size_t)p * VGPMP_MAX_SPHERES + q] = (float)(kSigmaFloor + softplus_d(a.raw_sigma[(size_t)p * VGPMP_MAX_SPHERES + q]));
}

struct LikUpdArgs {
    const vgpmp_robot* rb;
    const float *lik_partial, *sig_partial;
    int nblk;
    double inv_s;          // 1 / S_total
    double *raw_alpha, *raw_sigma, *m_alpha, *v_alpha, *m_sigma, *v_sigma, *g_alpha, *g_sigma;
    vg_lik_scratch sc;
    int do_adam, trainable;
    const uint32_t* ctr;   // ticked device counter (then the step size comes from it), else lr_t
    double lr, lr_t;
};
// gradient of the training loss wrt (raw_alpha, raw_sigma) of one problem, Adam, and the constants of the next step.
//   loss = -(ELBO + log sigmoid(raw_alpha) + sum_q log sigmoid(raw_sigma_q))      (vgpmp.h: vgpmp_lik_params)
//   d ELBO / d alpha = (1/S) sum_{s,n} logp,   d ELBO / d sigma_q = (alpha/S) 1/2 sum_{s,n} c_q^2 / sigma_q^2
// Four waves per problem, lane q = sphere q, wave w = every fourth workgroup of the likelihood: sums in a fixed order (per wave
// ascending, 32 partials requested together; then ((w0 + w1) + (w2 + w3))).  One wave walking all 896 partials of config 2
// eight at a time was 112 serial round trips: 36 us per launch.
constexpr int kLikUpdWaves = 4;
__global__ __launch_bounds__(kLikUpdWaves * VGPMP_MAX_SPHERES) void lik_update_kernel(LikUpdArgs a) {
    static_assert(VGPMP_MAX_SPHERES == VG_WAVE, "lane = sphere");
    __shared__ double part[2][kLikUpdWaves][VGPMP_MAX_SPHERES];
    const int p = blockIdx.x, q = threadIdx.x & (VG_WAVE - 1), w = threadIdx.x >> 6, nsph = a.rb->num_spheres;
    double ls = 0.0;
    for (int b = threadIdx.x; b < a.nblk; b += kLikUpdWaves * VGPMP_MAX_SPHERES) ls += (double)a.lik_partial[(size_t)p * a.nblk + b];
    ls = vg_wave_sum(ls);
    double c2 = 0.0;
    const float* sp = a.sig_partial + (size_t)p * a.nblk * VGPMP_MAX_SPHERES + q;
    for (int b0 = w; b0 < a.nblk; b0 += 32 * kLikUpdWaves) {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = sp[(size_t)min(b0 + k * kLikUpdWaves, a.nblk - 1) * VGPMP_MAX_SPHERES];
#pragma unroll
        for (int k = 0; k < 32; ++k) if (b0 + k * kLikUpdWaves < a.nblk) c2 += (double)v[k];
    }
    part[0][w][q] = c2; part[1][w][q] = ls;
    __syncthreads();
    if (w != 0) return;
    c2 = (part[0][0][q] + part[0][1][q]) + (part[0][2][q] + part[0][3][q]);      // sum_{s,n} c_q^2 / sigma_q
    ls = (part[1][0][q] + part[1][1][q]) + (part[1][2][q] + part[1][3][q]);      // sum_{s,n} logp
    const size_t pq = (size_t)p * VGPMP_MAX_SPHERES + q;
    const double lr_t = a.ctr ? adam_step_size(a.lr, (double)*a.ctr) : a.lr_t;
    double ra = a.raw_alpha[p], rs = a.raw_sigma[pq];
    const double alpha = kAlphaFloor + softplus_d(ra), sigma = kSigmaFloor + softplus_d(rs);
    const double gs = q < nsph ? -(alpha * a.inv_s * 0.5 * c2 / sigma * sigmoid_d(rs) + sigmoid_d(-rs)) : 0.0;
    a.g_sigma[pq] = gs;
    if (a.do_adam && (a.trainable & VGPMP_TRAIN_SIGMA_OBS) && q < nsph) adam_update(&rs, a.m_sigma + pq, a.v_sigma + pq, gs, lr_t);
    if (a.do_adam && (a.trainable & VGPMP_TRAIN_SIGMA_OBS)) a.raw_sigma[pq] = rs;
    a.sc.sigma_eff[pq] = (float)(kSigmaFloor + softplus_d(rs));
    if (q == 0) {
        const double ga = -(ls * a.inv_s * sigmoid_d(ra) + sigmoid_d(-ra));
        a.g_alpha[p] = ga;
        a.sc.alpha_fin[p] = alpha * a.inv_s;
        if (a.do_adam && (a.trainable & VGPMP_TRAIN_ALPHA)) {
            adam_update(&ra, a.m_alpha + p, a.v_alpha + p, ga, lr_t);
            a.raw_alpha[p] = ra;
        }
        a.sc.alpha_eff[p] = (float)((kAlphaFloor + softplus_d(ra)) * a.inv_s);
    }
}

}  // namespace
