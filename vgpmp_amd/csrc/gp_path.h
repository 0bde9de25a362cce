// Workspace layout and launchers of the GP half of the ELBO step (gp_path.hip).
#pragma once
#include "vgpmp_device.h"

// samples per workgroup of the path kernels.  A 32-sample variant exists (template parameter) but measured
// 10 % SLOWER at 64 problems/GPU (fewer workgroups in flight per CU), so 8 is used everywhere.
inline int vg_sc(const vgpmp_dims*) { return 8; }

struct vg_workspace {
    // float64 covariance path, per (problem, latent)
    double *ell, *var, *sig_ell, *sig_var;   // [P,L] constrained values and d(constrained)/d(raw)
    double *Kinv;            // [P,L,Mz,Mz]   (Kuu + jitter I)^-1
    double *Kd_ell;          // [P,L,Mz,Mz]   dKuu / d lengthscale
    double *Ks64, *Lk64, *Li64;  // [P,L,Mz,Mz] Kuu, chol(Kuu + jI) and its inverse (stage A -> stage B)
    double *kl_l;            // [P,L]
    double *gkl_qmu;         // [P,L,M]       dKL/dq_mu
    double *gkl_Q;           // [P,L,M,M]     dKL/dq_sqrt (lower)
    double *gkl_ell, *gkl_var;   // [P,L]     dKL/d lengthscale, d variance (constrained space)
    // float32 operands of the sample path
    float *A4;               // [P,L,N,Mz,4]  {A, dA/dell, dA/dvar, 0},  A = Kfu (Kuu + jI)^-1
    float *AT;               // [P,L,Mz,N]    A transposed
    float *C;                // [P,L,Mz,Mz]   q_sqrt (full)
    float *CT;               // [P,L,Mz,Mz]   q_sqrt^T
    float *CT_ell, *CT_var;  // [P,L,Mz,Mz]   (dC/d theta)^T
    float *Lk32;             // [P,L,Mz,Mz]   chol(Kuu + jitter I)
    float *m;                // [P,L,Mz]
    float *Phi, *dPhi;       // [P,L,J,B]
    float *F0, *H;           // [SK][P,S,L,J]
    float *R;                // [P,S,L,Mz]
    float *U;                // [P,S,L,Mz]   m + C eps, formed by stage B when the likelihood assembles the paths itself
    float *epsT, *eps2T;     // [P,L,S,Mz]   eps, eps' again, rows of a latent contiguous (stage B's U role, paths_fwd_regs, paths_bwd_regs)
    float *G;                // [P,S,L,N]     dloss/df
    float *lik_partial;      // [P,nblk]
    float *part;             // [P,L,NC,PART] per-chunk reductions of the reverse pass
    double *lr_t;            // [P]           [0]: bias-corrected Adam step size of the running step (set at the counter tick)
    double *theta_next;      // [P,L,6]       updated hyper-parameters + moments between stage 1 and stage 2
    double *prev_var, *prev_sig_ell, *prev_sig_var;   // [P,L] var / softplus slopes of the previous step
};

inline int vg_mz(const vgpmp_dims* d) { return d->M + 2; }
inline int vg_j(const vgpmp_dims* d) { return d->N + d->M + 2; }
inline int vg_chunks(const vgpmp_dims* d) { return (d->S + vg_sc(d) - 1) / vg_sc(d); }
inline size_t vg_part_len(const vgpmp_dims* d) {
    size_t mz = (size_t)vg_mz(d);
    return mz + mz * mz + 8;      // dm, dC, two sets of {s_ell, s_var, s_rff, -}
}

// vgpmp_lik_params.scratch: effective likelihood constants and the per-workgroup per-sphere sums of the likelihood
struct vg_lik_scratch {
    double* alpha_fin;     // [P]  alpha / S of the step whose ELBO pieces `final` reports
    float* alpha_eff;      // [P]  alpha / S the next likelihood launch uses
    float* sigma_eff;      // [P, MAX_SPHERES]
    float* sig_partial;    // [P, blocks per problem, MAX_SPHERES]  sum over a workgroup's configurations of c^2 / sigma
};
size_t vg_layout_lik_scratch(const vgpmp_dims* d, void* base, vg_lik_scratch* out);

// vgpmp_inducing_params.scratch: per-chunk reductions of the reverse pass wrt the inducing locations
constexpr int kIndBChunk = 64;      // Fourier bases per workgroup of the feature part
struct vg_ind_scratch {
    float* dA;             // [P, L, NC, N, Mz]   sum over a chunk's samples of G^T R
    float* dC;             // [P, L, NC, Mz, Mz]  sum over a chunk's samples of dR^T eps
    float* dR;             // [P, S, L, Mz]       G A
    float* rff;            // [P, L, B / kIndBChunk, Mz, L]  feature part of d loss / d Zy, per basis chunk
    double* cov;           // [P, L, Mz]          covariance part of d loss / d Zy[:, l]
    double* mt_part;       // [P, L, tiles, Mz, Mz]  A^T dA of a tile of time points (z_cov_rows_kernel)
    double* gz_part;       // [P, L, tiles, Mz]      the tile's part of d loss / d Zy through Kfu
};
constexpr int kIndRowTile = 16;      // time points per workgroup of z_cov_rows_kernel
size_t vg_layout_ind_scratch(const vgpmp_dims* d, void* base, vg_ind_scratch* out);
struct vg_ind_launch {
    const vgpmp_dims* d;
    const vgpmp_inducing_params* ind;
    const vg_workspace* ws;
    const vgpmp_noise* nz;
    const vgpmp_params* params;
    const double *X, *y_u;
    double jitter;
    int do_adam, trainable;
    const uint32_t* ctr;   // ticked device counter (then the step size comes from it), else lr_t
    double lr, lr_t;
};
int vg_launch_inducing_build(const vgpmp_dims* d, const vgpmp_inducing_params* ind, hipStream_t st);
int vg_launch_inducing_backward(const vg_ind_launch& a, hipStream_t st);

int vg_check_dims(const vgpmp_dims* d);
int vg_backward_fits(const vgpmp_dims* d);
size_t vg_layout_workspace(const vgpmp_dims* d, void* base, vg_workspace* ws);
int vg_workspace_lookup(const vgpmp_dims* d, const vg_workspace* ws, const char* name, void** ptr, size_t* count,
                        int32_t* is_double);
int vg_launch_rng(const vgpmp_dims* d, const vgpmp_noise* noise, uint32_t seed, uint32_t problem_base, uint32_t step,
                  const uint32_t* ctr, hipStream_t st, float* epsT = nullptr, float* eps2T = nullptr);
int vg_launch_adam(const vgpmp_dims* d, const vgpmp_params* params, const vgpmp_params* grad, const vgpmp_params* am,
                   const vgpmp_params* av, int trainable, double lr, int t, hipStream_t st);
int vg_elbo_steps(const vgpmp_dims* d, const vgpmp_robot* rb, const vgpmp_sdf* sdf, const vgpmp_problem* pb,
                  const vgpmp_params* params, const vgpmp_params* am, const vgpmp_params* av, const vgpmp_noise* noise,
                  const vgpmp_outputs* out, const vg_workspace* ws, int what, int trainable, double lr, int adam_t,
                  uint32_t seed, uint32_t problem_base, uint32_t step, int num_steps, hipStream_t st, hipEvent_t* ev);
int vg_launch_sample_paths(const vgpmp_dims* d, const vgpmp_robot* rb, const vg_workspace* ws, const float* f, const float* logp,
                           float* mean, int32_t* best, float* best_path, float* samples, float* ee_var, hipStream_t st);
constexpr int VG_NUM_STAGES = 8;   // cov_fwd, rng, features, prior_gemm, paths_fwd, loglik, paths_bwd, final
