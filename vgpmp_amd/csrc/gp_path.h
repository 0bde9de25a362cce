// Workspace layout and launchers of the GP half of the ELBO step (gp_path.hip).
#pragma once
#include "vgpmp_device.h"

constexpr int VG_SC = 16;   // samples per chunk of the path kernels

struct vg_workspace {
    // float64 covariance path, per (problem, latent)
    double *ell, *var;       // [P,L]
    double *K;               // [P,L,Mz,Mz]   Matern Kuu without jitter
    double *Lk;              // [P,L,Mz,Mz]   chol(Kuu + jitter I)
    double *Linv;            // [P,L,Mz,Mz]   Lk^-1
    double *Kinv;            // [P,L,Mz,Mz]   (Kuu + jitter I)^-1
    double *Kuf;             // [P,L,Mz,N]
    double *A64;             // [P,L,N,Mz]    Kfu Kinv
    double *afull;           // [P,L,Mz]      Lk^-1 (q_mu - p_mu)
    double *cvec;            // [P,L,2]       Kyy^-1 y
    double *kl_l;            // [P,L]
    double *dA64;            // [P,L,N,Mz]    reverse-pass scratch
    // float32 operands of the sample path
    float *A, *C, *m;        // [P,L,N,Mz], [P,L,Mz,Mz], [P,L,Mz]
    float *Phi, *dPhi;       // [P,L,J,B]
    float *F0, *H;           // [SK][P,S,L,J]
    float *R;                // [P,S,L,Mz]
    float *G;                // [P,S,L,N]     dloss/df
    float *lik_partial;      // [P,nblk]
    float *part;             // [P,L,NC,PART] per-chunk reductions of the reverse pass
};

inline int vg_mz(const vgpmp_dims* d) { return d->M + 2; }
inline int vg_j(const vgpmp_dims* d) { return d->N + d->M + 2; }
inline int vg_chunks(const vgpmp_dims* d) { return (d->S + VG_SC - 1) / VG_SC; }
inline size_t vg_part_len(const vgpmp_dims* d) {
    size_t mz = (size_t)vg_mz(d);
    return mz + mz * mz + (size_t)d->N * mz + 4;
}

int vg_check_dims(const vgpmp_dims* d);
size_t vg_layout_workspace(const vgpmp_dims* d, void* base, vg_workspace* ws);
int vg_workspace_lookup(const vgpmp_dims* d, const vg_workspace* ws, const char* name, void** ptr, size_t* count,
                        int32_t* is_double);
int vg_launch_rng(const vgpmp_dims* d, const vgpmp_noise* noise, uint32_t seed, uint32_t problem_base, uint32_t step,
                  const uint32_t* ctr, hipStream_t st);
int vg_launch_adam(const vgpmp_dims* d, const vgpmp_params* params, const vgpmp_params* grad, const vgpmp_params* am,
                   const vgpmp_params* av, int trainable, double lr, int t, hipStream_t st);
int vg_elbo_step(const vgpmp_dims* d, const vgpmp_robot* rb, const vgpmp_sdf* sdf, const vgpmp_problem* pb,
                 const vgpmp_params* params, const vgpmp_params* am, const vgpmp_params* av, const vgpmp_noise* noise,
                 const vgpmp_outputs* out, const vg_workspace* ws, int what, int trainable, double lr, int adam_t,
                 uint32_t seed, uint32_t problem_base, uint32_t step, hipStream_t st, hipEvent_t* ev);
constexpr int VG_NUM_STAGES = 8;   // rng, cov_fwd, features, prior_gemm, paths_fwd, loglik, paths_bwd, cov_bwd
