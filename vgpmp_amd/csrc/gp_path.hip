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
#include "gp_math.h"
#include <hip/hip_ext.h>
#include <string.h>

#ifdef VGPMP_BISECT
#include <stdlib.h>
#define VG_STOP(args, k) do { if ((args).stop == (k)) return; } while (0)
static int vg_bisect_stop(const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; }
int vg_trace_take_gp(unsigned long long* host, int cap) { return vg_trace_take(host, cap); }
#else
#define VG_STOP(args, k) do { } while (0)
#endif
// ---- schedule constants (each measured; the sweeps are in profiles/r03/final/problems_sweep.txt and DESIGN section 3)
#ifndef VG_JCHUNK_ONE
#define VG_JCHUNK_ONE 8
#endif
constexpr int kJChunkOne = VG_JCHUNK_ONE;          // points per workgroup of the feature role at one problem
constexpr int kHMt2MinTiles = 300;     // f16-split prior kernel: 128-row tiles above this many of them (36 problems: 330.5 -> 325.5 us per step)
constexpr int kRowsTpwWgs = 256;       // stage B rows role: two tiles per workgroup once the launch has this many workgroups (24 problems: 405.6 -> 392.7 us)
constexpr int kRowsTpwMax = 2;         // ... and never more (config-5 share: 976 / 965 / 1041 us per step with 1 / 2 / 4)
#ifndef VG_B_SPLIT
#define VG_B_SPLIT 1
#endif
#ifndef VG_FUSE_FWD
#define VG_FUSE_FWD 1
#endif
#ifndef VG_PB_MIN_WGS
#define VG_PB_MIN_WGS 1536
#endif
constexpr int kPbMinWgs = VG_PB_MIN_WGS;        // reverse path pass: chunks per workgroup double while this many workgroups remain (six per CU)

#include "gp_common.h"
#include "gp_rng.h"
#include "gp_paths.h"
#include "gp_update.h"
#include "gp_cov.h"
#include "gp_prior.h"
#include "gp_prior_split.h"
#include "gp_lik_consts.h"

namespace {

// =================================================================================================
// Role-dispatched launches for the few-problem regime.  One problem offers ~100 workgroups per kernel
// on a 256-CU part and every launch pays ~2.5 us of dispatch plus a first touch of data another XCD wrote, while
// overlapping kernels across HIP streams costs ~11 us per cross-stream edge on this platform (measured,
// tools/anyorder_probe.hip; hipExtAnyOrderLaunch is not honoured on gfx9).  So independent kernels of one
// dependency level are issued as ONE launch whose workgroup index selects the role:
//   stage 1   cov_a(t) | final(t-1) | eps(t) | features(t)     <- after hyper(t-1)
//   stage 2   cov_b(t) | prior GEMM(t)
//   stage 3   paths_fwd(t) | omega, beta, w of step t+1
// then loglik, paths_bwd and hyper as before.  Long roles come first in the grid so that they start first.
// =================================================================================================
struct Stage1Args {
    CovArgs cov; FinalArgs fin; RngArgs rng; FeatArgs feat;
    int n_cov, n_fin, n_eps, eps_gx, feat_gx, feat_gy;
    int fin_split;            // the q_mu / q_sqrt update on kFinSplit workgroups per (latent, problem): final_cols_body
    int skip;                 // measurement builds: bit mask of roles that return at once
    int n_feat;
};
template <bool PRO>
__global__ __launch_bounds__(kBlock, 2) void stage1_kernel(Stage1Args a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) { if (!(a.skip & 1)) cov_a_body(a.cov, sm, b % a.cov.L, b / a.cov.L); VG_TMAX(160); return; }
    b -= a.n_cov;
    if (b < a.n_fin) {
        if (a.skip & 2) return;
        if (a.fin_split) {
            const int q = b % kFinSplit;
            b /= kFinSplit;
            final_cols_body(a.fin, sm, q, b % a.fin.L, b / a.fin.L);
        } else {
            final_body(a.fin, sm, b % a.fin.L, b / a.fin.L);
        }
        VG_TMAX(161);
        return;
    }
    b -= a.n_fin;
    // the eps draws ahead of the feature role (one or two problems: everything is resident at once, and the workgroups at the
    // END of the grid were the last to finish -- 1.4 us behind cov_a)
    if (b < a.n_eps) {
        if (!(a.skip & 4)) {
            if (a.rng.epsT) rng_eps_t_body(a.rng, b % a.eps_gx, b / a.eps_gx, reinterpret_cast<float*>(sm));
            else rng_normals_body(a.rng, b % a.eps_gx, b / a.eps_gx, 0u, a.rng.nE);
        }
        VG_TMAX(162);
        return;
    }
    b -= a.n_eps;
    if (a.skip & 8) return;
    const int bx = b % a.feat_gx;
    b /= a.feat_gx;
    features_body<PRO>(a.feat, bx, b % a.feat_gy, b / a.feat_gy);
    VG_TMAX(163);
}

struct Stage2Args {
    CovArgs cov; GemmArgs gemm;
    int n_cov, cov_roles, gemm_gx, gemm_gy, gemm_per_xcd;
    int skip;
    int n_gemm;               // stage2_gemm_first_kernel
};
template <bool TANGENTS, int KS>
__global__ __launch_bounds__(kBlock, 2) void stage2_kernel(Stage2Args a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) {
        if (a.skip & 1) return;
        const int role = cov_role_rotated(b, a.cov_roles);
        b /= a.cov_roles;
        cov_b_body<TANGENTS>(a.cov, sm, role, b % a.cov.L, b / a.cov.L);
        return;
    }
    b -= a.n_cov;
    if (a.skip & 2) return;
    // Workgroups go to the 8 XCDs round robin: give each XCD a
    // CONTIGUOUS range of GEMM tiles, so the column / row tiles that share operands share an L2 as well.
    if (a.gemm_per_xcd > 0) b = (b & 7) * a.gemm_per_xcd + (b >> 3);
    const int bx = b % a.gemm_gx;
    b /= a.gemm_gx;
    if constexpr (KS < 0) prior_gemm_lds_body<KS == -2>(a.gemm, reinterpret_cast<float*>(sm), bx, b % a.gemm_gy, b / a.gemm_gy);
    else prior_gemm_body<KS>(a.gemm, bx, b % a.gemm_gy, b / a.gemm_gy);
}
// The same launch with the GEMM tiles at the FRONT of the grid, for whole-K tiles (no K-slices: 3 problems up).  They are
// the longest workgroups of the launch then and must not start behind the short covariance roles: 3 problems
// 113 -> 107 us per step, 4: 131 -> 121.  (With K-slices -- one or two problems -- the covariance chains are the longer
// workgroups and the order above is the better one by ~1 us.)
template <bool TANGENTS, int KS>
__global__ __launch_bounds__(kBlock, 2) void stage2_gemm_first_kernel(Stage2Args a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b >= a.n_gemm) {
        b -= a.n_gemm;
        const int role = cov_role_rotated(b, a.cov_roles);
        b /= a.cov_roles;
        cov_b_body<TANGENTS>(a.cov, sm, role, b % a.cov.L, b / a.cov.L);
        return;
    }
    if (a.gemm_per_xcd > 0) b = (b & 7) * a.gemm_per_xcd + (b >> 3);
    const int bx = b % a.gemm_gx;
    b /= a.gemm_gx;
    if constexpr (KS < 0) prior_gemm_lds_body<KS == -2>(a.gemm, reinterpret_cast<float*>(sm), bx, b % a.gemm_gy, b / a.gemm_gy);
    else prior_gemm_body<KS>(a.gemm, bx, b % a.gemm_gy, b / a.gemm_gy);
}

struct Stage3Args {
    PathArgs path; RngArgs rng;
    int n_path, n_basis, basis_gx, w_gx;
    int skip;
};
template <int SK, bool RAW>
__global__ __launch_bounds__(kBlock, 2) void stage3_kernel(Stage3Args a) {
    extern __shared__ float smf[];
    int b = blockIdx.x;
    if (b < a.n_path) {
        if (a.skip & 1) return;
        b = xcd_contiguous(b, a.path.xcd_span);
        const int nch = a.path.NC * a.path.nsplit;
        const int ch = b % nch;
        b /= nch;
        if (SK > 1 && a.path.nsplit == 2) {
            if constexpr (SK > 1) {
                if (a.path.Mz == 32) paths_fwd_split_body<SK, 32>(a.path, smf, ch, b % a.path.L, b / a.path.L);
                else paths_fwd_split_body<SK>(a.path, smf, ch, b % a.path.L, b / a.path.L);
            }
            return;
        }
        paths_fwd_body<SK, 8, RAW>(a.path, smf, ch, b % a.path.L, b / a.path.L);
        return;
    }
    b -= a.n_path;
    if (a.skip & 2) return;
    if (b < a.n_basis) { rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx, smf); return; }
    b -= a.n_basis;
    rng_normals_body(a.rng, b % a.w_gx, b / a.w_gx, a.rng.nW, 0u);
}

// The step without stage 3 (few problems, Mz = 32, K-slices): the likelihood assembles its own paths (fk_sdf.hip,
// loglik_paths_wide_kernel<8, SIG, SK>) from U = m + C eps that stage B leaves, and the draws of the next step's omega, beta, w
// -- the other role of stage 3 -- ride with the reverse path pass instead: four launches per step.
struct Stage4Args {
    PathArgs path; RngArgs rng;
    int n_bwd, bwd_gx, n_basis, basis_gx, w_gx;
};
template <int SK, int MZ>
__global__ __launch_bounds__(kBlock, 2) void stage4_kernel(Stage4Args a) {
    extern __shared__ float smf[];
    int b = blockIdx.x;
    if (b < a.n_bwd) { paths_bwd_split_body<SK, MZ>(a.path, smf, b, a.bwd_gx); return; }
    b -= a.n_bwd;
    if (b < a.n_basis) { rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx, smf); return; }
    b -= a.n_basis;
    rng_normals_body(a.rng, b % a.w_gx, b / a.w_gx, a.rng.nW, 0u);
}

// ---- medium batches (5 problems up, one launch per kernel): independent kernels of a dependency level share a
// launch here too.  The latency-bound ones (cov_a, cov_b, hyper-parameter update, final) then run beside the
// throughput-bound ones (noise draws, tiled GEMM) instead of in front of them: 7 launches per step instead of 11.
struct MidAArgs {            // cov_a | omega, beta | w, eps, eps'
    CovArgs cov; RngArgs rng;
    int n_cov, n_basis, basis_gx, n_gx, n_norm;      // n_norm: workgroups of rng_normals_body (n_gx per problem); behind them rng_eps_t_body's
    int e_gx;
};
__global__ __launch_bounds__(kCovThreads, 2) void mid_cov_a_rng_kernel(MidAArgs a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) { cov_a_body(a.cov, sm, b % a.cov.L, b / a.cov.L); return; }
    b -= a.n_cov;
    if (b < a.n_basis) { rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx, reinterpret_cast<float*>(sm)); return; }
    b -= a.n_basis;
    if (b < a.n_norm) { rng_normals_body(a.rng, b % a.n_gx, b / a.n_gx, a.rng.nW, a.rng.nE); return; }
    b -= a.n_norm;
    rng_eps_t_body(a.rng, b % a.e_gx, b / a.e_gx, reinterpret_cast<float*>(sm));
}

// Steps after the first of a call: the updates of step t - 1 ride in the first launch of step t, as in the few-problem schedule --
// the q_mu / q_sqrt gradient assembly + Adam as a role of its own BESIDE stage A and the draws, the hyper-parameter update as
// the prologue of stage A (cov_a_body, prologue 2) -- instead of a launch of their own between two steps: one dependency level
// and one launch boundary fewer per step, and the gradient assembly (per-CU ingest of the chunk partials) runs under the draws.
struct MidS1Args {
    MidAArgs a; FinalArgs fin; HyperArgs hy;
    int n_fin;
};
template <int MZCAP>
__global__ __launch_bounds__(kBlock, MZCAP <= 32 ? 4 : 1) void mid_stage1_kernel(MidS1Args s) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
#ifdef VGPMP_BISECT
    // (measurement build: start / end of eight workgroups of every role, evenly spaced -- ids 1700 / 1800 + 16 role + k; tools/step_trace.py)
    struct RoleStamp {
        int id;
        __device__ RoleStamp(int role, int b, int n) : id(-1) {
            const int step = n > 8 ? n / 8 : 1;
            if (b % step == 0 && b / step < 8) { id = 16 * role + b / step; VG_T(true, 1700 + id); }
        }
        __device__ ~RoleStamp() { if (id >= 0) VG_T(true, 1800 + id); }
    };
#define VG_ROLE_STAMP(role, b, n) RoleStamp rs_(role, b, n)
#else
#define VG_ROLE_STAMP(role, b, n) do { } while (0)
#endif
    const MidAArgs& a = s.a;
    // Stage A first: the longest chain of the launch since it also forms the rows of A (25 us alone, 45 with four of them on a CU).
    // (While the gradient assembly pulled sixteen sets of chunk partials per latent it belonged in front -- stage A first measured +2 %
    //  at 896 latent pairs; with one set per workgroup of the reverse pass: config 3 123.1 -> 121.9 us per step, the batches unchanged.
    //  Dealt out AMONG the draws -- every stride-th position of the rest of the grid -- the chains start late: config 3 202 us.)
    if (b < a.n_cov) { VG_ROLE_STAMP(1, b, a.n_cov); cov_a_body(a.cov, sm, b % a.cov.L, b / a.cov.L); return; }
    b -= a.n_cov;
    if (b < s.n_fin) {
        VG_ROLE_STAMP(0, b, s.n_fin);
        const HyperArgs& h = s.hy;
        const bool own = h.ctr && h.do_adam;
        FinalArgs fb = s.fin;      // the step size comes from the counter here, as in mid_hyper_final_kernel
        if (own) { fb.use_lr_dev = 0; fb.lr_t = adam_step_size(h.lr, (double)*h.ctr); }
        if constexpr (MZCAP <= 32) {
            final_body<MZCAP>(fb, sm, b % fb.L, b / fb.L);
        } else if (fb.split) {
            const int q = b % kFinSplit;
            b /= kFinSplit;
            final_cols_body(fb, sm, q, b % fb.L, b / fb.L);
        } else {
            final_body<MZCAP>(fb, sm, b % fb.L, b / fb.L);
        }
        return;
    }
    b -= s.n_fin;
    if (b < a.n_basis) { VG_ROLE_STAMP(2, b, a.n_basis); rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx, reinterpret_cast<float*>(sm)); return; }
    b -= a.n_basis;
    if (b < a.n_norm) { VG_ROLE_STAMP(3, b, a.n_norm); rng_normals_body(a.rng, b % a.n_gx, b / a.n_gx, a.rng.nW, a.rng.nE); return; }
    b -= a.n_norm;
    VG_ROLE_STAMP(4, b, (int)gridDim.x - s.n_fin - a.n_cov - a.n_basis - a.n_norm);
    rng_eps_t_body(a.rng, b % a.e_gx, b / a.e_gx, reinterpret_cast<float*>(sm));
}
#undef VG_ROLE_STAMP

// Few samples per problem (the f16-split few-sample prior kernel, S <= 32) in the batch schedule: stage B and the prior draws need
// only stage A and the noise, not each other -- ONE launch.  Stage B is four ~10 us chains per latent in 35 KB of LDS each (four
// workgroups per CU), the prior draws are 3-4 us workgroups with no LDS to speak of: behind the chains in the grid, they fill
// the slots the chains leave as they end instead of waiting for the launch boundary (config 3, 385 latents: stage B 19 us, then
// 2.5 us of boundary, then 7 us of draws -> 20 us together; profiles/r05/ab_runs.txt).
struct MidBArgs {
    CovArgs cov; FusedPriorArgs fp;
    int n_cov, latents, prior_gx;      // stage B: kCovFixedRoles x latents workgroups, role-major; then prior_gx x column-tile-pairs
};
template <int MT, int DM, bool DELL>
__global__ __launch_bounds__(kBlock, 4) void mid_cov_b_prior16_kernel(MidBArgs a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) {      // (order and roles of cov_b_kernel: the long roles first)
        const int ord = b / a.latents, lat = b - ord * a.latents;
        const int role = ord == 0 ? 2 : ord == 1 ? 1 : ord == 2 ? 0 : ord;
        VG_T((lat & 63) == 0 && lat < 512 && ord < 8, 1500 + 8 * ord + (lat >> 6));
        cov_b_body<true, false>(a.cov, sm, role, lat % a.cov.L, lat / a.cov.L);
        VG_T((lat & 63) == 0 && lat < 512 && ord < 8, 1400 + 8 * ord + (lat >> 6));
        return;
    }
    b -= a.n_cov;
    prior_small16_body<MT, DM, DELL, true>(a.fp, b % a.prior_gx, b / a.prior_gx);
}

// Large batches (the f16-split prior kernel, 512-thread workgroups): stage B -- KL, the two tangents, q_sqrt: 256-thread chains that need
// only stage A -- rides in the prior kernel's launch, BEHIND its tiles in the grid, two latents' instances of one role side by side
// in a workgroup (each half passes the same barriers; its own LDS region, reduction scratch and thread indices).  The prior tiles
// come in whole rounds of two per CU: at 896 latent pairs the second round leaves a quarter of the slots empty for the length of a
// tile (115 us), and a launch of its own for stage B (42 us, plus the boundary) follows -- here the chains take those slots.
struct PriorCovArgs {
    FusedBatchArgs fb; CovArgs cov; FusedFwdArgs fw;      // (fw: the path assembly as the tiles' epilogue, FWD)
    int nz_prior;            // grid.z of the prior tiles; behind them the stage-B pairs, role-major, the long roles first
    int latents, pairs;      // pairs = ceil(latents / 2) per role
    unsigned lds_half;       // bytes of LDS of one half's stage-B instance
};
template <bool DELL, int MT, bool FWD>
__global__ __launch_bounds__(kHThreads, 4) void prior_split_cov_b_kernel(PriorCovArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_lds[];
    if ((int)blockIdx.z < a.nz_prior) {
        prior_fused_split_body<DELL, MT, FWD>(a.fb, hs_lds, blockIdx.x, blockIdx.y, blockIdx.z, &a.fw);
        return;
    }
    __shared__ double red2[2][kCovThreads / VG_WAVE];
    const int b = (((int)blockIdx.z - a.nz_prior) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    if (b >= kCovFixedRoles * a.pairs) return;
    const int ord = b / a.pairs, pr = b - ord * a.pairs;
    const int role = ord == 0 ? 2 : ord == 1 ? 1 : ord == 2 ? 0 : ord;      // (the order of cov_b_kernel)
    const int half = (int)threadIdx.x >> 8, lat = min(2 * pr + half, a.latents - 1);      // (an odd count: the last latent twice, the same values)
    cov_b_body<true, false>(a.cov, reinterpret_cast<double*>(hs_lds + (size_t)half * a.lds_half), role, lat % a.cov.L, lat / a.cov.L,
                            (int)threadIdx.x & (kCovThreads - 1), kCovThreads, red2[half]);
}

struct MidGArgs {            // hyper-parameter update | q_mu, q_sqrt update + ELBO pieces
    HyperArgs hy; FinalArgs fin;
    int n_hyper;             // = problems
};
// MZCAP = 32 (Mz <= 32: every reference problem set): the gradient assembly's register arrays at half size, under 128 registers --
// four workgroups per CU, so the 960 workgroups of 896 latent pairs (config-5 share) are ONE round instead of a full and a thin one
template <int MZCAP>
__global__ __launch_bounds__(kBlock, MZCAP <= 32 ? 4 : 1) void mid_hyper_final_kernel(MidGArgs a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    const HyperArgs& h = a.hy;
    const bool own = h.ctr && h.do_adam;
    const double lr_own = own ? adam_step_size(h.lr, (double)*h.ctr) : 0.0;
    if (b < a.n_hyper) {
        const int p = b, l = threadIdx.x;
        if (own && p == 0 && l == 0) h.lr_store[0] = lr_own;
        if constexpr (MZCAP <= 32) {
            // the update by whole waves (hyper_update_wave: one load per lane, butterfly sums -- bit-identical to the one-lane form,
            // which keeps 16 chunks x 3 partials in registers: 165 of them, the reason this launch ran three workgroups per CU)
            for (int lw = threadIdx.x >> 6; lw < h.L; lw += kBlock / VG_WAVE) {
                const size_t plw = (size_t)p * h.L + lw;
                const HyperState o = hyper_update_wave(h, plw, own, lr_own);
                if ((threadIdx.x & (VG_WAVE - 1)) == 0) {
                    h.g_ell[plw] = o.g_ell;
                    h.g_var[plw] = o.g_var;
                    if (h.do_adam) {
                        h.p_ell[plw] = o.raw_ell; h.m_ell[plw] = o.m_ell; h.v_ell[plw] = o.v_ell;
                        h.p_var[plw] = o.raw_var; h.m_var[plw] = o.m_var; h.v_var[plw] = o.v_var;
                    }
                }
            }
            return;
        }
        if (l >= h.L) return;
        const size_t pl = (size_t)p * h.L + l;
        const HyperState o = hyper_update(h, pl, own, lr_own);
        h.g_ell[pl] = o.g_ell;
        h.g_var[pl] = o.g_var;
        if (h.do_adam) {
            h.p_ell[pl] = o.raw_ell; h.m_ell[pl] = o.m_ell; h.v_ell[pl] = o.v_ell;
            h.p_var[pl] = o.raw_var; h.m_var[pl] = o.m_var; h.v_var[pl] = o.v_var;
        }
        return;
    }
    b -= a.n_hyper;
    FinalArgs fb = a.fin;      // the step size comes from the counter here: the update role that stores it runs alongside
    if (own) { fb.use_lr_dev = 0; fb.lr_t = lr_own; }
    if constexpr (MZCAP <= 32) {      // (large batches only: never the column-strip form of small launches)
        final_body<MZCAP>(fb, sm, b % fb.L, b / fb.L);
    } else if (fb.split) {
        const int q = b % kFinSplit;
        b /= kFinSplit;
        final_cols_body(fb, sm, q, b % fb.L, b / fb.L);
    } else {
        final_body<MZCAP>(fb, sm, b % fb.L, b / fb.L);
    }
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

size_t vg_layout_lik_scratch(const vgpmp_dims* d, void* base, vg_lik_scratch* out) {
    char* cur = (char*)base;
    const bool real = base != nullptr;
    const size_t P = (size_t)d->num_problems, nb = (size_t)vg_loglik_blocks_per_problem(d->S, d->N);
    out->alpha_fin = carve<double>(cur, P, real);
    out->alpha_eff = carve<float>(cur, P, real);
    out->sigma_eff = carve<float>(cur, P * VGPMP_MAX_SPHERES, real);
    out->sig_partial = carve<float>(cur, P * nb * VGPMP_MAX_SPHERES, real);
    return (size_t)(cur - (char*)base) + 256;
}

int vg_check_dims(const vgpmp_dims* d) {
    if (d->num_problems < 1 || d->S < 1 || d->S_total < d->S || d->N < 1 || d->M < 1 || d->L < 1 || d->B < 16)
        return VGPMP_E_SHAPE;
    if (d->sample_offset < 0 || d->sample_offset + d->S > d->S_total) return VGPMP_E_SHAPE;
    if (d->M + 2 > VGPMP_MAX_MZ || d->L > VGPMP_MAX_DOF || (d->B % 16) != 0) return VGPMP_E_SHAPE;
    if (d->split_k != 1 && d->split_k != 2 && d->split_k != 4 && d->split_k != 8) return VGPMP_E_SHAPE;
    if ((d->B / d->split_k) % 16 != 0) return VGPMP_E_SHAPE;
    if (d->N > 4096) return VGPMP_E_SHAPE;
    // the path assembly keeps A = Kfu (Kuu + jI)^-1 of one latent ([N, Mz] float32) in LDS: N * Mz is bounded by the 160 KB
    // of a CU (include/vgpmp.h, "Limits"); the reverse pass holds four such images (vg_backward_fits)
    const size_t Mz = (size_t)d->M + 2, N = (size_t)d->N, J = N + Mz, SC = 8;
    if ((Mz * (Mz + 1) + Mz * N + 3 * SC * Mz + Mz + SC * J + 24) * sizeof(float) > 160 * 1024) return VGPMP_E_SHAPE;
    return 0;
}

// VGPMP_DO_BACKWARD: {A, dA/dell, dA/dvar} and G A of one latent live in LDS together
int vg_backward_fits(const vgpmp_dims* d) {
    const size_t Mz = (size_t)d->M + 2, N = (size_t)d->N, J = N + Mz, SC = 8;
    return (4 * N * Mz + 2 * Mz * Mz + SC * N + 2 * SC * J + 8 * SC * Mz + 40) * sizeof(float) <= 160 * 1024;
}

size_t vg_layout_workspace(const vgpmp_dims* d, void* base, vg_workspace* ws) {
    const bool real = base != nullptr;
    char* cur = real ? (char*)base : (char*)(uintptr_t)256;
    char* start = cur;
    const size_t P = d->num_problems, L = d->L, N = d->N, Mz = vg_mz(d), J = vg_j(d), S = d->S, B = d->B, M = d->M;
    const size_t PL = P * L;
    ws->ell = carve<double>(cur, PL, real);
    ws->var = carve<double>(cur, PL, real);
    ws->sig_ell = carve<double>(cur, PL, real);
    ws->sig_var = carve<double>(cur, PL, real);
    ws->Kinv = carve<double>(cur, PL * Mz * Mz, real);
    ws->Kd_ell = carve<double>(cur, PL * Mz * Mz, real);
    ws->Ks64 = carve<double>(cur, PL * Mz * Mz, real);
    ws->Lk64 = carve<double>(cur, PL * Mz * Mz, real);
    ws->Li64 = carve<double>(cur, PL * Mz * Mz, real);
    ws->kl_l = carve<double>(cur, PL, real);
    ws->gkl_qmu = carve<double>(cur, PL * M, real);
    ws->gkl_Q = carve<double>(cur, PL * M * M, real);
    ws->gkl_ell = carve<double>(cur, PL, real);
    ws->gkl_var = carve<double>(cur, PL, real);
    ws->A4 = carve<float>(cur, PL * N * Mz * 4, real);
    ws->AT = carve<float>(cur, PL * N * Mz, real);
    ws->C = carve<float>(cur, PL * Mz * Mz, real);
    ws->CT = carve<float>(cur, PL * Mz * Mz, real);
    ws->CT_ell = carve<float>(cur, PL * Mz * Mz, real);
    ws->CT_var = carve<float>(cur, PL * Mz * Mz, real);
    ws->Lk32 = carve<float>(cur, PL * Mz * Mz, real);
    ws->m = carve<float>(cur, PL * Mz, real);
    ws->Phi = carve<float>(cur, PL * J * B, real);
    ws->dPhi = carve<float>(cur, PL * J * B, real);
    ws->F0 = carve<float>(cur, (size_t)d->split_k * P * S * L * J, real);
    ws->H = carve<float>(cur, (size_t)d->split_k * P * S * L * J, real);
    ws->R = carve<float>(cur, P * S * L * Mz, real);
    ws->U = carve<float>(cur, P * S * L * Mz, real);
    ws->epsT = carve<float>(cur, P * S * L * Mz, real);
    ws->eps2T = carve<float>(cur, P * S * L * Mz, real);
    ws->G = carve<float>(cur, P * S * L * N, real);
    ws->lik_partial = carve<float>(cur, P * (size_t)vg_loglik_blocks_per_problem(d->S, d->N), real);
    ws->part = carve<float>(cur, PL * vg_chunks(d) * vg_part_len(d), real);
    ws->lr_t = carve<double>(cur, P, real);
    ws->theta_next = carve<double>(cur, PL * 6, real);
    ws->prev_var = carve<double>(cur, PL, real);
    ws->prev_sig_ell = carve<double>(cur, PL, real);
    ws->prev_sig_var = carve<double>(cur, PL, real);
    return (size_t)(cur - start) + 256;
}

int vg_workspace_lookup(const vgpmp_dims* d, const vg_workspace* ws, const char* name, void** ptr, size_t* count,
                        int32_t* is_double) {
    const size_t P = d->num_problems, L = d->L, N = d->N, Mz = vg_mz(d), J = vg_j(d), S = d->S, B = d->B;
    *is_double = 0;
    if (!strcmp(name, "A4")) { *ptr = ws->A4; *count = P * L * N * Mz * 4; }
    else if (!strcmp(name, "C")) { *ptr = ws->C; *count = P * L * Mz * Mz; }
    else if (!strcmp(name, "m")) { *ptr = ws->m; *count = P * L * Mz; }
    else if (!strcmp(name, "Phi")) { *ptr = ws->Phi; *count = P * L * J * B; }
    else if (!strcmp(name, "F0")) { *ptr = ws->F0; *count = (size_t)d->split_k * P * S * L * J; }
    else if (!strcmp(name, "H")) { *ptr = ws->H; *count = (size_t)d->split_k * P * S * L * J; }
    else if (!strcmp(name, "R")) { *ptr = ws->R; *count = P * S * L * Mz; }
    else if (!strcmp(name, "epsT")) { *ptr = ws->epsT; *count = P * S * L * Mz; }
    else if (!strcmp(name, "eps2T")) { *ptr = ws->eps2T; *count = P * S * L * Mz; }
    else if (!strcmp(name, "G")) { *ptr = ws->G; *count = P * S * L * N; }
    else if (!strcmp(name, "kl_l")) { *ptr = ws->kl_l; *count = P * L; *is_double = 1; }
    else if (!strcmp(name, "Kinv")) { *ptr = ws->Kinv; *count = P * L * Mz * Mz; *is_double = 1; }
    else return VGPMP_E_ARG;
    return 0;
}

static RngArgs make_rng_args(const vgpmp_dims* d, const vgpmp_noise* nz, uint32_t seed, uint32_t problem_base,
                             uint32_t step, const uint32_t* ctr, uint32_t bias) {
    RngArgs r;
    r.L = d->L; r.B = d->B; r.D = d->L;
    r.nW = (uint32_t)d->S * d->L * d->B;                 // nW % 16 == 0 since B % 16 == 0
    r.nE = (uint32_t)d->S * vg_mz(d) * d->L;
    r.wOff = (uint32_t)d->sample_offset * d->L * d->B;
    r.eOff = (uint32_t)d->sample_offset * vg_mz(d) * d->L;
    r.omega = nz->omega; r.beta = nz->beta; r.w = nz->w; r.eps = nz->eps; r.eps2 = nz->eps2;
    r.seed = seed; r.problem_base = problem_base; r.step = step; r.bias = bias; r.ctr = ctr;
    r.epsT = nullptr; r.eps2T = nullptr; r.Mz = vg_mz(d); r.S = d->S;
    r.eps_rows_log2 = 6;
    return r;
}

int vg_launch_rng(const vgpmp_dims* d, const vgpmp_noise* nz, uint32_t seed, uint32_t problem_base, uint32_t step,
                  const uint32_t* ctr, hipStream_t st, float* epsT, float* eps2T) {
    const int P = d->num_problems;
    RngArgs r = make_rng_args(d, nz, seed, problem_base, step, ctr, 0u);
    r.epsT = epsT; r.eps2T = eps2T;
    VG_GGL(rng_basis_kernel, dim3((r.L * r.B + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
    if (epsT) {      // eps / eps' in both layouts by their own launch, w alone by the other
        VG_GGL(rng_eps_t_kernel, dim3(rng_eps_t_blocks((uint32_t)r.S * r.Mz, r.eps_rows_log2), P), dim3(kBlock), 0, st, r);
        r.nE = 0;
    }
    const uint32_t nthr = rng_normal_threads(r.nW, r.nE, r.eOff);
    if (nthr) VG_GGL(rng_normals_kernel, dim3((nthr + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
    return (int)hipGetLastError();
}

static double adam_lr_t(double lr, int t) { return lr * sqrt(1.0 - pow(0.95, t)) / (1.0 - pow(0.8, t)); }

int vg_launch_adam(const vgpmp_dims* d, const vgpmp_params* x, const vgpmp_params* g, const vgpmp_params* am,
                   const vgpmp_params* av, int trainable, double lr, int t, hipStream_t st) {
    const size_t P = d->num_problems, L = d->L, M = d->M;
    const double lr_t = adam_lr_t(lr, t);
    AdamAllArgs aa;
    aa.nseg = 0; aa.lr_t = lr_t; aa.first_block[0] = 0;
    auto go = [&](size_t n, double* xx, const double* gg, double* mm, double* vv, int tril) {
        const int k = aa.nseg++;
        aa.n[k] = n; aa.x[k] = xx; aa.g[k] = gg; aa.m[k] = mm; aa.v[k] = vv; aa.tril_M[k] = tril;
        aa.first_block[k + 1] = aa.first_block[k] + (unsigned)((n + kBlock - 1) / kBlock);
    };
    if (trainable & VGPMP_TRAIN_Q_MU) go(P * L * M, x->q_mu, g->q_mu, am->q_mu, av->q_mu, 0);
    if (trainable & VGPMP_TRAIN_Q_SQRT) go(P * L * M * M, x->q_sqrt, g->q_sqrt, am->q_sqrt, av->q_sqrt, (int)M);
    if (trainable & VGPMP_TRAIN_LENGTHSCALES) go(P * L, x->raw_ell, g->raw_ell, am->raw_ell, av->raw_ell, 0);
    if (trainable & VGPMP_TRAIN_KERNEL_VARIANCE) go(P * L, x->raw_var, g->raw_var, am->raw_var, av->raw_var, 0);
    if (aa.nseg > 0 && aa.first_block[aa.nseg] > 0) VG_GGL(adam_all_kernel, dim3(aa.first_block[aa.nseg]), dim3(kBlock), 0, st, aa);
    return (int)hipGetLastError();
}

static int set_dyn_lds(const void* fn, size_t bytes) { return vg_grant_dyn_lds(fn, bytes); }

// `num_steps` consecutive steps.  Few problems (and not under the per-stage profiler): the role-dispatched
// stage launches above, with the variational-parameter update of step t riding in stage 1 of step t+1.
// Many problems: every kernel fills the chip by itself, plain launches in sequence.
int vg_elbo_steps(const vgpmp_dims* d, const vgpmp_robot* rb, const vgpmp_sdf* sdf, const vgpmp_problem* pb,
                  const vgpmp_params* params, const vgpmp_params* am, const vgpmp_params* av, const vgpmp_noise* nz,
                  const vgpmp_outputs* out, const vg_workspace* ws, int what, int trainable, double lr, int adam_t,
                  uint32_t seed, uint32_t problem_base, uint32_t step, int num_steps, hipStream_t st, hipEvent_t* ev) {
    const int P = d->num_problems, S = d->S, N = d->N, M = d->M, L = d->L, B = d->B, Mz = M + 2, J = N + Mz;
    const int SK = d->split_k, NC = vg_chunks(d), SC = vg_sc(d);
    const bool backward = (what & VGPMP_DO_BACKWARD) != 0, do_adam = (what & VGPMP_DO_ADAM) != 0;
    const bool gen = (what & VGPMP_GEN_NOISE) != 0;
    const bool want_dell = backward && (trainable & VGPMP_TRAIN_LENGTHSCALES);
    const bool fused = !ev && !(what & VGPMP_NO_FUSE) && SC == 8 && P * L <= vg_fuse_max_pl(S) && !pb->ind;
    const bool tiled_gemm = !fused && SK == 1 && (B % kTK) == 0;      // large batches: LDS-tiled kernel, no K-slices
    if (num_steps > 1 && !(backward && do_adam && gen)) return VGPMP_E_ARG;
    int evi = 0;
    auto mark = [&]() { if (ev) (void)hipEventRecord(ev[evi++], st); };
    uint32_t* ctr = pb->step_counter;
    int rc;
    // ---- argument blocks ----------------------------------------------------------------------
    CovArgs ca;
    ca.N = N; ca.M = M; ca.L = L; ca.D = L;
    const vgpmp_inducing_params* ind = pb->ind;      // inducing locations as variables: per-problem Zy, written from raw_Z
    const double* zy = ind ? ind->Zy : pb->Zy;
    const size_t zy_stride = ind ? (size_t)Mz * L : 0;
    ca.X = pb->X; ca.Zy = zy; ca.zy_stride = zy_stride; ca.y_u = pb->y_u; ca.jitter = pb->jitter;
    ca.q_mu = params->q_mu; ca.q_sqrt = params->q_sqrt; ca.raw_ell = params->raw_ell; ca.raw_var = params->raw_var;
    ca.want_dell = want_dell ? 1 : 0;
    ca.stop = -1;
    ca.elim_wave = (what & VGPMP_ELIM_BLOCK) ? 0 : 1;
    // batches (stage A and stage B are launches of their own, stage A off the critical path behind the generator roles): the
    // inverse once per latent in stage A (its two-panel form) instead of in every row-tile workgroup of stage B
    ca.ki_in_a = (!fused && ca.elim_wave && Mz > 16 && Mz <= 32 && N <= 256 && !(what & VGPMP_COV_LDS_ROWS)) ? 1 : 0;      // (N: cov_rows_tail's four passes)
    ca.rows_wave = ca.ki_in_a;      // ... which then goes on to the rows of A itself, one wave per 16 time points, in registers
    ca.tick = (fused && do_adam) ? ctr : nullptr;
    ca.lr = lr; ca.lr_dev = ws->lr_t;
    ca.rows_tpw = 1; ca.prologue = 0; ca.commit = 0; ca.keep_prev = (fused && backward) ? 1 : 0;
    ca.ws = *ws;
    FeatArgs fe;
    fe.N = N; fe.Mz = Mz; fe.L = L; fe.D = L; fe.B = B;
    // few problems: few points per workgroup (more parallelism); many: sweep 16 points per lane (omega reuse)
    // (shared launches: 8 for one problem -- the role is off the pole either way --, 16 from two: 105 -> 95 us per step)
    fe.jchunk = fused ? (P > 1 ? 16 : kJChunkOne) : (P * L >= 16 ? 16 : 4);
    fe.X = pb->X; fe.Zy = zy; fe.zy_stride = zy_stride; fe.raw_ell = params->raw_ell; fe.raw_var = params->raw_var;
    fe.omega = nz->omega; fe.beta = nz->beta; fe.Phi = ws->Phi; fe.dPhi = want_dell ? ws->dPhi : nullptr;
    fe.tick = (!fused && do_adam) ? ctr : nullptr;
    const dim3 feat_grid((B + kBlock - 1) / kBlock, (J + fe.jchunk - 1) / fe.jchunk, P * L);
    const size_t slab = (size_t)P * S * L * J;
    GemmArgs ga;
    ga.S = S; ga.L = L; ga.J = J; ga.B = B; ga.SK = SK; ga.nsel = want_dell ? 2 : 1;
    ga.W = nz->w; ga.Phi = ws->Phi; ga.dPhi = ws->dPhi; ga.F0 = ws->F0; ga.H = ws->H; ga.slab = slab;
    ga.dbg = 0; ga.w_exact = gen ? 1 : 0; ga.ell = ws->ell; ga.var = ws->var;
    TiledGemmArgs tga;
    tga.S = S; tga.L = L; tga.J = J; tga.B = B; tga.nsel = ga.nsel;
    tga.W = nz->w; tga.Phi = ws->Phi; tga.dPhi = ws->dPhi; tga.F0 = ws->F0; tga.H = ws->H;
    const dim3 gemm_grid((J + 16 * kNT - 1) / (16 * kNT), (S + 63) / 64, P * L * SK * ga.nsel);
    PathArgs pa;
    pa.S = S; pa.N = N; pa.Mz = Mz; pa.L = L; pa.SK = SK; pa.NC = NC; pa.slab = slab; pa.part_len = vg_part_len(d);
    pa.sqrt_jitter = (float)sqrt(pb->jitter);
    pa.A4 = reinterpret_cast<const float4*>(ws->A4); pa.AT = ws->AT;
    pa.C = ws->C; pa.CT = ws->CT; pa.nsplit = 1; pa.CT_ell = ws->CT_ell; pa.CT_var = ws->CT_var; pa.m = ws->m;
    pa.F0 = ws->F0; pa.H = ws->H; pa.want_dell = want_dell ? 1 : 0;
    pa.eps = nz->eps; pa.eps2 = nz->eps2; pa.epsT = nullptr; pa.eps2T = nullptr; pa.R = ws->R; pa.f = out->f; pa.G = ws->G; pa.part = ws->part;
    HyperArgs hy;
    hy.L = L; hy.Mz = Mz; hy.NC = NC; hy.want_dell = want_dell ? 1 : 0; hy.part_len = vg_part_len(d); hy.part = ws->part;
    hy.gkl_ell = ws->gkl_ell; hy.gkl_var = ws->gkl_var; hy.var = ws->var; hy.sig_ell = ws->sig_ell; hy.sig_var = ws->sig_var;
    hy.kl_scale = pb->kl_scale; hy.lr_t = 0.0;
    hy.g_ell = out->grad.raw_ell; hy.g_var = out->grad.raw_var; hy.lr_dev = ws->lr_t; hy.next = ws->theta_next;
    hy.m_ell = am ? am->raw_ell : nullptr; hy.m_var = am ? am->raw_var : nullptr;
    hy.v_ell = av ? av->raw_ell : nullptr; hy.v_var = av ? av->raw_var : nullptr;
    hy.p_ell = params->raw_ell; hy.p_var = params->raw_var;
    hy.do_adam = do_adam ? 1 : 0; hy.trainable = trainable; hy.use_lr_dev = (ctr && do_adam) ? 1 : 0;
    hy.ctr = fused ? nullptr : ctr; hy.lr = lr; hy.lr_store = ws->lr_t;      // one launch per kernel: hyper_kernel derives it
    HyperArgs hyp = hy;                  // prologue form: var / slopes of the PREVIOUS step (kept by stage 2)
    hyp.var = ws->prev_var; hyp.sig_ell = ws->prev_sig_ell; hyp.sig_var = ws->prev_sig_var;
    ca.hy = hyp; fe.hy = hyp;
    pa.stop = -1;
    pa.xcd_span = 0;
    pa.tick = nullptr;
    // paths_bwd_sc8: sample chunks per workgroup -- as many as leave six workgroups per CU.  (paths_bwd_sc8's values do not depend on it: a
    // set of sums per CHUNK, added in float64 by the assembly.  paths_bwd_regs leaves one set per WORKGROUP, its chunks added in float32: cpw
    // -- chosen from P L and NC -- then decides where float32 roundings fall, and a problem's gradient bits depend on the batch it rides in.)
    pa.cpw = 1;
    if (!(what & VGPMP_BWD_ONE_CHUNK)) {
        while (pa.cpw < NC && (size_t)P * L * ((NC + 2 * pa.cpw - 1) / (2 * pa.cpw)) >= kPbMinWgs) pa.cpw *= 2;
        // the register-resident path kernels (Mz = 32, one slab) work on PAIRS of chunks: two per workgroup as soon as that
        // leaves a workgroup per CU, four from ~600 workgroups (6 problems of config 2's shape: 155 -> 147 us per step, 13: 187 ->
        // 168, 24: 290 -> 245; the rule above alone left them on the one-chunk kernels below 28 problems)
        const bool pairs = SK == 1 && Mz == 32 && (N & 3) == 0 && N <= 100 && NC >= 2 && P * L > vg_fuse_max_pl(S);
        if (pairs && pa.cpw < 2 && (size_t)P * L * ((NC + 1) / 2) >= 256) pa.cpw = 2;
        if (pairs && pa.cpw == 2 && NC >= 4 && (size_t)P * L * ((NC + 3) / 4) >= 600) pa.cpw = 4;
    }
    // large batches: the register-resident reverse pass (paths_bwd_regs, below) leaves ONE set of partial sums per workgroup, not per chunk
    const bool regs_bwd = backward && SK == 1 && Mz == 32 && (N & 3) == 0 && N <= 100 && pa.cpw >= 2;
    const int NCp = regs_bwd ? (NC + pa.cpw - 1) / pa.cpw : NC;
    pa.NCp = NCp; hy.NC = NCp; hyp.NC = NCp; ca.hy.NC = NCp; fe.hy.NC = NCp;
    const double lik_scale = pb->alpha / (double)d->S_total;
    FinalArgs fa;
    fa.M = M; fa.L = L; fa.NC = NCp; fa.nblk = P ? vg_loglik_blocks_per_problem(S, N) : 0; fa.part_len = vg_part_len(d);
    fa.part = ws->part; fa.Lk32 = ws->Lk32; fa.lik_partial = ws->lik_partial;
    fa.gkl_qmu = ws->gkl_qmu; fa.gkl_Q = ws->gkl_Q; fa.kl_l = ws->kl_l;
    fa.kl_scale = pb->kl_scale; fa.lik_scale = lik_scale; fa.out_lik = out->lik; fa.out_kl = out->kl;
    const vgpmp_lik_params* lk = pb->lik;
    vg_lik_scratch lsc = {nullptr, nullptr, nullptr, nullptr};
    if (lk) vg_layout_lik_scratch(d, lk->scratch, &lsc);
    fa.alpha_fin = lk ? lsc.alpha_fin : nullptr;
    fa.g_qmu = out->grad.q_mu; fa.g_qsqrt = out->grad.q_sqrt;
    fa.do_adam = do_adam ? 1 : 0; fa.trainable = trainable; fa.lr_dev = ws->lr_t; fa.dma = 0;
    fa.lr_t = 0.0; fa.use_lr_dev = (ctr && do_adam) ? 1 : 0;
    fa.mq_mu = am ? am->q_mu : nullptr; fa.mq_sqrt = am ? am->q_sqrt : nullptr;
    fa.vq_mu = av ? av->q_mu : nullptr; fa.vq_sqrt = av ? av->q_sqrt : nullptr;
    fa.pq_mu = params->q_mu; fa.pq_sqrt = params->q_sqrt;
    fa.stop = -1;
    fa.tshift = 0;
    int skip1 = 0, skip2 = 0, skip3 = 0;
#ifdef VGPMP_BISECT
    ga.dbg = vg_bisect_stop("VGPMP_GEMM_DBG") > 0 ? vg_bisect_stop("VGPMP_GEMM_DBG") : 0;
    skip1 = vg_bisect_stop("VGPMP_S1_SKIP") > 0 ? vg_bisect_stop("VGPMP_S1_SKIP") : 0;
    skip2 = vg_bisect_stop("VGPMP_S2_SKIP") > 0 ? vg_bisect_stop("VGPMP_S2_SKIP") : 0;
    skip3 = vg_bisect_stop("VGPMP_S3_SKIP") > 0 ? vg_bisect_stop("VGPMP_S3_SKIP") : 0;
    ca.stop = vg_bisect_stop("VGPMP_STOP_COV");
    pa.stop = vg_bisect_stop("VGPMP_STOP_PATHS");
    fa.stop = vg_bisect_stop("VGPMP_STOP_FINAL");
#endif
    // ---- dynamic LDS sizes and kernel variants --------------------------------------------------
    const int Mp = (Mz + 15) & ~15;
    const size_t lds_cov = ((size_t)4 * Mp * (Mp + 2) + 8 * Mp) * sizeof(double);
    const size_t lds_cov_a = ((size_t)4 * Mp * (Mp + 1) + 2 * Mp) * sizeof(double);
    // row tiles per workgroup of stage B's rows role: one while the launch is small (latency), two from a few hundred
    // workgroups (a 16-row MFMA pass is full then; four make the rows workgroups the long ones of the launch: slower again)
    const int row_tiles = (N + kRowTile - 1) / kRowTile;
    int rows_tpw = 1;
    while (rows_tpw < kRowsTpwMax && (size_t)P * L * (kCovFixedRoles + (row_tiles + 2 * rows_tpw - 1) / (2 * rows_tpw)) >= kRowsTpwWgs) rows_tpw *= 2;
    ca.rows_tpw = rows_tpw;
    // (Mz = 32: cov_rows_body; any other Mz: cov_rows_padded_body, operands zero padded to Mp columns)
    const size_t lds_rows = Mz == 32
        ? ((size_t)3 * Mz * ((Mz + 2) & ~1) + (size_t)4 * kRowTile * Mz + Mz + rows_tpw * kRowTile) * sizeof(double)
        : ((size_t)3 * Mp * (Mp + 2) + (size_t)(Mp >= 32 ? 2 : 4) * 16 * Mp + Mp + rows_tpw * kRowTile) * sizeof(double);
    const size_t lds_cov_b = lds_cov > lds_rows ? lds_cov : lds_rows;
    // path kernels: operands + (when it fits) the raw split-K slabs of the prior draws
    const size_t raw_f = SK == 1 ? 0 : (size_t)SK * SC * J * sizeof(float);      // one slab lands in place
    size_t lds_pf = ((size_t)Mz * (Mz + 1) + (size_t)Mz * N + (size_t)3 * SC * Mz + Mz + (size_t)SC * J + 6 * 4) * sizeof(float);
    size_t lds_pb = ((size_t)3 * N * Mz + (size_t)2 * Mz * Mz + (size_t)SC * N + (size_t)2 * SC * J +
                     (size_t)8 * SC * Mz + 10 * 4) * sizeof(float);
    const bool raw_fwd = lds_pf + raw_f <= 64 * 1024, raw_bwd = lds_pb + 2 * raw_f <= 160 * 1024;
    if (raw_fwd) lds_pf += raw_f;
    if (raw_bwd) lds_pb += 2 * raw_f;
    size_t lds_fin = ((size_t)Mz * Mz + Mz + 1) * sizeof(double) + ((size_t)Mz * Mz + 4) * sizeof(float);
    const size_t raw_fin = (size_t)NCp * (Mz + Mz * Mz) * sizeof(float);
    const bool fin_dma = lds_fin + raw_fin <= 96 * 1024;      // every chunk's partials at once
    // otherwise in passes of as many chunks (a multiple of 8: the summation order goes by eights) as 96 KB hold
    const size_t row_fin = (size_t)(Mz + Mz * Mz) * sizeof(float);
    int fin_pass = fin_dma ? NCp : (int)(((96 * 1024 - lds_fin) / row_fin) & ~(size_t)7);
    // large batches: passes of eight chunks -- 46 instead of 80 KB of LDS, a third workgroup per CU; the same sums in the same order
    // (config-5 share: 51 -> 47 us for the launch)
    if (!fused && NCp > 8 && (size_t)P * L >= 512) fin_pass = 8;
    // ... and of four from 768 latent pairs (Mz <= 32: mid_hyper_final_kernel<32>, four workgroups per CU at 30 KB each): one round
    // (config-5 share: 41 -> 3x us for the launch; half sums carried between passes: the same additions in the same order)
    // ... and wherever the assembly shares the first launch of the next step with stage A and the draws (mid_stage1_kernel: every
    // role of a launch is granted the launch's LDS -- 80 KB of chunk partials per workgroup left two workgroups per CU for all of them)
    if (!fused && NCp > 4 && Mz <= 32 && (size_t)P * L > 128) fin_pass = 4;      // (up to 128 latent pairs: possibly the column-strip form)
    lds_fin += (size_t)fin_pass * row_fin;
    fa.dma = fin_pass;
    // few problems: the update role of stage 1 by column strips on kFinSplit workgroups (its LDS need is below lds_fin)
    const bool fin_split = fin_dma && Mz % (4 * kFinSplit) == 0 && (size_t)(M + M * (Mz / kFinSplit)) <= 2 * kBlock &&
                           !(what & VGPMP_NO_SPLIT);
    const size_t lds_s1 = lds_cov_a > lds_fin ? lds_cov_a : lds_fin;
    const void* fn_cov_b = ca.rows_wave ? (backward ? VG_FN(cov_b_kernel<true, false>) : VG_FN(cov_b_kernel<false, false>))
                                        : (backward ? VG_FN(cov_b_kernel<true, true>) : VG_FN(cov_b_kernel<false, true>));
    const bool k8 = (B / SK) % 128 == 0;      // K-slice in passes of 8 steps of 16: a pass's operands in one request
    // K-slices of a multiple of 128 and enough samples: operands through LDS by DMA (needs 59 KB per workgroup)
    const bool glds = (B / SK) % kGK == 0 && S >= 48 && !(what & VGPMP_GEMM_DIRECT);      // 64-row tiles: few samples waste them
    const bool gemm_first = SK == 1;      // whole-K tiles (3 problems up) are the longest workgroups of stage 2: at its front
#define VG_S2(T, K) (gemm_first ? VG_FN(stage2_gemm_first_kernel<T, K>) : VG_FN(stage2_kernel<T, K>))
    // many samples (the sample-sharded job on few ranks): the GEMM role is the step -- its products on the f16 matrix pipe
    const bool glds16 = glds && S >= kGemmF16MinSamples && !(what & VGPMP_PRIOR_F32);
    const void* fn_s2 = glds16 ? (backward ? VG_S2(true, -2) : VG_S2(false, -2))
                      : glds ? (backward ? VG_S2(true, -1) : VG_S2(false, -1))
                      : backward ? (k8 ? VG_S2(true, 8) : VG_S2(true, 0)) : (k8 ? VG_S2(false, 8) : VG_S2(false, 0));
#undef VG_S2
    size_t lds_s2 = glds && kGemmLds > lds_cov_b ? kGemmLds : lds_cov_b;
    if (SC != 8) return VGPMP_E_SHAPE;
#define VG_PICK(kernel, raw)                                                                                        \
    (SK == 1 ? (raw ? VG_FN(kernel<1, true>) : VG_FN(kernel<1, false>))                                 \
     : SK == 2 ? (raw ? VG_FN(kernel<2, true>) : VG_FN(kernel<2, false>))                               \
     : SK == 4 ? (raw ? VG_FN(kernel<4, true>) : VG_FN(kernel<4, false>))                               \
               : (raw ? VG_FN(kernel<8, true>) : VG_FN(kernel<8, false>)))
    const void* fn_pf = VG_PICK(paths_fwd_sc8, raw_fwd);
    const void* fn_s3 = VG_PICK(stage3_kernel, raw_fwd);
    const void* fn_pb = VG_PICK(paths_bwd_sc8, raw_bwd);
#undef VG_PICK
    // two workgroups per (chunk, latent) while the launch leaves half the chip idle
    const int Mh = Mz / 2, nxw = N - ((N / 2) & ~3);
    const size_t lds_pbs = ((size_t)4 * N * Mh + (size_t)2 * Mz * Mh + (size_t)SC * N + (size_t)2 * SC * Mh + (size_t)SC * Mz +
                            (size_t)2 * SC * nxw + (size_t)2 * SC * Mh + (size_t)2 * SK * SC * nxw + (size_t)2 * SK * SC * Mh +
                            (size_t)5 * SC * Mh + 12 * 4) * sizeof(float);
    const bool split_bwd = backward && SK > 1 && Mz % 8 == 0 && N % 4 == 0 && N >= 8 && lds_pbs <= 80 * 1024 &&
                           (size_t)P * L * NC * 2 <= 512 && !(what & VGPMP_NO_SPLIT);
    const size_t lds_pfs = ((size_t)Mz * Mz + (size_t)Mz * nxw + (size_t)4 * SC * Mz + Mz + (size_t)SC * nxw +
                            (size_t)SK * SC * nxw + (size_t)SK * SC * Mz + 10 * 4) * sizeof(float);
    const bool split_fwd = SK > 1 && Mz % 4 == 0 && N % 4 == 0 && N >= 8 && lds_pfs <= 64 * 1024 &&
                           (size_t)P * L * NC * 2 <= 512 && !(what & VGPMP_NO_SPLIT);
    if (split_fwd) { pa.nsplit = 2; lds_pf = lds_pfs; }
    // few problems, Mz = 32: no stage 3 -- the likelihood assembles its paths, the reverse pass carries the noise roles
    const bool lik_paths = fused && gen && backward && split_fwd && split_bwd && Mz == 32 && !lk && !pb->ind &&
                           !(what & (VGPMP_LIK_LANES | VGPMP_LIK_LDS_STATE | VGPMP_NO_SPLIT)) && vg_lik_paths_fit(L, SK) &&
                           (long long)P * S * N <= 28672;
    ca.form_u = lik_paths ? 1 : 0; ca.S = S; ca.eps = ws->epsT;
    const void* fn_s4 = SK == 2 ? VG_FN(stage4_kernel<2, 32>) : SK == 4 ? VG_FN(stage4_kernel<4, 32>) : VG_FN(stage4_kernel<8, 32>);
    if (split_bwd) {
        fn_pb = Mz == 32 ? (SK == 2 ? VG_FN(paths_bwd_split<2, 32>) : SK == 4 ? VG_FN(paths_bwd_split<4, 32>)
                                                                                        : VG_FN(paths_bwd_split<8, 32>))
                         : (SK == 2 ? VG_FN(paths_bwd_split<2, 0>) : SK == 4 ? VG_FN(paths_bwd_split<4, 0>)
                                                                                  : VG_FN(paths_bwd_split<8, 0>));
        lds_pb = lds_pbs;
    }
    // large batches: the latent's constants in registers, pairs of chunks through 40 KB of LDS (paths_bwd_regs)
    // (regs_bwd, above: SK == 1 excludes split_bwd)
    if (regs_bwd) {
        fn_pb = VG_FN(paths_bwd_regs<25>);
        lds_pb = ((size_t)kPbrBufs * (16 * N + (VG_PBR_DIRECT ? 0 : 32 * J) + 2 * 16 * Mz + 16) + (size_t)6 * 16 * Mz + (size_t)Mz * Mz + Mz + 8 * 4) * sizeof(float);
    }
    // ... and the forward assembly likewise (paths_fwd_regs); both take pa.cpw chunks per workgroup
    const bool regs_fwd = !fused && !split_fwd && SK == 1 && Mz == 32 && N <= 128 && pa.cpw >= 2;
    if (regs_fwd) {
        fn_pf = VG_FN(paths_fwd_regs<2>);
        lds_pf = ((size_t)16 * (3 * Mz + J) + 4 * 4) * sizeof(float);
    }
    if (backward && (rc = set_dyn_lds(fn_pb, lds_pb))) return rc;      // forward-only calls never launch the reverse pass
    if (lik_paths && (rc = set_dyn_lds(fn_s4, lds_pb))) return rc;
    if (fused) {
        if ((rc = set_dyn_lds(VG_FN(stage1_kernel<false>), lds_s1))) return rc;
        if ((rc = set_dyn_lds(VG_FN(stage1_kernel<true>), lds_s1))) return rc;
        if (lik_paths && lds_cov_b + (size_t)S * Mz * sizeof(float) > lds_s2) lds_s2 = lds_cov_b + (size_t)S * Mz * sizeof(float);
        if ((rc = set_dyn_lds(fn_s2, lds_s2))) return rc;
        if ((rc = set_dyn_lds(fn_s3, lds_pf))) return rc;
    } else {
        if ((rc = set_dyn_lds(VG_FN(cov_a_kernel), lds_cov_a))) return rc;
        if (glds && (rc = set_dyn_lds(VG_FN(prior_gemm_lds_kernel), kGemmLds))) return rc;
        if ((rc = set_dyn_lds(fn_cov_b, lds_cov_b))) return rc;
        if ((rc = set_dyn_lds(fn_pf, lds_pf))) return rc;
    }
    if ((rc = set_dyn_lds(VG_FN(final_kernel), lds_fin))) return rc;
    // medium batches: merged launches (not while profiling stage by stage, not with the shared stage launches)
    // measured on config 2 shapes: 5 problems 197 -> 182 us per step, 9: 218 -> 205, 16: equal, 24 and 64: 2-6 % slower
    // (the chip is full by then, the merged kernels only cost registers) -- hence the bound
    // few samples, four K-slices: features inside the GEMM (prior_fused_small_kernel)
    const bool fused_small = !fused && SK == 4 && S <= 32 && (B % 64) == 0 && !(what & VGPMP_GEMM_DIRECT);
    if (!fused && (rc = set_dyn_lds(VG_FN(mid_cov_a_rng_kernel), lds_cov_a))) return rc;      // (the large-batch schedule merges its small launches with these two as well)
    // (the register arrays of the gradient assembly at half size wherever Mz <= 32 -- every reference problem set -- and the assembly is
    //  not the column-strip form of small launches: 99 instead of 167 registers)
    const bool fin_mz32 = Mz <= 32 && !(fin_split && (size_t)L * P <= 128);
    const void* fn_mhf = fin_mz32 ? VG_FN(mid_hyper_final_kernel<32>) : VG_FN(mid_hyper_final_kernel<48>);
    if ((rc = set_dyn_lds(fn_mhf, lds_fin))) return rc;
    const dim3 cov_b_grid(kCovFixedRoles + (ca.rows_wave ? 0 : (row_tiles + rows_tpw - 1) / rows_tpw), L, P);      // (rows_wave: the rows of A are stage A's)
    // eps / eps' also as [P,L,S,Mz] wherever a consumer stages them per latent (the register-resident path kernels, stage B's U
    // role): drawn by rng_eps_t_body then, kEpsRows rows of (s, k) per workgroup
    // (only for noise drawn here: the caller's own eps -- generate = false -- come in the interface's layout alone)
    const bool eps_t = gen && (lik_paths || regs_fwd || regs_bwd);
    pa.epsT = eps_t && (regs_fwd || regs_bwd) ? ws->epsT : nullptr; pa.eps2T = eps_t && regs_fwd ? ws->eps2T : nullptr;
    float* const eps_t1 = eps_t ? ws->epsT : nullptr;
    float* const eps_t2 = eps_t && regs_fwd ? ws->eps2T : nullptr;
    const uint32_t eps_gx = eps_t ? rng_eps_t_blocks((uint32_t)S * Mz, 6)
                                  : (2u * rng_eps_quads((uint32_t)S * Mz * L, (uint32_t)d->sample_offset * Mz * L) + kBlock - 1) / kBlock;
    auto mid_normal_grid = [&](MidAArgs& ma) -> unsigned {      // the normal-draw roles of mid_cov_a_rng_kernel
        if (eps_t) { ma.rng.epsT = eps_t1; ma.rng.eps2T = eps_t2; ma.rng.nE = 0; }
        const uint32_t n_thr = rng_normal_threads(ma.rng.nW, ma.rng.nE, ma.rng.eOff);
        ma.n_gx = (int)((n_thr + kBlock - 1) / kBlock);
        ma.n_norm = ma.n_gx * P;
        // (256 rows per workgroup wherever that still leaves a workgroup per slot: 64-row workgroups are launch cost, 22 us of them
        //  at 896 latent pairs)
        if (eps_t && (size_t)P * rng_eps_t_blocks((uint32_t)S * Mz, 8) >= 1024) ma.rng.eps_rows_log2 = 8;
        ma.e_gx = eps_t ? (int)rng_eps_t_blocks((uint32_t)S * Mz, ma.rng.eps_rows_log2) : 0;
        return (unsigned)ma.n_norm + (unsigned)ma.e_gx * P;
    };
    const uint32_t basis_gx = ((uint32_t)L * B + kBlock - 1) / kBlock;
    const uint32_t w_gx = (((uint32_t)S * L * B >> 3) + kBlock - 1) / kBlock;      // a thread per counter of the W stream: eight normals
    auto launch = [&](const void* fn, dim3 grid, void* arg, size_t lds) -> int {
        void* kargs[] = {arg};
        vg_sched_note_fn(fn);
        return (int)hipLaunchKernel(fn, grid, dim3(kBlock), kargs, lds, st);
    };
    auto fused_small_args = [&]() {
        FusedPriorArgs fp;
        fp.S = S; fp.L = L; fp.J = J; fp.N = N; fp.D = L; fp.B = B; fp.want_dell = want_dell ? 1 : 0;
        fp.X = pb->X; fp.Zy = zy; fp.zy_stride = zy_stride; fp.raw_ell = params->raw_ell; fp.raw_var = params->raw_var;
        fp.omega = nz->omega; fp.beta = nz->beta; fp.W = nz->w; fp.F0 = ws->F0; fp.H = ws->H; fp.slab = slab;
        fp.tick = fe.tick;
        return fp;
    };
    const dim3 fgrid(P * L, (J + kFNT * 16 - 1) / (kFNT * 16));
    const bool small16 = !(what & VGPMP_PRIOR_F32) && (B / 4) % kHK == 0;      // the f16-split form of the few-sample prior kernel
    // stage B and the few-sample prior draws as ONE launch (mid_cov_b_prior16_kernel): the batch schedule drawing its own noise
    auto launch_cov_b_prior16 = [&](const CovArgs& cb) -> int {
        MidBArgs mb;
        mb.cov = cb; mb.fp = fused_small_args();
        mb.latents = L * P; mb.n_cov = kCovFixedRoles * L * P; mb.prior_gx = (int)fgrid.x;
        const dim3 grid(mb.n_cov + fgrid.x * fgrid.y);
        int rc = 0;
#define VG_CBP(MT_, DM_)                                                                                                          \
    do {                                                                                                                          \
        const void* fn_ = want_dell ? VG_FN(mid_cov_b_prior16_kernel<MT_, DM_, true>) : VG_FN(mid_cov_b_prior16_kernel<MT_, DM_, false>); \
        if (!(rc = set_dyn_lds(fn_, lds_cov_b))) rc = launch(fn_, grid, &mb, lds_cov_b);                                          \
    } while (0)
        if (L == 7) { if (S <= 16) VG_CBP(1, 7); else VG_CBP(2, 7); }
        else if (L == 6) { if (S <= 16) VG_CBP(1, 6); else VG_CBP(2, 6); }
        else if (L <= 8) { if (S <= 16) VG_CBP(1, 8); else VG_CBP(2, 8); }
        else { if (S <= 16) VG_CBP(1, 16); else VG_CBP(2, 16); }
#undef VG_CBP
        return rc;
    };
    auto launch_fused_small = [&](hipEvent_t g0, hipEvent_t g1) {
        const FusedPriorArgs fp = fused_small_args();
#define VG_FUSED_SMALL(MT_, DM_)                                            \
    (want_dell ? VG_EXT_GGL((prior_fused_small_kernel<MT_, DM_, true>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp)                                  \
               : VG_EXT_GGL((prior_fused_small_kernel<MT_, DM_, false>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp))
        // the f16-split form (gp_prior_split.h): K steps of 32 bases (every K-slice a multiple of that); the float32-MFMA kernel
        // stays behind VGPMP_PRIOR_F32, as for the large batches
        if (small16) {
#define VG_FUSED_SMALL16_(MT_, DM_, WX_)                                        \
    (want_dell ? VG_EXT_GGL((prior_fused_small16_kernel<MT_, DM_, true, WX_>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp)                                  \
               : VG_EXT_GGL((prior_fused_small16_kernel<MT_, DM_, false, WX_>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp))
            // (weights drawn by the library are float16 values: no low half -- the WX form; a caller's own weights take the split)
#define VG_FUSED_SMALL16(MT_, DM_) (gen ? VG_FUSED_SMALL16_(MT_, DM_, true) : VG_FUSED_SMALL16_(MT_, DM_, false))
            if (L == 7) { if (S <= 16) VG_FUSED_SMALL16(1, 7); else VG_FUSED_SMALL16(2, 7); }
            else if (L == 6) { if (S <= 16) VG_FUSED_SMALL16(1, 6); else VG_FUSED_SMALL16(2, 6); }
            else if (L <= 8) { if (S <= 16) VG_FUSED_SMALL16(1, 8); else VG_FUSED_SMALL16(2, 8); }
            else { if (S <= 16) VG_FUSED_SMALL16(1, 16); else VG_FUSED_SMALL16(2, 16); }
#undef VG_FUSED_SMALL16_
#undef VG_FUSED_SMALL16
            return;
        }
        // (6- and 7-joint arms: no padded column -- the compiler cannot drop a multiply by a zero it must assume could meet a NaN)
        if (L == 7) { if (S <= 16) VG_FUSED_SMALL(1, 7); else VG_FUSED_SMALL(2, 7); }
        else if (L == 6) { if (S <= 16) VG_FUSED_SMALL(1, 6); else VG_FUSED_SMALL(2, 6); }
        else if (L <= 8) { if (S <= 16) VG_FUSED_SMALL(1, 8); else VG_FUSED_SMALL(2, 8); }
        else { if (S <= 16) VG_FUSED_SMALL(1, 16); else VG_FUSED_SMALL(2, 16); }
#undef VG_FUSED_SMALL
    };
    // (stand-alone / merged launches: strips while the launch leaves half the chip idle -- 8 problems: 177 -> 172 us per
    // step; at 64 problems four times the workgroups each re-staging the factor LOSE 15 us)
    const bool fin_split_batch = fin_split && (size_t)L * P <= 128;
    fa.split = fin_split_batch ? 1 : 0;
    bool batch_merged = false;      // this call's steps run the large-batch schedule with its small launches merged
    auto launch_final = [&]() -> int { return launch(VG_FN(final_kernel), dim3(L * (fin_split_batch ? kFinSplit : 1), P), &fa, lds_fin); };
    // likelihood constants as variables (vgpmp_lik_params): effective values from the raw ones at the start of the call
    LikUpdArgs lu;
    if (lk) {
        LikConstArgs lc;
        lc.raw_alpha = lk->raw_alpha; lc.raw_sigma = lk->raw_sigma; lc.sc = lsc; lc.inv_s = 1.0 / (double)d->S_total;
        VG_GGL(lik_consts_kernel, dim3(P), dim3(VGPMP_MAX_SPHERES), 0, st, lc);
        lu.rb = rb; lu.lik_partial = ws->lik_partial; lu.sig_partial = lsc.sig_partial; lu.nblk = 0;
        lu.inv_s = lc.inv_s;
        lu.raw_alpha = lk->raw_alpha; lu.raw_sigma = lk->raw_sigma; lu.m_alpha = lk->m_alpha; lu.v_alpha = lk->v_alpha;
        lu.m_sigma = lk->m_sigma; lu.v_sigma = lk->v_sigma; lu.g_alpha = lk->g_alpha; lu.g_sigma = lk->g_sigma;
        lu.sc = lsc; lu.do_adam = do_adam ? 1 : 0; lu.trainable = trainable;
        lu.ctr = do_adam ? ctr : nullptr; lu.lr = lr; lu.lr_t = 0.0;
    }

    vg_lik_paths lpa;
    lpa.SK = SK; lpa.slab = slab; lpa.sqrt_jitter = pa.sqrt_jitter; lpa.AT = ws->AT; lpa.F0 = ws->F0; lpa.U = ws->U;
    lpa.eps2 = nz->eps2; lpa.R = ws->R; lpa.f = out->f;
    for (int i = 0; i < num_steps; ++i) {
        const bool first = i == 0, more = i + 1 < num_steps;
        vg_sched_clear();      // (include/vgpmp_debug.h: the log holds the launches of the call's last step)
        bool draw_next = false;      // (lik_paths) the reverse path launch also draws the next step's omega, beta, w
        const uint32_t step_i = step + (uint32_t)i;
        if (ind && (rc = vg_launch_inducing_build(d, ind, st))) return rc;       // Zy = [0; 1; Z(raw_Z)] of every problem
        if (fused) {
            // noise of the first step of a call: everything up front; afterwards eps rides in stage 1 and the
            // prior noise of step i was drawn by stage 3 of step i-1
            // (VGPMP_NOISE_READY: a previous call's stage 3 drew this step's omega, beta, w already -- only eps is missing and
            //  rides in stage 1; VGPMP_NOISE_AHEAD: stage 3 of this call's last step draws the next call's.  The sample-sharded
            //  step is one call per step with an all-reduce in between: without these it paid two noise launches per step.)
            const bool ready = !first || (what & VGPMP_NOISE_READY);
            const bool ahead = more || (what & VGPMP_NOISE_AHEAD);
            if (gen && !ready && (rc = vg_launch_rng(d, nz, seed, problem_base, step_i, ctr, st, eps_t1, eps_t2))) return rc;
            // steps after the first of a call: the hyper-parameter update of the previous step is a prologue of the
            // cov_a and feature roles, its q_mu / q_sqrt update (final) another role of the same launch
            const bool prologue = !first && backward;
            const double lr_prev = do_adam ? adam_lr_t(lr, adam_t + i - 1 > 0 ? adam_t + i - 1 : 1) : 0.0;
            hyp.lr_t = lr_prev;
            Stage1Args s1;
            s1.skip = skip1;
            s1.cov = ca; s1.fin = fa; s1.feat = fe;
            s1.fin.lr_t = lr_prev;
            s1.fin.tshift = 40;
            s1.cov.hy = hyp; s1.feat.hy = hyp;
            s1.cov.prologue = prologue ? 1 : 0;
            s1.rng = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
            s1.rng.epsT = eps_t1; s1.rng.eps2T = eps_t2;
            s1.n_cov = L * P;
            s1.fin_split = fin_split ? 1 : 0;
            s1.n_fin = first ? 0 : L * P * (fin_split ? kFinSplit : 1);
            s1.eps_gx = (int)eps_gx;
            s1.n_eps = (gen && ready) ? (int)eps_gx * P : 0;
            s1.feat_gx = (int)feat_grid.x; s1.feat_gy = (int)feat_grid.y;
            s1.n_feat = (int)(feat_grid.x * feat_grid.y * feat_grid.z);
            const unsigned n1 = s1.n_cov + s1.n_fin + s1.n_eps + feat_grid.x * feat_grid.y * feat_grid.z;
            if ((rc = launch(prologue ? VG_FN(stage1_kernel<true>) : VG_FN(stage1_kernel<false>), dim3(n1), &s1, lds_s1)))
                return rc;
            Stage2Args s2;
            s2.skip = skip2;
            s2.cov = ca; s2.gemm = ga;
            s2.cov.hy = hyp; s2.cov.commit = prologue ? 1 : 0;
            s2.cov_roles = (int)cov_b_grid.x; s2.n_cov = (int)(cov_b_grid.x * cov_b_grid.y * cov_b_grid.z);
            s2.gemm_gx = (int)gemm_grid.x; s2.gemm_gy = (int)gemm_grid.y;
            const int n_gemm = (int)(gemm_grid.x * gemm_grid.y * gemm_grid.z);
            s2.n_gemm = n_gemm;
            // (tiles whose index agrees mod 8 share an XCD whatever the number of covariance workgroups in front of them)
            s2.gemm_per_xcd = n_gemm % 8 == 0 ? n_gemm / 8 : 0;
            if ((rc = launch(fn_s2, dim3(s2.n_cov + gemm_grid.x * gemm_grid.y * gemm_grid.z), &s2, lds_s2))) return rc;
            if (!lik_paths) {
                Stage3Args s3;
                s3.skip = skip3;
                s3.path = pa;
                // the counter has ticked in stage 2: it already names the next step
                s3.rng = make_rng_args(d, nz, seed, problem_base, step_i + 1u, ctr, 0u);
                s3.n_path = NC * pa.nsplit * L * P;
                s3.path.xcd_span = s3.n_path % 8 == 0 ? s3.n_path / 8 : 0;
                s3.basis_gx = (int)basis_gx; s3.w_gx = (int)w_gx;
                s3.n_basis = (gen && ahead) ? (int)basis_gx * P : 0;
                const unsigned n3 = s3.n_path + s3.n_basis + ((gen && ahead) ? w_gx * P : 0u);
                if ((rc = launch(fn_s3, dim3(n3), &s3, lds_pf))) return rc;
            }
            draw_next = gen && ahead;
        } else {
            mark();
            // large batches drawing their own noise: W and the features are formed inside the GEMM (prior_fused_batch_kernel);
            // trainable inducing locations read W back (inducing.hip), so they keep it in memory
            const bool fbatch = gen && tiled_gemm && !fused_small && !ind && !(what & VGPMP_NO_FUSE_PRIOR);
            // the small launches of a step share launches here too (not while profiling stage by stage): stage A of the
            // covariance path beside the noise draws, the two updates at the end in one, the counter tick inside paths_fwd
            const bool batch_merge = gen && backward && !ev && !ind && !lk && !(what & (VGPMP_COV_ONLY | VGPMP_NO_FUSE));
            if (batch_merge) {
                MidAArgs ma;
                ma.cov = ca;
                ma.rng = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
                if (fbatch) ma.rng.nW = 0;               // omega, beta, eps, eps2 only
                ma.n_cov = L * P; ma.basis_gx = (int)((ma.rng.L * ma.rng.B + kBlock - 1) / kBlock); ma.n_basis = ma.basis_gx * P;
                const unsigned n_draw = mid_normal_grid(ma);
                if (!first && do_adam) {
                    // the updates of step i - 1 (left out at its end, below) beside stage A and the draws of this step
                    MidS1Args s1;
                    s1.a = ma;
                    s1.a.cov.prologue = 2;
                    hy.lr_t = adam_lr_t(lr, adam_t + i - 1 > 0 ? adam_t + i - 1 : 1);
                    fa.lr_t = hy.lr_t;
                    s1.a.cov.hy = hy; s1.hy = hy; s1.fin = fa;
                    s1.n_fin = L * P * (fin_split_batch ? kFinSplit : 1);
                    const size_t lds_s1b = lds_cov_a > lds_fin ? lds_cov_a : lds_fin;
                    const void* fn_ms1 = fin_mz32 ? VG_FN(mid_stage1_kernel<32>) : VG_FN(mid_stage1_kernel<48>);
                    if ((rc = set_dyn_lds(fn_ms1, lds_s1b))) return rc;
                    if ((rc = launch(fn_ms1, dim3(s1.n_fin + ma.n_cov + ma.n_basis + n_draw), &s1, lds_s1b))) return rc;
                } else if ((rc = launch(VG_FN(mid_cov_a_rng_kernel), dim3(ma.n_cov + ma.n_basis + n_draw), &ma, lds_cov_a))) return rc;
            } else {
                VG_GGL(cov_a_kernel, dim3(L, P), dim3(kCovThreads), lds_cov_a, st, ca);
            }
            // Stage B and the prior draws are independent of each other (both need only stage A | noise): they share a launch wherever the
            // schedule draws its own noise.  (Rounds 4-5 ran stage B on a second stream of the caller's beside the prior kernel for 64-512
            // latent pairs -- 0.192 / 0.199 ms per step at 16 Franka problems then; since the rows of A left stage B the second queue
            // measures no gain at 16 / 32 / 64 problems and +4 % at 896 latent pairs: vgpmp_problem.aux_stream is ignored.)
            const bool b_prior16 = batch_merge && fused_small && small16 && ca.rows_wave && backward;      // (stage B here has no rows role)
            // ... and with many samples behind the tiles of the f16-split prior kernel (prior_split_cov_b_kernel)
            const bool b_split = batch_merge && fbatch && !(what & VGPMP_PRIOR_F32) && ca.rows_wave && backward &&
                                 2 * lds_cov_b <= 80 * 1024 && VG_B_SPLIT;
            // ... and the path assembly as those tiles' epilogue (one column tile per latent: J <= 144; the register-resident assembly's shapes)
            const bool fuse_fwd = b_split && regs_fwd && regs_bwd && eps_t && J <= kTJ && VG_FUSE_FWD;      // (regs_bwd: it carries the tick)
            if (b_split) {
                // (launched with the prior tiles, below)
            } else if (b_prior16) {
                if ((rc = launch_cov_b_prior16(ca))) return rc;      // ... and the few-sample prior draws with it
            } else if ((rc = launch(fn_cov_b, cov_b_grid, &ca, lds_cov_b))) return rc;
            if (what & VGPMP_COV_ONLY) return (int)hipGetLastError();     // Kuu, Cholesky, q_sqrt, A, per-latent KL: done
            mark();
            if (gen && !batch_merge) {
                RngArgs r = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
                if (fbatch) r.nW = 0;                    // omega, beta, eps, eps2 only
                VG_GGL(rng_basis_kernel, dim3((r.L * r.B + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
                if (eps_t) {
                    r.epsT = ws->epsT; r.eps2T = ws->eps2T;
                    VG_GGL(rng_eps_t_kernel, dim3(rng_eps_t_blocks((uint32_t)S * Mz, r.eps_rows_log2), P), dim3(kBlock), 0, st, r);
                    r.nE = 0;
                }
                const uint32_t nthr = rng_normal_threads(r.nW, r.nE, r.eOff);
                if (nthr) VG_GGL(rng_normals_kernel, dim3((nthr + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
            }
            mark();
            if (!fused_small && !fbatch) VG_GGL(features_kernel, feat_grid, dim3(kBlock), 0, st, fe);
            mark();
            hipEvent_t g0 = ev ? ev[VG_NUM_STAGES + 3] : nullptr, g1 = ev ? ev[VG_NUM_STAGES + 4] : nullptr;
            if (fbatch) {
                FusedBatchArgs fb;
                fb.S = S; fb.L = L; fb.J = J; fb.N = N; fb.D = L; fb.B = B; fb.want_dell = want_dell ? 1 : 0;
                fb.X = pb->X; fb.Zy = zy; fb.zy_stride = zy_stride; fb.raw_ell = params->raw_ell; fb.raw_var = params->raw_var;
                fb.omega = nz->omega; fb.beta = nz->beta; fb.F0 = ws->F0; fb.H = ws->H;
                fb.seed = seed; fb.problem_base = problem_base; fb.step = step_i; fb.ctr = ctr;
                fb.wOff = (uint32_t)d->sample_offset * L * B;
                const int dm = L <= 8 ? 8 : 16;
                // 128-row tiles (one workgroup per CU, the features of a K step shared by twice the rows): 64 Franka problems
                // 883 -> 862 us per step.  Their workgroups run twice as long, so a thin last round costs a whole one: only
                // when the last round of P L workgroups over the 256 CUs is at least 70 % full (40 problems: +10 % otherwise);
                // the 16-wide joint padding (9-16 joints) measured 1.4 % slower with them (config 5), so 8-wide only.
                const int fb_tail = (int)(((size_t)P * L) % 256);
                const int fmt = S > kTS && dm == 8 && (size_t)P * L >= 256 &&
                                (fb_tail == 0 || fb_tail >= 180) ? 2 : 1;
                const size_t lds_fb = ((size_t)kTS * fmt * kFBLd + (size_t)2 * kTJ * kFBLd + (size_t)kTJ * dm + (size_t)2 * kFBK * (dm + 4)) * sizeof(float);
                const dim3 fb_grid((J + kTJ - 1) / kTJ, (S + kTS * fmt - 1) / (kTS * fmt), P * L);
                // the f16-split form (gp_prior_split.h): 512-thread workgroups, 128-row tiles whenever there are more than 64
                // samples, two workgroups per CU; the float32-MFMA kernel stays behind VGPMP_PRIOR_F32 (the tests' reference form)
                const bool split16 = !(what & VGPMP_PRIOR_F32);
                if (split16) {
                    // (128-row tiles halve the feature work per sample, but below ~one tile per CU their workgroups run alone:
                    //  64-row tiles then, twice the workgroups at ~0.6 of the duration)
                    const size_t h_tiles2 = (size_t)((J + kTJ - 1) / kTJ) * ((S + 2 * kTS - 1) / (2 * kTS)) * P * L;
                    const int hmt = S > kTS && h_tiles2 > kHMt2MinTiles ? 2 : 1;
                    const size_t lds_h = vg_fused_split_lds(hmt);
                    const dim3 hgrid((J + kTJ - 1) / kTJ, (S + kTS * hmt - 1) / (kTS * hmt), P * L);
#define VG_FH(DELL_, MT_)                                                                                                 \
    do {                                                                                                                  \
        if ((rc = set_dyn_lds(VG_FN(prior_fused_split_kernel<DELL_, MT_>), lds_h))) return rc;                      \
        VG_EXT_GGL((prior_fused_split_kernel<DELL_, MT_>), hgrid, dim3(kHThreads), lds_h, st, g0, g1, 0, fb);   \
    } while (0)
#define VG_FHC(DELL_, MT_)                                                                                                \
    do {                                                                                                                  \
        const void* fn_ = fuse_fwd ? VG_FN(prior_split_cov_b_kernel<DELL_, MT_, true>) : VG_FN(prior_split_cov_b_kernel<DELL_, MT_, false>); \
        if ((rc = set_dyn_lds(fn_, lds_hc))) return rc;                                                                   \
        void* kargs_[] = {&pc};                                                                                           \
        vg_sched_note_fn(fn_);                                                                                            \
        const unsigned per_z_ = hgrid.x * hgrid.y;                                                                        \
        VG_CHECK_HIP(hipLaunchKernel(fn_, dim3(hgrid.x, hgrid.y, hgrid.z + (kCovFixedRoles * pc.pairs + per_z_ - 1) / per_z_), dim3(kHThreads), kargs_, lds_hc, st)); \
    } while (0)
                    if (b_split) {      // stage B rides behind the prior tiles (prior_split_cov_b_kernel)
                        PriorCovArgs pc;
                        pc.fb = fb; pc.cov = ca; pc.nz_prior = (int)hgrid.z; pc.latents = L * P; pc.pairs = (L * P + 1) / 2;
                        pc.lds_half = (unsigned)((lds_cov_b + 15) & ~(size_t)15);
                        if (fuse_fwd) {
                            pc.fw.Lk64 = ws->Lk64; pc.fw.q_sqrt = params->q_sqrt; pc.fw.q_mu = params->q_mu; pc.fw.y_u = pb->y_u;
                            pc.fw.jitter = pb->jitter; pc.fw.AT = ws->AT; pc.fw.epsT = ws->epsT; pc.fw.eps2T = ws->eps2T;
                            pc.fw.sqrt_jitter = pa.sqrt_jitter; pc.fw.f = out->f; pc.fw.R = ws->R;
                        }
                        const size_t lds_hc = lds_h > 2 * (size_t)pc.lds_half ? lds_h : 2 * (size_t)pc.lds_half;
                        if (want_dell) { if (hmt == 2) VG_FHC(true, 2); else VG_FHC(true, 1); }
                        else { if (hmt == 2) VG_FHC(false, 2); else VG_FHC(false, 1); }
                    } else if (want_dell) { if (hmt == 2) VG_FH(true, 2); else VG_FH(true, 1); }
                    else { if (hmt == 2) VG_FH(false, 2); else VG_FH(false, 1); }
#undef VG_FHC
#undef VG_FH
                } else {
#define VG_FB(DELL_, DM_)                                                                                                 \
    do {                                                                                                                  \
        if (fmt == 2) VG_EXT_GGL((prior_fused_batch_kernel<DELL_, DM_, 2>), fb_grid, dim3(kBlock), lds_fb, st, g0, g1, 0, fb); \
        else VG_EXT_GGL((prior_fused_batch_kernel<DELL_, DM_, 1>), fb_grid, dim3(kBlock), lds_fb, st, g0, g1, 0, fb);          \
    } while (0)
                    if (want_dell) { if (dm == 8) VG_FB(true, 8); else VG_FB(true, 16); }
                    else { if (dm == 8) VG_FB(false, 8); else VG_FB(false, 16); }
#undef VG_FB
                }
                if (fe.tick && batch_merge) pa.tick = fe.tick;      // the feature kernel's tick: by paths_fwd (next launch), or alone
                else if (fe.tick) VG_GGL(tick_kernel, dim3(1), dim3(1), 0, st, fe.tick);
            } else if (fused_small) {      // features formed inside the GEMM (few samples: Phi / dPhi traffic is the cost)
                if (!b_prior16) launch_fused_small(g0, g1);      // (b_prior16: they went with stage B)
            } else if (tiled_gemm) {
                const int mt = 1;      // 128-sample tiles (mt = 2) measured slower: 90 vs 98 TF/s at 64 problems
                const size_t lds_tg = (size_t)2 * (kTS * mt + kTJ) * kTLd * sizeof(float);
                const void* fn_tg = mt == 2 ? VG_FN(prior_gemm_tiled_kernel<2>) : VG_FN(prior_gemm_tiled_kernel<1>);
                if ((rc = set_dyn_lds(fn_tg, lds_tg))) return rc;
                const dim3 tg_grid((J + kTJ - 1) / kTJ, (S + kTS * mt - 1) / (kTS * mt), P * L * ga.nsel);
                if (mt == 2) VG_EXT_GGL(prior_gemm_tiled_kernel<2>, tg_grid, dim3(kBlock), lds_tg, st, g0, g1, 0, tga);
                else VG_EXT_GGL(prior_gemm_tiled_kernel<1>, tg_grid, dim3(kBlock), lds_tg, st, g0, g1, 0, tga);
            } else if (glds)
                VG_EXT_GGL(prior_gemm_lds_kernel, gemm_grid, dim3(kBlock), kGemmLds, st, g0, g1, 0, ga);
            else
                VG_EXT_GGL(prior_gemm_kernel<0>, gemm_grid, dim3(kBlock), 0, st, g0, g1, 0, ga);
            mark();
            if (fuse_fwd) {
                // (f and R came with the prior tiles; the counter's tick rides with the reverse pass: pa.tick stays set until then)
            } else {
                if ((rc = launch(fn_pf, dim3(regs_fwd ? (NC + pa.cpw - 1) / pa.cpw : NC * pa.nsplit, L, P), &pa, lds_pf))) return rc;
                pa.tick = nullptr;
            }
            mark();
            batch_merged = batch_merge;
        }
        // ---- likelihood forward + reverse (fk_sdf.hip)
        int nblk = 0;
        rc = vg_launch_loglik_paths(rb, sdf, out->f, P, S, L, N, (float)(-lik_scale), ws->G, out->logp, ws->lik_partial,
                                    &nblk, st, ev ? ev[VG_NUM_STAGES + 1] : nullptr, ev ? ev[VG_NUM_STAGES + 2] : nullptr,
                                    lk ? lsc.alpha_eff : nullptr, lk ? lsc.sigma_eff : nullptr,
                                    lk ? lsc.sig_partial : nullptr, (what & VGPMP_LIK_LDS_STATE) ? 2 : (what & VGPMP_LIK_LANES) ? 1 : 0,
                                    lik_paths ? &lpa : nullptr);
        if (rc) return rc;
        fa.nblk = nblk;
        mark();
        if (!backward) {
            VG_GGL(elbo_pieces_kernel, dim3(P), dim3(kBlock), 0, st, L, nblk, ws->lik_partial, ws->kl_l, lik_scale,
                               pb->kl_scale, out->lik, out->kl, fa.alpha_fin);
            return (int)hipGetLastError();
        }
        // ---- reverse of the path assembly (+ hyper-parameter update), then (here or in the next stage 1) the rest
        pa.xcd_span = split_bwd && (2 * NC * L * P) % 8 == 0 ? 2 * NC * L * P / 8 : 0;
        if (lik_paths) {
            Stage4Args s4;
            s4.path = pa;
            s4.rng = make_rng_args(d, nz, seed, problem_base, step_i + 1u, ctr, 0u);      // (the counter has ticked in stage 2)
            s4.n_bwd = 2 * NC * L * P; s4.bwd_gx = 2 * NC;
            s4.basis_gx = (int)basis_gx; s4.w_gx = (int)w_gx;
            s4.n_basis = draw_next ? (int)basis_gx * P : 0;
            const unsigned n4 = (unsigned)s4.n_bwd + (unsigned)s4.n_basis + (draw_next ? w_gx * P : 0u);
            if ((rc = launch(fn_s4, dim3(n4), &s4, lds_pb))) return rc;
        } else if ((rc = launch(fn_pb, dim3(split_bwd ? 2 * NC : (NC + pa.cpw - 1) / pa.cpw, L, P), &pa, lds_pb))) return rc;
        pa.tick = nullptr;
        if (ind) {     // inducing locations as variables: reverse through the covariance path and the prior draw at Zy
            vg_ind_launch il;
            il.d = d; il.ind = ind; il.ws = ws; il.nz = nz; il.params = params; il.X = pb->X; il.y_u = pb->y_u;
            il.jitter = pb->jitter; il.do_adam = do_adam ? 1 : 0; il.trainable = trainable;
            il.ctr = do_adam ? ctr : nullptr; il.lr = lr;
            il.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            if ((rc = vg_launch_inducing_backward(il, st))) return rc;
        }
        if (lk) {      // trainable likelihood constants: their gradient / update, and the constants of the next step
            lu.nblk = nblk;
            lu.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            VG_GGL(lik_update_kernel, dim3(P), dim3(kLikUpdWaves * VGPMP_MAX_SPHERES), 0, st, lu);
        }
        mark();
        if (batch_merged && more && do_adam) {
            // (more steps to come in this call: both updates ride in the first launch of the next step, mid_stage1_kernel)
        } else if (batch_merged || (fused && !more)) {      // (fused, more steps to come: both ride in stage 1 of the next step)
            MidGArgs mg;
            hy.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            fa.lr_t = hy.lr_t;
            mg.hy = hy; mg.fin = fa; mg.n_hyper = P;
            if ((rc = launch(fn_mhf, dim3(P + L * P * (fin_split_batch ? kFinSplit : 1)), &mg, lds_fin))) return rc;
        } else if (!fused) {
            hy.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            fa.lr_t = hy.lr_t;
            VG_GGL(hyper_kernel, dim3(P), dim3(64), 0, st, hy);
            if ((rc = launch_final())) return rc;
        }
        mark();
    }
    return (int)hipGetLastError();
}
