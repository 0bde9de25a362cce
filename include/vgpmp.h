/*
 * vgpmp.h -- C ABI of the MI355X-native vGPMP hot path (libvgpmp_hip.so).
 *
 * The reference (luke-ck/vgpmp) is pure Python on TensorFlow/GPflow and has no FFI; the entry
 * points below are what a binding for its ELBO inner loop would call.  Each one names the
 * reference code it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer marked `dev` is DEVICE memory owned by the caller (PyTorch-ROCm tensors in
 *     the Python host); the library never allocates or frees caller memory; scratch comes from
 *     the caller-provided workspace (vgpmp_workspace_bytes).
 *   - return value: 0 = ok, negative = argument/shape error (VGPMP_E_*), positive = hipError_t.
 *   - thread-compatible; every launch goes to the explicit `stream`; no hidden global state.
 *   - arrays are dense, C order, leading dimension = problem index for batched buffers.
 *
 * Limits (VGPMP_E_SHAPE beyond them): dof <= 16, spheres <= 64, Mz = M + 2 <= 48, B a multiple of 16, N <= 4096 and,
 * because one latent's A = Kfu (Kuu + jI)^-1 ([N, Mz] float32) is LDS-resident in the path kernels,
 *   forward  (VGPMP_DO_FORWARD):   4 * (Mz (Mz + 1) + Mz N + 33 Mz + 8 N + 24)  <= 160 KiB   (N <= 970 at M = 30)
 *   backward (VGPMP_DO_BACKWARD):  4 * (4 N Mz + 2 Mz^2 + 24 N + 80 Mz + 40)   <= 160 KiB   (N <= 238 at M = 30, 292 at M = 24)
 * vgpmp_workspace_bytes checks the forward bound (it does not know `what`); a step with VGPMP_DO_BACKWARD checks its own.
 */
#ifndef VGPMP_H
#define VGPMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGPMP_MAX_DOF 16
#define VGPMP_MAX_FRAMES 17
#define VGPMP_MAX_SPHERES 64
#define VGPMP_MAX_MZ 48          /* inducing points + 2 conditioned end points (LDS-resident float64 algebra) */

#define VGPMP_E_ARG (-1)         /* null pointer / inconsistent argument */
#define VGPMP_E_SHAPE (-2)       /* dimension outside the supported range */
#define VGPMP_E_WORKSPACE (-3)   /* workspace too small */
#define VGPMP_E_COMM (-4)        /* RCCL not loadable, or a collective call failed */

typedef void* vgpmp_stream;      /* hipStream_t */

/* Robot + likelihood constants: what utils/sampler.py:28-56 and likelihoods/likelihood.py:22-55
 * keep (DH table, twist, convention, base pose, frame of each sphere, sphere offsets and radii,
 * sigma_obs per sphere, epsilon, joint limits, scene offset).  Plain host struct; upload it with
 * vgpmp_robot_upload and pass the device copy to the kernels. */
typedef struct vgpmp_robot {
    int32_t dof;
    int32_t num_spheres;
    int32_t craig;                         /* 1 = modified (Craig) DH, 0 = classic */
    int32_t reserved;
    float dh_d[VGPMP_MAX_DOF];
    float dh_a[VGPMP_MAX_DOF];
    float cos_alpha[VGPMP_MAX_DOF];        /* cos/sin of the constant link twists (float64 -> float32) */
    float sin_alpha[VGPMP_MAX_DOF];
    float twist[VGPMP_MAX_DOF];
    float low[VGPMP_MAX_DOF];
    float high[VGPMP_MAX_DOF];
    float base[12];                        /* base pose, 3x4 row major */
    int32_t sphere_frame[VGPMP_MAX_SPHERES]; /* frame (0..dof) of each sphere, non-decreasing */
    float sphere_off[VGPMP_MAX_SPHERES][3];
    float radius[VGPMP_MAX_SPHERES];
    float sigma_obs[VGPMP_MAX_SPHERES];
    float epsilon;
    float reserved2;
    double scene_offset[3];                /* subtracted from sphere centres before the lookup */
    /* The rest is DERIVED by vgpmp_robot_upload (whatever the caller left there): the same numbers regrouped so that one
     * joint / one sphere is one aligned wide load on the device. */
    float inv_sigma_obs[VGPMP_MAX_SPHERES];  /* 1 / sigma_obs (float32) */
    float joint_tab[VGPMP_MAX_DOF][8];       /* {cos_alpha, sin_alpha, d, a, twist, low, high, high - low} */
    float sphere_a[VGPMP_MAX_SPHERES][4];    /* {offset x, y, z, frame (int32 bits)}; entries >= num_spheres: frame = dof */
    float sphere_b[VGPMP_MAX_SPHERES][2];    /* {radius, 1 / sigma_obs} */
    int32_t frame_first[VGPMP_MAX_FRAMES + 3]; /* filled by vgpmp_robot_upload: spheres [frame_first[k], frame_first[k+1])
                                                * ride on frame k (k = 0 .. dof); entries beyond dof + 1 repeat num_spheres */
} vgpmp_robot;

/* Signed distance field prepared by vgpmp_sdf_pack: one float4 {d, gx, gy, gz} per voxel of the reference's
 * data[x, y, z] (utils/sdf_utils.py:25-33), the gradient being its clamped central difference with exact zeros
 * replaced by 0.1 (utils/sdf_utils.py:100-136).  Voxel (x, y, z) sits at element
 *   VGPMP_SDF_LINEAR:  (x*ny + y)*nz + z
 *   VGPMP_SDF_BRICK4:  (((x>>2)*nby + (y>>2))*nbz + (z>>2))*64 + morton(x&3, y&3, z&3),  nb? = ceil(n?/4),
 *                      morton = z0 | y0<<1 | x0<<2 | z1<<3 | y1<<4 | x1<<5  (bit k of each local coordinate):
 *                      a 4x4x4 brick is 1 KiB, every 128-byte line of it a 2x2x2 cube, so that sphere queries that
 *                      are neighbours in space share lines and DRAM pages whatever the direction they differ in.
 * The voxel INDEX a query resolves to is the reference's (utils/sdf_utils.py:62-66) under both layouts; only the
 * address of its record differs.
 * brick_min (optional, BRICK4 only): the smallest distance inside every brick, [nbx*nby*nbz] floats.  A sphere
 * whose brick satisfies eps - (brick_min - radius) <= 0 has hinge cost exactly 0 at every voxel of the brick
 * (likelihoods/likelihood.py:131-143), so the batch likelihood kernel leaves out its table gather: results are
 * bit-identical, the (cache-resident) summary replaces the HBM access for the free-space majority of queries. */
#define VGPMP_SDF_LINEAR 0
#define VGPMP_SDF_BRICK4 1
typedef struct vgpmp_sdf {
    const void* table;                     /* dev float4[vgpmp_sdf_table_bytes / 16] */
    int32_t nx, ny, nz;
    int32_t layout;                        /* VGPMP_SDF_LINEAR or VGPMP_SDF_BRICK4 */
    double origin[3];
    double delta;
    const void* brick_min;                 /* dev float[nbx*nby*nbz], or NULL */
    /* Free-space masks (optional, BRICK4 only; built by vgpmp_sdf_free_mask from brick_min): mask k holds ONE BIT per block of
     * (1 << mask_shift)^3 voxels, blocks in [bx][by][bz] order (nb? = ceil(n? / 2^mask_shift)), bit b of 32-bit word b >> 5:
     * set iff every voxel of the block lies at least mask_clearance[k] from the obstacles.  A sphere of radius r whose voxel falls
     * in a set block of a mask with mask_clearance[k] >= epsilon + r has hinge cost exactly 0 (likelihoods/likelihood.py:131-143),
     * like the brick_min test -- but a mask is a few KB (32 KB at 512^3 with 8^3-voxel blocks) and is read out of LDS by the batch
     * likelihood kernel: the test costs no memory request.  mask_count = 0: none.  Clearances ascend; mask k starts at word
     * k * mask_words. */
    const void* free_mask;                 /* dev uint32[mask_count * mask_words], or NULL */
    int32_t mask_shift;                    /* log2 of the block edge in voxels, >= 2 */
    int32_t mask_count;                    /* 0 .. VGPMP_MAX_MASKS */
    int32_t mask_words;                    /* 32-bit words per mask, a multiple of 4 */
    int32_t reserved;
    float mask_clearance[4];
} vgpmp_sdf;
#define VGPMP_MAX_MASKS 4

/* Problem-batch dimensions. */
typedef struct vgpmp_dims {
    int32_t num_problems;
    int32_t S;             /* Monte-Carlo samples handled by this rank */
    int32_t S_total;       /* samples over all ranks (== S unless the sample axis is sharded) */
    int32_t N;             /* time points */
    int32_t M;             /* trainable inducing points (Mz = M + 2) */
    int32_t L;             /* latent GPs == dof */
    int32_t B;             /* Fourier bases, multiple of 16 */
    int32_t split_k;       /* K-slices of the prior GEMM (1, 2, 4 or 8) */
    int32_t sample_offset; /* index of this rank's first sample in the global sample stream: the prior
                            * weights / eps / eps2 of local sample s are the draws of global sample
                            * sample_offset + s (the Fourier basis omega, beta is the same on every rank) */
    int32_t reserved;
} vgpmp_dims;

/* Variational + kernel parameters in UNCONSTRAINED space, float64, with their Adam moments
 * (models/vgpmp.py:255-263, utils/miscellaneous.py:324-343).  q_sqrt keeps the M x M lower
 * triangle per latent (FillTriangular is a fixed permutation of these entries). */
typedef struct vgpmp_params {
    double* q_mu;          /* dev [P, L, M]       */
    double* q_sqrt;        /* dev [P, L, M, M]    */
    double* raw_ell;       /* dev [P, L]  lengthscale = softplus(raw)        */
    double* raw_var;       /* dev [P, L]  variance = 0.1 + softplus(raw)     */
} vgpmp_params;

/* Injected randomness of one ELBO evaluation (float32).  Same layout the generator fills. */
typedef struct vgpmp_noise {
    float* omega;          /* dev [P, L, B, D]  Student-t spectral frequencies  */
    float* beta;           /* dev [P, L, B]     phases U(0, 2 pi)               */
    float* w;              /* dev [P, S, L, B]  prior weights ~ N(0,1): any float32 when injected; drawn by the library they are float16
                            *                   VALUES (a table draw: see vgpmp_generate_noise)          */
    float* eps;            /* dev [P, S, Mz, L] N(0,1) for u = q_mu + q_sqrt eps */
    float* eps2;           /* dev [P, S, Mz, L] N(0,1) jitter perturbation      */
} vgpmp_noise;

/* The likelihood's constants as TRAINABLE variables (trainable_params.sigma_obs / alpha of
 * utils/miscellaneous.py:324-343; reference default: both fixed).  Optional: with problem->lik == NULL,
 * robot.sigma_obs and problem->alpha are the constants of every problem.  Otherwise, per problem,
 *   alpha = 1e-4 + softplus(raw_alpha)                 (models/vgpmp.py:82, positive(1e-4))
 *   sigma_obs[q] = 1e-5 + softplus(raw_sigma[q])       (likelihoods/likelihood.py:31-41, positive(1e-5))
 * whatever the VGPMP_TRAIN_* bits say (the bits only gate the Adam update).  The training loss is
 * -(ELBO + log sigmoid(raw_alpha) + sum_q log sigmoid(raw_sigma[q])): the Normal priors that
 * disable_param_opt attaches are centred on the parameters themselves, so only the bijectors' log-det-Jacobians of
 * GPflow's log_prior_density depend on the variables.  Not available with a sharded sample axis. */
typedef struct vgpmp_lik_params {
    double* raw_alpha;     /* dev [P]                       */
    double* raw_sigma;     /* dev [P, VGPMP_MAX_SPHERES]    (entries >= num_spheres unused) */
    double* m_alpha;       /* dev, Adam moments, same shapes (may be NULL without VGPMP_DO_ADAM) */
    double* v_alpha;
    double* m_sigma;
    double* v_sigma;
    double* g_alpha;       /* dev [P]                       out: d loss / d raw_alpha */
    double* g_sigma;       /* dev [P, VGPMP_MAX_SPHERES]    out: d loss / d raw_sigma */
    void* scratch;         /* dev, vgpmp_lik_scratch_bytes(dims) bytes: effective constants, per-workgroup sums */
} vgpmp_lik_params;

/* The inducing locations as TRAINABLE variables (trainable_params.inducing_variable of utils/miscellaneous.py:338; reference
 * default: fixed).  Optional: with problem->ind == NULL, problem->Zy is the one set of times every problem uses.  Otherwise,
 * per problem, Z = 0.09 + 0.82 sigmoid(raw_Z) [M, L] (models/vgpmp.py:29-42, tfb.Sigmoid(0.09, 0.91); every column its own
 * values) and Zy = [0; 1; Z]: the library writes `Zy` from raw_Z at the start of every evaluation and uses it in place of
 * problem->Zy.  Column l feeds latent l's Kuu / Kuf (kernel_conditioning/multioutput/cond_kernel.py:17-25), whole rows feed
 * the random-feature prior of every latent.  The gradient goes in REVERSE through A = Kfu (Kuu + jI)^-1, the Cholesky factor
 * (q_sqrt assembly, KL) and the prior draw at Zy.  One launch per kernel; not available with a sharded sample axis. */
typedef struct vgpmp_inducing_params {
    double* raw_Z;         /* dev [P, M, L]                   */
    double* m_Z;           /* dev, Adam moments, same shape (may be NULL without VGPMP_DO_ADAM) */
    double* v_Z;
    double* g_Z;           /* dev [P, M, L]   out: d loss / d raw_Z */
    double* Zy;            /* dev [P, Mz, L]  out: conditioned + inducing times of every problem */
    void* scratch;         /* dev, vgpmp_inducing_scratch_bytes(dims) bytes */
} vgpmp_inducing_params;

typedef struct vgpmp_problem {
    const double* X;       /* dev [N, D]   time grid (utils/miscellaneous.py:115-127)        */
    const double* Zy;      /* dev [Mz, D]  conditioned + inducing times (inducing_variables.py:73-82) */
    const double* y_u;     /* dev [P, 2, L] start/goal in unconstrained space (vgpmp.py:75-76) */
    double alpha;          /* likelihood temperature (vgpmp.py:82) */
    double jitter;         /* 1e-6 */
    double kl_scale;       /* 1 on the rank that owns the KL term, 0 elsewhere */
    uint32_t* step_counter; /* dev, optional: number of completed training steps.  When set, the noise key
                             * uses *step_counter and the Adam step count is *step_counter + 1; a
                             * VGPMP_DO_ADAM step increments it on the device, so a captured hipGraph of
                             * the steps can be replayed */
    const vgpmp_lik_params* lik; /* host pointer, optional (see vgpmp_lik_params) */
    const vgpmp_inducing_params* ind; /* host pointer, optional (see vgpmp_inducing_params) */
    vgpmp_stream aux_stream; /* reserved (NULL).  Rounds 4-5 ran stage B of the covariance path on this second stream beside the prior
                              * draws; stage B now rides in the prior kernel's launch and the field is ignored (kept: struct layout) */
} vgpmp_problem;

/* Outputs of an ELBO evaluation. */
typedef struct vgpmp_outputs {
    float* f;              /* dev [P, S, L, N]  latent paths (before the joint sigmoid)    */
    float* logp;           /* dev [P, S, N]     log p(e|f) per sample and time             */
    double* lik;           /* dev [P]           alpha/S_total * sum_{s,n} logp             */
    double* kl;            /* dev [P]           KL(q||p) * kl_scale                         */
    vgpmp_params grad;     /* dev, gradient of loss = -(lik - kl) wrt the unconstrained variables */
} vgpmp_outputs;

#define VGPMP_TRAIN_Q_MU 1
#define VGPMP_TRAIN_Q_SQRT 2
#define VGPMP_TRAIN_LENGTHSCALES 4
#define VGPMP_TRAIN_KERNEL_VARIANCE 8
#define VGPMP_TRAIN_SIGMA_OBS 16      /* needs problem->lik */
#define VGPMP_TRAIN_ALPHA 32          /* needs problem->lik */
#define VGPMP_TRAIN_INDUCING 64       /* needs problem->ind */

/* `what`: what a call computes.  (Bits not listed here are measurement and test switches: include/vgpmp_debug.h.) */
#define VGPMP_DO_FORWARD 1      /* ELBO forward only (models/vgpmp.py:265-289)               */
#define VGPMP_DO_BACKWARD 2     /* + gradient of -ELBO (utils/miscellaneous.py:77-80)        */
#define VGPMP_DO_ADAM 4         /* + Adam.apply_gradients (utils/miscellaneous.py:82)        */
#define VGPMP_GEN_NOISE 8       /* draw the noise with the device Philox generator first     */
#define VGPMP_COV_ONLY 2048     /* with VGPMP_DO_FORWARD alone: only the covariance stage -- Kuu, its Cholesky, q_sqrt, A and the
                                 * per-latent prior KL (kullback_leiblers/prior_kl.py:16-35) land in the workspace (views "kl_l", "C",
                                 * "Kinv", "A4"); no noise, no likelihood: dev_robot and sdf may be NULL, the members of `noise` and `out` too */
#define VGPMP_NOISE_AHEAD 32768  /* few problems: the call's last step also draws the NEXT step's omega, beta, w (beside its path assembly / reverse pass) */
#define VGPMP_NOISE_READY 65536  /* ... and this call's first step finds its own already drawn (a previous call ran with NOISE_AHEAD at step - 1) */

/* ---- set-up -------------------------------------------------------------------------------- */

/* Copies the host struct to device memory (`dev_robot` has sizeof(vgpmp_robot) bytes). */
int vgpmp_robot_upload(const vgpmp_robot* host_robot, void* dev_robot, vgpmp_stream stream);

/* Bytes of the voxel table and of the brick summary for a grid of nx*ny*nz voxels in `layout`. */
int vgpmp_sdf_table_bytes(int32_t nx, int32_t ny, int32_t nz, int32_t layout, size_t* table_bytes,
                          size_t* brick_min_bytes);

/* 32-bit words (rounded up to a multiple of 4) of ONE free-space mask of an nx x ny x nz grid with blocks of (1 << shift)^3 voxels. */
int vgpmp_sdf_mask_words(int32_t nx, int32_t ny, int32_t nz, int32_t shift, size_t* words);

/* Fills sdf->free_mask (mask_count masks of mask_words words, block edge 1 << mask_shift, clearances mask_clearance[]) from
 * sdf->brick_min, which must be complete (every slab packed).  Replaces nothing in the reference: a device-side acceleration
 * structure of the nearest-voxel lookup (utils/sdf_utils.py:62-76) whose use leaves every result bit-identical. */
int vgpmp_sdf_free_mask(const vgpmp_sdf* sdf, vgpmp_stream stream);

/* Builds the per-voxel {d, gx, gy, gz} records of the voxels x0 <= x < x1 of `sdf` (whose table / brick_min
 * pointers name the destination) from float64 rows of the grid data[x, y, z]: clamped central differences with
 * exact zeros replaced by 0.1 (utils/sdf_utils.py:100-136).  dev_rows holds the rows row_lo <= x < row_hi,
 * dense [row_hi - row_lo, ny, nz]; it must contain rows max(x0-1, 0) .. min(x1, nx-1), so a grid larger than a
 * staging buffer is packed slab by slab.  BRICK4: x0 must be a multiple of 4 and x1 a multiple of 4 or nx. */
int vgpmp_sdf_pack(const vgpmp_sdf* sdf, const double* dev_rows, int32_t row_lo, int32_t row_hi,
                   int32_t x0, int32_t x1, vgpmp_stream stream);

/* Signed distance grid of a triangle mesh (replaces the external SDFGen binary of utils/gen_sdf.py:16-43):
 * dev_triangles [T, 9] = vertices a, b, c of each triangle; dev_part [T] = index of the closed part each
 * triangle belongs to, non-decreasing; host `origin[3]`; grid[x,y,z] = signed distance (negative inside)
 * at origin + delta * (x, y, z).  Exact point-triangle distance, sign from the winding number per part. */
int vgpmp_mesh_sdf(const double* dev_triangles, const int32_t* dev_part, int32_t num_triangles,
                   int32_t nx, int32_t ny, int32_t nz, const double* origin, double delta,
                   double* dev_grid, vgpmp_stream stream);

/* ---- stand-alone pieces (parity tests, debugging) ------------------------------------------ */

/* Sampler.forward_kinematics_cost (utils/sampler.py:216-235): q [n, dof] -> sphere centres
 * pos [n, P, 3]; optional frames [n, dof+1, 12] (3x4 row major) of forward_kinematics (:103-120). */
int vgpmp_fk_spheres(const vgpmp_robot* dev_robot, const float* dev_q, int64_t n,
                     float* dev_pos, float* dev_frames, vgpmp_stream stream);

/* SignedDistanceField.get_distance_tf / get_distance_grad_tf (utils/sdf_utils.py:62-136) on
 * positions already relative to the scene: idx [n,3] int32, dist [n], grad [n,3] (any may be NULL). */
int vgpmp_sdf_query(const vgpmp_sdf* sdf, const double* dev_rel_pos, int64_t n,
                    int32_t* dev_idx, float* dev_dist, float* dev_grad, vgpmp_stream stream);

/* The voxel indices the ELBO kernels themselves compute (utils/sdf_utils.py:62-66 on float32 sphere centres in the robot
 * frame, `host_scene_offset` [3] subtracted in float64 as likelihoods/likelihood.py:146-176 does): float32 quotient, and
 * the reference's float64 index without a division wherever that quotient is within its error of a cell boundary.  For the
 * parity tests: idx [n,3] int32 must equal clip(trunc(((double(pos) - offset) - origin) / delta), 0, n - 1) bit for bit.
 * A first component < 0 flags a disagreement between the kernels' two index forms (-1 - index). */
int vgpmp_sdf_index_float(const vgpmp_sdf* sdf, const double* host_scene_offset, const float* dev_pos, int64_t n,
                        int32_t* dev_idx, vgpmp_stream stream);

/* VariationalMonteCarloLikelihood.log_prob (likelihoods/likelihood.py:57-176) on joint angles
 * g [n, dof]: logp [n] and, if dev_dlogp_dg != NULL, its gradient [n, dof]. */
int vgpmp_log_prob(const vgpmp_robot* dev_robot, int32_t dof, const vgpmp_sdf* sdf, const float* dev_g,
                   int64_t n, float* dev_logp, float* dev_dlogp_dg, vgpmp_stream stream);

/* Velocity-constrained kernel variant (kernels.py:4-6 FirstOrderKernelDerivativeSeparateIndependent; unreachable from
 * VGPMP.initialize in the reference): per latent l, with ny = the two conditioned times Zy[:2, l],
 *   Kuu = [[d2k(ny, ny) + 1e-6 I, dk(ny, Zy)], [dk(Zy, ny), k(Zy, Zy)]] + jitter I   [L, Mz + 2, Mz + 2]
 *   Kuf = [dk(ny, X); k(Zy, X)]                                                    [L, Mz + 2, N]
 * (covariances/multioutput/Kuus.py:17-39, Kufs.py:14-23) with dk = d k(x, y) / dy (derivatives/first_order.py:14-29)
 * and d2k = d^2 k / dx dy (second_order.py:27-58, exact zeros replaced by 5/3 / ell^2 as there).  kind: 0 Matern-5/2,
 * 1 squared exponential.  Zy [Mz, L], X [N, L], ell / var [L], all float64 on the device. */
int vgpmp_kernel_derivative(int32_t kind, int32_t order, const double* dev_x, int32_t n, const double* dev_y, int32_t m,
                            double lengthscale, double variance, double* dev_out, vgpmp_stream stream);
/* ^ the K_grad / K_grad_grad dispatchers on plain arrays (derivatives/dispatch.py): out [n, m] = k (order 0),
 * dk/dy (order 1: K_grad) or d2k/dxdy (order 2: K_grad_grad) of every pair (x_i, y_j). */

int vgpmp_velocity_kuu_kuf(int32_t kind, const double* dev_Zy, const double* dev_X, int32_t Mz, int32_t N, int32_t L,
                           const double* dev_ell, const double* dev_var, double jitter, double* dev_Kuu, double* dev_Kuf,
                           vgpmp_stream stream);

/* The covariance dispatchers on plain arrays (covariances/multioutput/Kuus.py:42-53, Kufs.py:26-34, covariances/Kfus.py:36-42
 * through kernel_conditioning/multioutput/cond_kernel.py:17-25): out [L, nz, nx], out[l, i, j] = k_l(Z[i, l], X[j, l]) with the
 * latent's own lengthscale / variance, plus `jitter` where i == j (pass 0 unless Z and X are the same set: Kuu).  Z [nz, L],
 * X [nx, L], ell / var [L], float64 on the device; kind 0 Matern-5/2, 1 squared exponential.  Kfu is the caller's transpose. */
int vgpmp_cov_matrices(int32_t kind, const double* dev_Z, int32_t nz, const double* dev_X, int32_t nx, int32_t L,
                       const double* dev_ell, const double* dev_var, double jitter, double* dev_out, vgpmp_stream stream);

/* ---- the ELBO step ------------------------------------------------------------------------- */

int vgpmp_workspace_bytes(const vgpmp_dims* dims, size_t* bytes);

/* Size of vgpmp_lik_params.scratch for these dimensions. */
int vgpmp_lik_scratch_bytes(const vgpmp_dims* dims, size_t* bytes);

/* Size of vgpmp_inducing_params.scratch for these dimensions. */
int vgpmp_inducing_scratch_bytes(const vgpmp_dims* dims, size_t* bytes);

/* Fills `noise` with the Philox-4x32-10 draws of (seed, problem index, step): this implementation's own random streams (TensorFlow's
 * generator, models/vgpmp.py:281 through GPflowSampling, cannot be reproduced).  omega (Student-t spectral draw), eps, eps2: Box-Muller
 * normals, four per counter.  The prior weights w are a TABLE draw, eight per counter: w = +-T[h & 0x1fff] for every 16-bit half h of
 * the block (sign: bit 15), T = the means of |z|, z ~ N(0, 1), over 8192 equally probable bins of the half-normal distribution as
 * float16 -- E[w] = 0, Var[w] = 1 - 5e-6, |w| <= 4.074 --, so that a generated weight is an exact f16 operand of the matrix pipe.
 * Restated bit for bit by oracle/vgpmp_oracle.py::philox_noise. */
int vgpmp_generate_noise(const vgpmp_dims* dims, const vgpmp_noise* noise, uint32_t seed,
                         uint32_t problem_base, uint32_t step, vgpmp_stream stream);

/* One evaluation of VGPMP.elbo (models/vgpmp.py:265-289) for every problem of the batch and,
 * depending on `what` (VGPMP_DO_*), its reverse pass and the Adam update of
 * utils/miscellaneous.py:68-84.  `adam_t` is the 1-based step count after this update.
 * DEPLOYMENT NOTE (profiles/r06/flake.md).  On MI355X a packed-FP32 instruction (v_pk_fma / mul / add_f32, which
 * compilers form from ordinary float arithmetic) whose op_sel and op_sel_hi both select source 1's high register reads 0.0 for it in
 * lanes 48-63 while ANOTHER wave of the same compute unit runs a wide f16 / bf16 matrix instruction (v_mfma_f32_16x16x32_f16 ...), whether
 * that wave belongs to another process or to another stream of this one (tools/pk_probe.hip).  This library contains no packed
 * instruction (a test holds the disassembly at zero), so its results do not depend on what runs beside it; but its prior draws ARE such
 * matrix instructions: other code that shares the GPU with a running planner and holds the packed form computes wrong values.  Audit it
 * (tools/audit_packed.py) or keep it off this GPU.  Applies to every vgpmp_elbo_step* entry. */
int vgpmp_elbo_step(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                    const vgpmp_problem* problem, const vgpmp_params* params,
                    const vgpmp_params* adam_m, const vgpmp_params* adam_v,
                    const vgpmp_noise* noise, const vgpmp_outputs* out,
                    void* dev_workspace, size_t workspace_bytes,
                    int32_t what, int32_t trainable, double learning_rate, int32_t adam_t,
                    uint32_t seed, uint32_t problem_base, uint32_t step, vgpmp_stream stream);

/* `num_steps` consecutive training steps (the body of training_loop, utils/miscellaneous.py:87-112) in one
 * call: step i uses noise key `step + i` and Adam count `adam_t + i` (or the device counter).  Results are
 * identical to `num_steps` calls of vgpmp_elbo_step.  For small batches independent kernels of a step share
 * launches, and the q_mu / q_sqrt update of step i runs next to the covariance and feature kernels of step
 * i+1 (which only need the kernel hyper-parameters).  Requires VGPMP_DO_ADAM | VGPMP_GEN_NOISE. */
int vgpmp_elbo_steps(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                     const vgpmp_problem* problem, const vgpmp_params* params,
                     const vgpmp_params* adam_m, const vgpmp_params* adam_v,
                     const vgpmp_noise* noise, const vgpmp_outputs* out,
                     void* dev_workspace, size_t workspace_bytes,
                     int32_t what, int32_t trainable, double learning_rate, int32_t adam_t,
                     uint32_t seed, uint32_t problem_base, uint32_t step, int32_t num_steps, vgpmp_stream stream);

/* Same kernels, one launch each, with a HIP event recorded on `stream` around every stage; synchronises the
 * stream and ADDS the elapsed milliseconds of the 8 stages {cov_fwd, noise, features, prior_gemm,
 * paths_fwd, loglik(FK+SDF), paths_bwd, final+adam} to host_stage_ms[0..7], and the device-side duration
 * (kernel start to kernel end, what a profiler reports) of the likelihood kernel and of the prior GEMM
 * kernel to host_stage_ms[8] and [9].  host_stage_ms has VGPMP_NUM_TIMES entries.  Measurement only. */
#define VGPMP_NUM_STAGES 8
#define VGPMP_NUM_TIMES 10
int vgpmp_elbo_step_profiled(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                             const vgpmp_problem* problem, const vgpmp_params* params,
                             const vgpmp_params* adam_m, const vgpmp_params* adam_v,
                             const vgpmp_noise* noise, const vgpmp_outputs* out,
                             void* dev_workspace, size_t workspace_bytes,
                             int32_t what, int32_t trainable, double learning_rate, int32_t adam_t,
                             uint32_t seed, uint32_t problem_base, uint32_t step, vgpmp_stream stream,
                             float* host_stage_ms);

/* Plan extraction, VGPMP.sample_from_posterior + get_best_sample (models/vgpmp.py:312-339), after a forward-only
 * vgpmp_elbo_step (VGPMP_DO_FORWARD | VGPMP_GEN_NOISE) at Xnew with dims.S = 150 pathwise samples, on the state that
 * call left in the workspace (A = Kfu (Kuu + jI)^-1, q_mu) and in out->f / out->logp.  Per problem:
 *   dev_mean      [P, N, L]     joint_sigmoid of the posterior mean, gpflow conditional with whiten=False (:316-317)
 *   dev_best      [P]           get_best_sample: arg-max over the samples of sum_n logp[s, n], first maximum (:336-339)
 *   dev_best_path [P, N, L]     samples[best] as joint angles
 *   dev_samples   [P, S, N, L]  every sample as joint angles (:318-320), or NULL
 *   dev_ee_var    [P, N, 3]     compute_uncertainty=True (:322-327): population variance over the samples of the last
 *                               frame's origin; the caller returns 2 sqrt of it; or NULL */
int vgpmp_sample_paths(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, void* dev_workspace, size_t workspace_bytes,
                       const float* dev_f, const float* dev_logp, float* dev_mean, int32_t* dev_best,
                       float* dev_best_path, float* dev_samples, float* dev_ee_var, vgpmp_stream stream);

/* Adam.apply_gradients alone (after an external all-reduce of out->grad when samples are sharded). */
int vgpmp_adam_step(const vgpmp_dims* dims, const vgpmp_params* params, const vgpmp_params* grad,
                    const vgpmp_params* adam_m, const vgpmp_params* adam_v, int32_t trainable,
                    double learning_rate, int32_t adam_t, vgpmp_stream stream);

/* ---- sharded Monte-Carlo sample axis: the one collective of the path ----------------------------
 * Each rank holds dims.S of the dims.S_total samples (dims.sample_offset = its first), the KL term belongs to the
 * rank with problem.kl_scale = 1.  Per step: vgpmp_elbo_step(FORWARD | BACKWARD) -> ONE in-place sum over the ranks
 * of the caller's contiguous float64 buffer [grad.q_mu | grad.q_sqrt | grad.raw_ell | grad.raw_var | lik | kl] (the
 * caller lays out->grad / lik / kl out that way: ~3.5 k doubles at M = 30, L = 7, a latency-bound message) ->
 * vgpmp_adam_step on every rank (identical update, replicated optimizer state).  The reference is single-process
 * (models/vgpmp.py:287 takes the sample mean on one device); this is what a multi-GPU binding would add.
 * RCCL over xGMI, resolved at run time; one communicator per (process, device), created on the CURRENT device. */
typedef struct vgpmp_comm vgpmp_comm;
#define VGPMP_COMM_ID_BYTES 128
/* Rank 0 creates the rendezvous id and hands the bytes to every rank by any host-side channel. */
int vgpmp_comm_unique_id(void* id_bytes);
int vgpmp_comm_init(const void* id_bytes, int32_t world_size, int32_t rank, vgpmp_comm** comm);
/* In-place sum over all ranks of `count` float64 values at dev_buf, ordered on `stream`. */
int vgpmp_allreduce_grads(vgpmp_comm* comm, double* dev_buf, size_t count, vgpmp_stream stream);
int vgpmp_comm_destroy(vgpmp_comm* comm);

/* The sample-sharded training loop in ONE call (SampleShardedPlanner.step() `num_steps` times without a host round trip per
 * step): local forward + reverse of this rank's samples (`what`: measurement flags only; forward, backward, noise generation and
 * the noise-ahead chaining are implied), the in-place all-reduce(sum) of `dev_reduce_buf` -- the caller's contiguous
 * [gradient | lik | kl] buffer that out->grad / out->lik / out->kl point into -- over `comm` (NULL: a single rank, nothing to
 * exchange), then Adam.apply_gradients (utils/miscellaneous.py:82) on every rank, all enqueued on `stream`.  `adam_t` = updates
 * applied before this call, `step` = noise key of its first step; problem->step_counter must be NULL.  The loop being sharded:
 * benchmarking.py:68-85 with the mean over samples of models/vgpmp.py:287. */
int vgpmp_elbo_steps_reduced(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                             const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                             const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                             void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                             double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base, uint32_t step,
                             int32_t num_steps, vgpmp_comm* comm, double* dev_reduce_buf, size_t reduce_count,
                             vgpmp_stream stream);

/* Reads back intermediates of the last vgpmp_elbo_step from the workspace (parity tests):
 * name in {"A","C","m","F0","H","R","G","Phi"}; returns pointer and element count. */
int vgpmp_workspace_view(const vgpmp_dims* dims, void* dev_workspace, const char* name,
                         void** dev_ptr, size_t* count, int32_t* is_double);

/* "vgpmp-hip <major.minor> (gfx950)".  0.2: vgpmp_sdf gained the free-space mask fields and vgpmp_problem aux_stream (struct
 * layouts changed against 0.1); the measurement bits of `what` moved to include/vgpmp_debug.h (same values). */
const char* vgpmp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VGPMP_H */
