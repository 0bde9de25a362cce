/* vgpmp_debug.h -- measurement switches and test hooks of libvgpmp_hip.so.  NOT part of the binding surface: a binder of
 * include/vgpmp.h never needs this file.  The bits below share the `what` argument of vgpmp_elbo_step* with the public VGPMP_DO_* /
 * VGPMP_GEN_NOISE / VGPMP_COV_ONLY / VGPMP_NOISE_* flags; every one of them selects another schedule or kernel form of the same
 * computation (the tests hold the forms against each other), none changes what a call computes.  Bit-identical results except where a
 * switch says otherwise: VGPMP_PRIOR_F32 / VGPMP_NO_FUSE_PRIOR (float32-MFMA prior products: agreement to float32 rounding with the
 * f16-split kernels) and VGPMP_BWD_ONE_CHUNK where the register-resident reverse kernel runs (a float32 sum of chunk sums). */
#ifndef VGPMP_DEBUG_H
#define VGPMP_DEBUG_H
#include "vgpmp.h"
#ifdef __cplusplus
extern "C" {
#endif

#define VGPMP_NO_FUSE 16        /* one launch per kernel even for small batches  */
#define VGPMP_GEMM_DIRECT 32    /* stage-2 GEMM role with operands straight from L2; with VGPMP_NO_FUSE also the prior
                                 * draws of few samples as stored features + GEMM instead of the few-sample kernel (which forms
                                 * its features inside the product, with other float32 roundings): the tests' bitwise reference */
#define VGPMP_NO_SPLIT 64       /* reverse path pass on one workgroup per (chunk, latent) */
#define VGPMP_ELIM_BLOCK 128    /* Kuu elimination by the whole workgroup through LDS instead of one wave in registers */
#define VGPMP_LIK_LANES 256     /* the batch form of the likelihood (one lane per configuration) at any batch size */
#define VGPMP_LIK_LDS_STATE 512 /* that form with the per-frame force / moment sums in LDS instead of registers */
#define VGPMP_COV_LDS_ROWS 1024 /* batches keep stage B's rows role in its LDS form (four waves per 16 time points, the inverse
                                 * formed by every row-tile workgroup) instead of one wave per 16 time points in registers: the same bits */
#define VGPMP_NO_FUSE_PRIOR 4096 /* large batches with the generator, the feature kernel and the tiled GEMM as three launches */
#define VGPMP_PRIOR_F32 8192    /* large batches form the prior draws with float32 MFMAs (the round-2 kernel) instead of
                                 * the f16-split products of prior_fused_split_kernel */
#define VGPMP_BWD_ONE_CHUNK 16384 /* reverse path pass with one sample chunk per workgroup.  Bit-identical results EXCEPT where the
                                   * register-resident reverse kernel runs (Mz = 32, N % 4 == 0, N <= 100, batches): that kernel adds the
                                   * partial sums of a workgroup's chunks in float32 and leaves one set per workgroup, so gradients differ
                                   * from the one-chunk form by the rounding of a float32 sum (<= 2e-6 relative, tests/test_gpu_surface.py) */

/* The sphere centres the ELBO kernels themselves form from latent paths: dev_f [P, S, L, N] float32 (the layout of
 * vgpmp_outputs.f) -> dev_pos [P, S, N, num_spheres, 3] float32, by the arithmetic of the likelihood launch that
 * vgpmp_elbo_step selects for the same (P, S, N, L, what): joint sigmoid (likelihoods/likelihood.py:49-52), sin / cos, the DH chain
 * (utils/sampler.py:103-120) serially or as the 8-lane prefix product of the few-problem form, sphere offsets (:237-244).  For the
 * parity tests: the nearest-voxel lookup (utils/sdf_utils.py:62-66) is piecewise constant, so a float64 oracle that looks its
 * voxels up at float64(these centres) sees the voxels the device saw, and the comparison needs no allowance for queries that
 * fell into a neighbouring cell.  (vgpmp_fk_spheres keeps the library sincosf: its centres may differ by an ulp.) */
int vgpmp_debug_sphere_centres(const vgpmp_robot* dev_robot, const float* dev_f, int32_t num_problems, int32_t S, int32_t L,
                               int32_t N, int32_t what, float* dev_pos, vgpmp_stream stream);

/* Names of the kernels the calling thread's last vgpmp_elbo_step / vgpmp_elbo_steps / vgpmp_elbo_step_profiled /
 * vgpmp_elbo_steps_reduced call enqueued for the LAST step it ran, in launch order, one per line (as a profiler prints them, template
 * arguments included), written NUL-terminated into buf[0 .. buf_bytes).  Returns the number of bytes the full text needs
 * (including the NUL); nothing is written beyond buf_bytes.  For bench.py and tools/pmc_aggregate.py: what ran is asked of the
 * library instead of being re-derived from shape rules. */
int64_t vgpmp_debug_last_schedule(char* buf, size_t buf_bytes);

/* A kernel that does nothing but f16 matrix instructions (v_mfma_f32_16x16x32_f16, the instruction of this library's prior draws):
 * `workgroups` x 256 lanes, `iterations` x 16 MFMAs per wave, few registers, no LDS -- it co-resides with anything.  For
 * tests/test_gpu_attach.py: on MI355X a wave running such an MFMA on a compute unit makes a packed-FP32 instruction of ANOTHER wave
 * of that unit read 0.0 for source 1 in lanes 48-63 when its op_sel and op_sel_hi both select that source's high register
 * (profiles/r06/flake.md, "What triggers it").  This library holds no packed instruction (vgpmp_amd/build.py); the test keeps this
 * kernel running on a second stream beside the likelihood and the ELBO step and demands bit-identical results.  dev_sink: >= 1 float. */
int vgpmp_debug_mfma_load(float* dev_sink, int32_t workgroups, int32_t iterations, vgpmp_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VGPMP_DEBUG_H */
