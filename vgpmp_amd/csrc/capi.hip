// extern "C" entry points of libvgpmp_hip.so (see include/vgpmp.h for the contract).
#include "vgpmp_device.h"
#include "gp_path.h"
#include <string.h>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

// ---- schedule log ---------------------------------------------------------------------------------------------------
// Per launch: one shared-lock lookup of an interned name and a push of its POINTER (cleaning and copying a std::string under a
// process-wide mutex on every launch was on the ~50 us / step host path of the few-problem schedules).  Names are cleaned once, when a
// launch site (its string literal) or a kernel (its function pointer) is first seen; map nodes never move.
namespace {
std::shared_mutex g_fn_mu;
std::unordered_map<const void*, std::string>& fn_names() { static std::unordered_map<const void*, std::string> m; return m; }
thread_local std::vector<const std::string*> t_sched;
const std::string kUnknown("?");
// the launch site's text -> the name a profiler prints: no enclosing parentheses, constants by value
std::string sched_clean(const char* name) {
    std::string n(name);
    while (!n.empty() && n.front() == '(' && n.back() == ')') n = n.substr(1, n.size() - 2);
    static const struct { const char* sym; const char* val; } consts[] = {{"kLikBatchBlock", "64"}};
    for (const auto& c : consts)
        for (size_t at; (at = n.find(c.sym)) != std::string::npos;) n.replace(at, strlen(c.sym), c.val);
    return n;
}
const std::string* interned(const void* key) {
    std::shared_lock<std::shared_mutex> lock(g_fn_mu);
    auto it = fn_names().find(key);
    return it == fn_names().end() ? nullptr : &it->second;
}
const std::string* intern(const void* key, const char* name) {
    std::unique_lock<std::shared_mutex> lock(g_fn_mu);
    return &fn_names().emplace(key, sched_clean(name)).first->second;
}
}  // namespace
void vg_sched_clear() { t_sched.clear(); }
void vg_sched_note(const char* name) {      // `name`: a string literal of the launch site -- its address is its identity
    const std::string* n = interned(name);
    t_sched.push_back(n ? n : intern(name, name));
}
const void* vg_fn_reg(const void* fn, const char* name) {
    if (!interned(fn)) intern(fn, name);
    return fn;
}
void vg_sched_note_fn(const void* fn) {
    const std::string* n = interned(fn);
    t_sched.push_back(n ? n : &kUnknown);
}

namespace {
// vgpmp_debug_mfma_load (include/vgpmp_debug.h): f16 matrix instructions and nothing else
typedef _Float16 vg_dbg_h8 __attribute__((ext_vector_type(8)));
typedef float vg_dbg_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void debug_mfma_load_kernel(float* __restrict__ sink, int iterations) {
    const float seed = (float)((blockIdx.x * 256u + threadIdx.x) & 1023u) * (1.0f / 1024.0f);
    vg_dbg_h8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(seed + 0.125f * k); b[k] = (_Float16)(0.5f - seed); }
    vg_dbg_f4 c = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iterations * 16; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (c[0] + c[1] + c[2] + c[3] == 123.456f) sink[0] = c[0];      // (never true: the products stay in the kernel)
}
}  // namespace

extern "C" {

const char* vgpmp_version(void) { return "vgpmp-hip 0.2 (gfx950)"; }

int vgpmp_robot_upload(const vgpmp_robot* host_robot, void* dev_robot, vgpmp_stream stream) {
    if (!host_robot || !dev_robot) return VGPMP_E_ARG;
    if (host_robot->dof < 1 || host_robot->dof > VGPMP_MAX_DOF || host_robot->num_spheres < 0 ||
        host_robot->num_spheres > VGPMP_MAX_SPHERES)
        return VGPMP_E_SHAPE;
    for (int p = 1; p < host_robot->num_spheres; ++p)
        if (host_robot->sphere_frame[p] < host_robot->sphere_frame[p - 1]) return VGPMP_E_ARG;
    for (int p = 0; p < host_robot->num_spheres; ++p)
        if (host_robot->sphere_frame[p] < 0 || host_robot->sphere_frame[p] > host_robot->dof) return VGPMP_E_ARG;
    vgpmp_robot up = *host_robot;          // the per-frame sphere ranges are derived here, whatever the caller left there
    for (int p = 0; p < VGPMP_MAX_SPHERES; ++p) {
        const bool in = p < up.num_spheres;
        up.inv_sigma_obs[p] = in ? 1.0f / up.sigma_obs[p] : 0.f;
        const int32_t fr = in ? up.sphere_frame[p] : up.dof;
        for (int k = 0; k < 3; ++k) up.sphere_a[p][k] = in ? up.sphere_off[p][k] : 0.f;
        ::memcpy(&up.sphere_a[p][3], &fr, sizeof(float));
        up.sphere_b[p][0] = in ? up.radius[p] : 0.f;
        up.sphere_b[p][1] = up.inv_sigma_obs[p];
    }
    for (int j = 0; j < VGPMP_MAX_DOF; ++j) {
        const float row[8] = {up.cos_alpha[j], up.sin_alpha[j], up.dh_d[j], up.dh_a[j], up.twist[j], up.low[j], up.high[j],
                              up.high[j] - up.low[j]};
        ::memcpy(up.joint_tab[j], row, sizeof(row));
    }
    for (int k = 0; k < VGPMP_MAX_FRAMES + 3; ++k) {
        int c = 0;
        for (int p = 0; p < up.num_spheres; ++p) c += up.sphere_frame[p] < k ? 1 : 0;
        up.frame_first[k] = c;
    }
    VG_CHECK_HIP(hipMemcpyAsync(dev_robot, &up, sizeof(vgpmp_robot), hipMemcpyHostToDevice, (hipStream_t)stream));
    // the host struct may be a temporary of the caller: make the copy complete before returning
    VG_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

static int check_sdf(const vgpmp_sdf* sdf) {
    if (!sdf || !sdf->table) return VGPMP_E_ARG;
    if (sdf->nx < 1 || sdf->ny < 1 || sdf->nz < 1 || !(sdf->delta > 0.0)) return VGPMP_E_SHAPE;
    if (sdf->layout != VGPMP_SDF_LINEAR && sdf->layout != VGPMP_SDF_BRICK4) return VGPMP_E_ARG;
    if (sdf->brick_min && sdf->layout != VGPMP_SDF_BRICK4) return VGPMP_E_ARG;
    if (sdf->free_mask) {       // the mask kernels size their LDS image and their bit indices from these fields
        if (sdf->layout != VGPMP_SDF_BRICK4) return VGPMP_E_ARG;
        if (sdf->mask_shift < 2 || sdf->mask_shift > 12 || sdf->mask_count < 1 || sdf->mask_count > VGPMP_MAX_MASKS)
            return VGPMP_E_SHAPE;
        const size_t e = (size_t)1 << sdf->mask_shift;
        const size_t bits = ((sdf->nx + e - 1) >> sdf->mask_shift) * ((sdf->ny + e - 1) >> sdf->mask_shift) *
                            ((sdf->nz + e - 1) >> sdf->mask_shift);
        if ((size_t)sdf->mask_words != ((((bits + 31) / 32) + 3) & ~(size_t)3)) return VGPMP_E_ARG;
        for (int k = 1; k < sdf->mask_count; ++k)
            if (!(sdf->mask_clearance[k] >= sdf->mask_clearance[k - 1])) return VGPMP_E_ARG;
    }
    return 0;
}

int vgpmp_sdf_table_bytes(int32_t nx, int32_t ny, int32_t nz, int32_t layout, size_t* table_bytes, size_t* brick_min_bytes) {
    if (!table_bytes) return VGPMP_E_ARG;
    if (nx < 1 || ny < 1 || nz < 1) return VGPMP_E_SHAPE;
    if (layout == VGPMP_SDF_LINEAR) {
        *table_bytes = (size_t)nx * ny * nz * 16;
        if (brick_min_bytes) *brick_min_bytes = 0;
    } else if (layout == VGPMP_SDF_BRICK4) {
        const size_t bricks = (size_t)((nx + 3) / 4) * ((ny + 3) / 4) * ((nz + 3) / 4);
        *table_bytes = bricks * 64 * 16;
        if (brick_min_bytes) *brick_min_bytes = bricks * sizeof(float);
    } else {
        return VGPMP_E_ARG;
    }
    return 0;
}

int vgpmp_sdf_pack(const vgpmp_sdf* sdf, const double* dev_rows, int32_t row_lo, int32_t row_hi, int32_t x0, int32_t x1,
                   vgpmp_stream stream) {
    int rc = check_sdf(sdf);
    if (rc) return rc;
    if (!dev_rows) return VGPMP_E_ARG;
    if (x0 < 0 || x1 > sdf->nx || x0 >= x1) return VGPMP_E_SHAPE;
    if (row_lo > (x0 > 0 ? x0 - 1 : 0) || row_hi < (x1 < sdf->nx ? x1 + 1 : sdf->nx)) return VGPMP_E_ARG;   // halo rows missing
    if (sdf->layout == VGPMP_SDF_BRICK4 && ((x0 & 3) || ((x1 & 3) && x1 != sdf->nx))) return VGPMP_E_ARG;
    return vg_launch_sdf_pack(sdf, dev_rows, row_lo, row_hi, x0, x1, (hipStream_t)stream);
}

int vgpmp_sdf_mask_words(int32_t nx, int32_t ny, int32_t nz, int32_t shift, size_t* words) {
    if (!words) return VGPMP_E_ARG;
    if (nx < 1 || ny < 1 || nz < 1 || shift < 2 || shift > 12) return VGPMP_E_SHAPE;
    const size_t e = (size_t)1 << shift;
    const size_t bits = ((nx + e - 1) >> shift) * ((ny + e - 1) >> shift) * ((nz + e - 1) >> shift);
    *words = (((bits + 31) / 32) + 3) & ~(size_t)3;
    return 0;
}

int vgpmp_sdf_free_mask(const vgpmp_sdf* sdf, vgpmp_stream stream) {
    int rc = check_sdf(sdf);
    if (rc) return rc;
    if (sdf->layout != VGPMP_SDF_BRICK4 || !sdf->brick_min || !sdf->free_mask) return VGPMP_E_ARG;
    if (sdf->mask_count < 1 || sdf->mask_count > VGPMP_MAX_MASKS) return VGPMP_E_SHAPE;
    size_t words = 0;
    if ((rc = vgpmp_sdf_mask_words(sdf->nx, sdf->ny, sdf->nz, sdf->mask_shift, &words))) return rc;
    if ((size_t)sdf->mask_words != words) return VGPMP_E_ARG;
    for (int k = 1; k < sdf->mask_count; ++k)
        if (!(sdf->mask_clearance[k] >= sdf->mask_clearance[k - 1])) return VGPMP_E_ARG;
    return vg_launch_sdf_free_mask(sdf, (hipStream_t)stream);
}

int vgpmp_mesh_sdf(const double* dev_triangles, const int32_t* dev_part, int32_t num_triangles, int32_t nx, int32_t ny,
                   int32_t nz, const double* origin, double delta, double* dev_grid, vgpmp_stream stream) {
    if (!dev_triangles || !dev_part || !origin || !dev_grid) return VGPMP_E_ARG;
    if (num_triangles < 1 || nx < 1 || ny < 1 || nz < 1 || !(delta > 0.0)) return VGPMP_E_SHAPE;
    return vg_launch_mesh_sdf(dev_triangles, dev_part, num_triangles, nx, ny, nz, origin, delta, dev_grid,
                              (hipStream_t)stream);
}

int vgpmp_debug_sphere_centres(const vgpmp_robot* dev_robot, const float* dev_f, int32_t num_problems, int32_t S, int32_t L,
                               int32_t N, int32_t what, float* dev_pos, vgpmp_stream stream) {
    if (!dev_robot || num_problems < 0 || S < 0 || N < 0) return VGPMP_E_ARG;
    if (L < 1 || L > VGPMP_MAX_DOF) return VGPMP_E_SHAPE;
    if ((size_t)num_problems * S * N > 0 && (!dev_f || !dev_pos)) return VGPMP_E_ARG;
    return vg_launch_sphere_centres(dev_robot, dev_f, num_problems, S, L, N,
                                    (what & VGPMP_LIK_LDS_STATE) ? 2 : (what & VGPMP_LIK_LANES) ? 1 : 0, dev_pos, (hipStream_t)stream);
}

int vgpmp_debug_mfma_load(float* dev_sink, int32_t workgroups, int32_t iterations, vgpmp_stream stream) {
    if (!dev_sink) return VGPMP_E_ARG;
    if (workgroups < 1 || iterations < 1) return VGPMP_E_SHAPE;
    hipLaunchKernelGGL(debug_mfma_load_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, dev_sink, (int)iterations);
    return (int)hipGetLastError();
}

int64_t vgpmp_debug_last_schedule(char* buf, size_t buf_bytes) {
    std::string all;
    for (const std::string* n : t_sched) { all += *n; all += '\n'; }
    if (buf && buf_bytes) {
        const size_t k = all.size() < buf_bytes - 1 ? all.size() : buf_bytes - 1;
        ::memcpy(buf, all.data(), k);
        buf[k] = 0;
    }
    return (int64_t)all.size() + 1;
}

int vgpmp_fk_spheres(const vgpmp_robot* dev_robot, const float* dev_q, int64_t n, float* dev_pos, float* dev_frames,
                     vgpmp_stream stream) {
    if (!dev_robot || (!dev_q && n > 0) || n < 0) return VGPMP_E_ARG;
    return vg_launch_fk_spheres(dev_robot, dev_q, n, dev_pos, dev_frames, (hipStream_t)stream);
}

int vgpmp_sdf_query(const vgpmp_sdf* sdf, const double* dev_rel_pos, int64_t n, int32_t* dev_idx, float* dev_dist,
                    float* dev_grad, vgpmp_stream stream) {
    int rc = check_sdf(sdf);
    if (rc) return rc;
    if ((!dev_rel_pos && n > 0) || n < 0) return VGPMP_E_ARG;
    return vg_launch_sdf_query(sdf, dev_rel_pos, n, dev_idx, dev_dist, dev_grad, (hipStream_t)stream);
}

int vgpmp_sdf_index_float(const vgpmp_sdf* sdf, const double* host_scene_offset, const float* dev_pos, int64_t n, int32_t* dev_idx,
                        vgpmp_stream stream) {
    int rc = check_sdf(sdf);
    if (rc) return rc;
    if (!host_scene_offset || (n > 0 && (!dev_pos || !dev_idx)) || n < 0) return VGPMP_E_ARG;
    return vg_launch_sdf_index_f32(sdf, host_scene_offset, dev_pos, n, dev_idx, (hipStream_t)stream);
}

int vgpmp_log_prob(const vgpmp_robot* dev_robot, int32_t dof, const vgpmp_sdf* sdf, const float* dev_g, int64_t n,
                   float* dev_logp, float* dev_dlogp_dg, vgpmp_stream stream) {
    int rc = check_sdf(sdf);
    if (rc) return rc;
    if (!dev_robot || (!dev_g && n > 0) || (!dev_logp && n > 0) || n < 0) return VGPMP_E_ARG;
    if (dof < 1 || dof > VGPMP_MAX_DOF) return VGPMP_E_SHAPE;
    return vg_launch_log_prob_impl(dev_robot, dof, sdf, dev_g, n, dev_logp, dev_dlogp_dg, (hipStream_t)stream);
}

int vgpmp_workspace_bytes(const vgpmp_dims* dims, size_t* bytes) {
    if (!dims || !bytes) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    vg_workspace ws;
    *bytes = vg_layout_workspace(dims, nullptr, &ws);
    return 0;
}

int vgpmp_kernel_derivative(int32_t kind, int32_t order, const double* dev_x, int32_t n, const double* dev_y, int32_t m,
                            double lengthscale, double variance, double* dev_out, vgpmp_stream stream) {
    if (!dev_x || !dev_y || !dev_out) return VGPMP_E_ARG;
    if ((kind != 0 && kind != 1) || order < 0 || order > 2 || n < 0 || m < 0 || !(lengthscale > 0.0)) return VGPMP_E_SHAPE;
    return vg_launch_kernel_derivative(kind, order, dev_x, n, dev_y, m, lengthscale, variance, dev_out, (hipStream_t)stream);
}

int vgpmp_cov_matrices(int32_t kind, const double* dev_Z, int32_t nz, const double* dev_X, int32_t nx, int32_t L,
                       const double* dev_ell, const double* dev_var, double jitter, double* dev_out, vgpmp_stream stream) {
    if (!dev_Z || !dev_X || !dev_ell || !dev_var || !dev_out) return VGPMP_E_ARG;
    if ((kind != 0 && kind != 1) || nz < 0 || nx < 0 || L < 1) return VGPMP_E_SHAPE;
    return vg_launch_cov_matrices(kind, dev_Z, nz, dev_X, nx, L, dev_ell, dev_var, jitter, dev_out, (hipStream_t)stream);
}

int vgpmp_velocity_kuu_kuf(int32_t kind, const double* dev_Zy, const double* dev_X, int32_t Mz, int32_t N, int32_t L,
                           const double* dev_ell, const double* dev_var, double jitter, double* dev_Kuu, double* dev_Kuf,
                           vgpmp_stream stream) {
    if (!dev_Zy || !dev_X || !dev_ell || !dev_var || !dev_Kuu || !dev_Kuf) return VGPMP_E_ARG;
    if ((kind != 0 && kind != 1) || Mz < 2 || N < 1 || L < 1) return VGPMP_E_SHAPE;
    return vg_launch_velocity_kuu_kuf(kind, dev_Zy, dev_X, Mz, N, L, L, dev_ell, dev_var, jitter, dev_Kuu, dev_Kuf,
                                      (hipStream_t)stream);
}

int vgpmp_sample_paths(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, void* dev_workspace, size_t workspace_bytes,
                       const float* dev_f, const float* dev_logp, float* dev_mean, int32_t* dev_best, float* dev_best_path,
                       float* dev_samples, float* dev_ee_var, vgpmp_stream stream) {
    if (!dims || !dev_robot || !dev_workspace || !dev_f || !dev_logp) return VGPMP_E_ARG;
    if ((dev_best == nullptr) != (dev_best_path == nullptr)) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    vg_workspace ws;
    if (vg_layout_workspace(dims, dev_workspace, &ws) > workspace_bytes) return VGPMP_E_WORKSPACE;
    return vg_launch_sample_paths(dims, dev_robot, &ws, dev_f, dev_logp, dev_mean, dev_best, dev_best_path, dev_samples,
                                  dev_ee_var, (hipStream_t)stream);
}

int vgpmp_lik_scratch_bytes(const vgpmp_dims* dims, size_t* bytes) {
    if (!dims || !bytes) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    vg_lik_scratch sc;
    *bytes = vg_layout_lik_scratch(dims, nullptr, &sc);
    return 0;
}

int vgpmp_inducing_scratch_bytes(const vgpmp_dims* dims, size_t* bytes) {
    if (!dims || !bytes) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    vg_ind_scratch sc;
    *bytes = vg_layout_ind_scratch(dims, nullptr, &sc);
    return 0;
}

int vgpmp_generate_noise(const vgpmp_dims* dims, const vgpmp_noise* noise, uint32_t seed, uint32_t problem_base,
                         uint32_t step, vgpmp_stream stream) {
    if (!dims || !noise || !noise->omega || !noise->beta || !noise->w || !noise->eps || !noise->eps2) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    return vg_launch_rng(dims, noise, seed, problem_base, step, nullptr, (hipStream_t)stream);
}

static int elbo_step_impl(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                          const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                          const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                          void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                          double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base, uint32_t step,
                          vgpmp_stream stream, hipEvent_t* ev, int num_steps = 1) {
    if (!dims || !problem || !params || !noise || !out || !dev_workspace) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    const bool cov_only = (what & VGPMP_COV_ONLY) != 0;      // covariance stage alone: nothing below it is launched
    if (cov_only) {
        if (what & (VGPMP_DO_BACKWARD | VGPMP_DO_ADAM | VGPMP_GEN_NOISE)) return VGPMP_E_ARG;
        if (num_steps != 1 || ev) return VGPMP_E_ARG;
        what |= VGPMP_NO_FUSE;
    } else {
        if (!dev_robot) return VGPMP_E_ARG;
        rc = check_sdf(sdf);
        if (rc) return rc;
        if (!out->f || !out->logp || !out->lik || !out->kl) return VGPMP_E_ARG;
    }
    if (!problem->X || (!problem->Zy && !problem->ind) || !problem->y_u) return VGPMP_E_ARG;
    if (!params->q_mu || !params->q_sqrt || !params->raw_ell || !params->raw_var) return VGPMP_E_ARG;
    if ((what & VGPMP_DO_BACKWARD) &&
        (!out->grad.q_mu || !out->grad.q_sqrt || !out->grad.raw_ell || !out->grad.raw_var))
        return VGPMP_E_ARG;
    if ((what & VGPMP_DO_ADAM) &&
        (!(what & VGPMP_DO_BACKWARD) || !adam_m || !adam_v || (adam_t < 1 && !problem->step_counter)))
        return VGPMP_E_ARG;
    if ((trainable & (VGPMP_TRAIN_SIGMA_OBS | VGPMP_TRAIN_ALPHA)) && !problem->lik) return VGPMP_E_ARG;
    if ((trainable & VGPMP_TRAIN_INDUCING) && !problem->ind) return VGPMP_E_ARG;
    if (const vgpmp_inducing_params* iv = problem->ind) {
        if (!iv->raw_Z || !iv->Zy || dims->S_total != dims->S) return VGPMP_E_ARG;
        if ((what & VGPMP_DO_BACKWARD) && (!iv->g_Z || !iv->scratch)) return VGPMP_E_ARG;
        if ((what & VGPMP_DO_BACKWARD) && (dims->B % 64) != 0) return VGPMP_E_SHAPE;       // 64 bases per workgroup of the feature part
        if ((what & VGPMP_DO_ADAM) && (trainable & VGPMP_TRAIN_INDUCING) && (!iv->m_Z || !iv->v_Z)) return VGPMP_E_ARG;
    }
    if (const vgpmp_lik_params* lk = problem->lik) {
        if (!lk->raw_alpha || !lk->raw_sigma || !lk->scratch || dims->S_total != dims->S) return VGPMP_E_ARG;
        if ((what & VGPMP_DO_BACKWARD) && (!lk->g_alpha || !lk->g_sigma)) return VGPMP_E_ARG;
        if ((what & VGPMP_DO_ADAM) && (trainable & VGPMP_TRAIN_ALPHA) && (!lk->m_alpha || !lk->v_alpha)) return VGPMP_E_ARG;
        if ((what & VGPMP_DO_ADAM) && (trainable & VGPMP_TRAIN_SIGMA_OBS) && (!lk->m_sigma || !lk->v_sigma)) return VGPMP_E_ARG;
    }
    vg_workspace ws;
    size_t need = vg_layout_workspace(dims, dev_workspace, &ws);
    if (workspace_bytes < need) return VGPMP_E_WORKSPACE;
    if ((what & VGPMP_DO_BACKWARD) && !vg_backward_fits(dims)) return VGPMP_E_SHAPE;      // include/vgpmp.h, "Limits"
    return vg_elbo_steps(dims, dev_robot, sdf, problem, params, adam_m, adam_v, noise, out, &ws, what, trainable,
                         learning_rate, adam_t, seed, problem_base, step, num_steps, (hipStream_t)stream, ev);
}

int vgpmp_elbo_step(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                    const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                    const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                    void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                    double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base, uint32_t step,
                    vgpmp_stream stream) {
    return elbo_step_impl(dims, dev_robot, sdf, problem, params, adam_m, adam_v, noise, out, dev_workspace,
                          workspace_bytes, what, trainable, learning_rate, adam_t, seed, problem_base, step, stream,
                          nullptr);
}

int vgpmp_elbo_steps(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                     const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                     const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                     void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                     double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base, uint32_t step,
                     int32_t num_steps, vgpmp_stream stream) {
    const int32_t need = VGPMP_DO_FORWARD | VGPMP_DO_BACKWARD | VGPMP_DO_ADAM | VGPMP_GEN_NOISE;
    if (num_steps < 1 || (what & need) != need) return VGPMP_E_ARG;
    return elbo_step_impl(dims, dev_robot, sdf, problem, params, adam_m, adam_v, noise, out, dev_workspace,
                          workspace_bytes, what, trainable, learning_rate, adam_t, seed, problem_base, step, stream,
                          nullptr, num_steps);
}

int vgpmp_elbo_steps_reduced(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                             const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                             const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                             void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                             double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base, uint32_t step,
                             int32_t num_steps, vgpmp_comm* comm, double* dev_reduce_buf, size_t reduce_count,
                             vgpmp_stream stream) {
    if (num_steps < 1 || adam_t < 0 || !adam_m || !adam_v || (comm && (!dev_reduce_buf || !reduce_count))) return VGPMP_E_ARG;
    if (what & (VGPMP_DO_ADAM | VGPMP_COV_ONLY)) return VGPMP_E_ARG;      // the update follows the exchange: this call applies it
    if (problem && problem->step_counter) return VGPMP_E_ARG;             // (the step comes from the arguments)
    // the exchange buffer and the update below carry q_mu, q_sqrt, lengthscales and kernel variance only (include/vgpmp.h:
    // trainable sigma_obs / alpha / inducing locations do not shard over samples) -- refuse instead of leaving them untrained
    if (trainable & (VGPMP_TRAIN_SIGMA_OBS | VGPMP_TRAIN_ALPHA | VGPMP_TRAIN_INDUCING)) return VGPMP_E_ARG;
    what |= VGPMP_DO_FORWARD | VGPMP_DO_BACKWARD | VGPMP_GEN_NOISE | VGPMP_NOISE_AHEAD;
    for (int i = 0; i < num_steps; ++i) {
        // local samples: forward + reverse into out->grad / lik / kl; every step but the caller's first finds its prior noise drawn
        const int32_t w = (i > 0) ? (what | VGPMP_NOISE_READY) : what;
        int rc = elbo_step_impl(dims, dev_robot, sdf, problem, params, adam_m, adam_v, noise, out, dev_workspace, workspace_bytes,
                                w, trainable, learning_rate, adam_t + i + 1, seed, problem_base, step + (uint32_t)i, stream, nullptr);
        if (rc) return rc;
        if (comm && (rc = vgpmp_allreduce_grads(comm, dev_reduce_buf, reduce_count, stream))) return rc;
        if ((rc = vg_launch_adam(dims, params, &out->grad, adam_m, adam_v, trainable, learning_rate, adam_t + i + 1, (hipStream_t)stream)))
            return rc;
    }
    return 0;
}

int vgpmp_elbo_step_profiled(const vgpmp_dims* dims, const vgpmp_robot* dev_robot, const vgpmp_sdf* sdf,
                             const vgpmp_problem* problem, const vgpmp_params* params, const vgpmp_params* adam_m,
                             const vgpmp_params* adam_v, const vgpmp_noise* noise, const vgpmp_outputs* out,
                             void* dev_workspace, size_t workspace_bytes, int32_t what, int32_t trainable,
                             double learning_rate, int32_t adam_t, uint32_t seed, uint32_t problem_base,
                             uint32_t step, vgpmp_stream stream, float* host_stage_ms) {
    if (!host_stage_ms) return VGPMP_E_ARG;
    // [0, 8]: stage boundaries on the stream;  [9, 10] / [11, 12]: device start / end of the likelihood / GEMM kernel
    constexpr int kEv = VGPMP_NUM_STAGES + 5;
    hipEvent_t ev[kEv];
    for (int i = 0; i < kEv; ++i) VG_CHECK_HIP(hipEventCreate(&ev[i]));
    int rc = elbo_step_impl(dims, dev_robot, sdf, problem, params, adam_m, adam_v, noise, out, dev_workspace,
                            workspace_bytes, what, trainable, learning_rate, adam_t, seed, problem_base, step, stream,
                            ev);
    if (rc == 0) rc = (int)hipStreamSynchronize((hipStream_t)stream);
    if (rc == 0 && (what & VGPMP_DO_BACKWARD)) {
        for (int i = 0; i < VGPMP_NUM_STAGES; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) host_stage_ms[i] += ms;
        }
        for (int k = 0; k < 2; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[VGPMP_NUM_STAGES + 1 + 2 * k], ev[VGPMP_NUM_STAGES + 2 + 2 * k]) == hipSuccess)
                host_stage_ms[VGPMP_NUM_STAGES + k] += ms;
        }
    }
    for (int i = 0; i < kEv; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

int vgpmp_adam_step(const vgpmp_dims* dims, const vgpmp_params* params, const vgpmp_params* grad,
                    const vgpmp_params* adam_m, const vgpmp_params* adam_v, int32_t trainable, double learning_rate,
                    int32_t adam_t, vgpmp_stream stream) {
    if (!dims || !params || !grad || !adam_m || !adam_v || adam_t < 1) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    return vg_launch_adam(dims, params, grad, adam_m, adam_v, trainable, learning_rate, adam_t, (hipStream_t)stream);
}

int vgpmp_workspace_view(const vgpmp_dims* dims, void* dev_workspace, const char* name, void** dev_ptr, size_t* count,
                         int32_t* is_double) {
    if (!dims || !dev_workspace || !name || !dev_ptr || !count || !is_double) return VGPMP_E_ARG;
    int rc = vg_check_dims(dims);
    if (rc) return rc;
    vg_workspace ws;
    vg_layout_workspace(dims, dev_workspace, &ws);
    return vg_workspace_lookup(dims, &ws, name, dev_ptr, count, is_double);
}

#ifdef VGPMP_BISECT
// measurement builds only: (id, 100 MHz time stamp) pairs recorded by the kernels since the last call
int vgpmp_debug_trace(unsigned long long* host_pairs, int32_t capacity) {
    int n = vg_trace_take_gp(host_pairs, capacity);
    if (n < 0) return n;
    int m = vg_trace_take_lik(host_pairs + 2 * (size_t)n, capacity - n);
    return m < 0 ? m : n + m;
}
#endif

}  // extern "C"
