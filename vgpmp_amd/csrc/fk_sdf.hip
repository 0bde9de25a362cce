// Forward kinematics -> sphere placement -> signed-distance lookup -> hinge likelihood, forward and
// reverse, for gfx950.  One lane owns one joint configuration (one (sample, time) pair): the DH chain
// lives in registers, every sphere costs ONE 16-byte gather from the {d, grad} voxel table, and the
// reverse pass is the geometric Jacobian accumulated per frame (force / moment sums), so no
// intermediate leaves the registers.
//
// Reference path: likelihoods/likelihood.py:57-176, utils/sampler.py:103-120,142-244,
// utils/sdf_utils.py:62-136.
#include "vgpmp_device.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ vg_sdf_dev load_sdf(const vgpmp_sdf& s) {
    vg_sdf_dev d;
    d.table = reinterpret_cast<const float4*>(s.table);
    d.nx = s.nx; d.ny = s.ny; d.nz = s.nz;
    d.ox = s.origin[0]; d.oy = s.origin[1]; d.oz = s.origin[2];
    d.delta = s.delta;
    return d;
}

struct Frame {
    vg_float3 cx, cy, cz, t;   // rotation columns and origin
};

__device__ __forceinline__ vg_float3 axpy(float a, vg_float3 x, vg_float3 y) {
    return vg_make3(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z));
}
__device__ __forceinline__ vg_float3 lin2(float a, vg_float3 x, float b, vg_float3 y) {
    return vg_make3(fmaf(a, x.x, b * y.x), fmaf(a, x.y, b * y.y), fmaf(a, x.z, b * y.z));
}

// T_i = T_{i-1} * A_i(theta) for joint j = i-1 (0-based table index)
__device__ __forceinline__ void dh_step(const vgpmp_robot* __restrict__ rb, int j, float theta, Frame& T) {
    float st, ct;
    sincosf(theta + rb->twist[j], &st, &ct);
    const float ca = rb->cos_alpha[j], sa = rb->sin_alpha[j], d = rb->dh_d[j], a = rb->dh_a[j];
    if (rb->craig) {
        // Rx(alpha) Tx(a) Rz(theta) Tz(d)      (utils/sampler.py:190-214)
        vg_float3 y1 = lin2(ca, T.cy, sa, T.cz);
        vg_float3 z1 = lin2(-sa, T.cy, ca, T.cz);
        T.t = axpy(a, T.cx, T.t);
        vg_float3 x2 = lin2(ct, T.cx, st, y1);
        vg_float3 y2 = lin2(-st, T.cx, ct, y1);
        T.cx = x2; T.cy = y2; T.cz = z1;
        T.t = axpy(d, z1, T.t);
    } else {
        // Rz(theta) Tz(d) Tx(a) Rx(alpha)      (utils/sampler.py:142-168)
        vg_float3 x1 = lin2(ct, T.cx, st, T.cy);
        vg_float3 y1 = lin2(-st, T.cx, ct, T.cy);
        T.t = axpy(d, T.cz, axpy(a, x1, T.t));
        vg_float3 y2 = lin2(ca, y1, sa, T.cz);
        vg_float3 z2 = lin2(-sa, y1, ca, T.cz);
        T.cx = x1; T.cy = y2; T.cz = z2;
    }
}

__device__ __forceinline__ Frame base_frame(const vgpmp_robot* __restrict__ rb) {
    Frame T;
    T.cx = vg_make3(rb->base[0], rb->base[4], rb->base[8]);
    T.cy = vg_make3(rb->base[1], rb->base[5], rb->base[9]);
    T.cz = vg_make3(rb->base[2], rb->base[6], rb->base[10]);
    T.t = vg_make3(rb->base[3], rb->base[7], rb->base[11]);
    return T;
}

// log p(e | g) of one configuration and (GRAD) d logp / d g.   g, dg: register arrays.
template <int DMAX, bool GRAD>
__device__ __forceinline__ float loglik_config(const vgpmp_robot* __restrict__ rb, const vg_sdf_dev& sdf,
                                               const float (&g)[DMAX], float (&dg)[DMAX]) {
    const int D = rb->dof, P = rb->num_spheres;
    const float eps = rb->epsilon;
    const double offx = rb->scene_offset[0], offy = rb->scene_offset[1], offz = rb->scene_offset[2];
    vg_float3 az[DMAX + 1], ao[DMAX + 1], Fk[DMAX + 1], Mk[DMAX + 1];
    Frame T = base_frame(rb);
    float acc = 0.f;
    int p = 0;
#pragma unroll
    for (int i = 0; i <= DMAX; ++i) {
        if (i <= D) {
            if (i > 0) dh_step(rb, i - 1, g[i - 1], T);
            vg_float3 F = vg_make3(0.f, 0.f, 0.f), Mo = vg_make3(0.f, 0.f, 0.f);
            while (p < P && rb->sphere_frame[p] == i) {
                const float ox = rb->sphere_off[p][0], oy = rb->sphere_off[p][1], oz = rb->sphere_off[p][2];
                vg_float3 pos = axpy(ox, T.cx, axpy(oy, T.cy, axpy(oz, T.cz, T.t)));
                int ix, iy, iz;
                size_t vi = vg_voxel_index(sdf, (double)pos.x - offx, (double)pos.y - offy, (double)pos.z - offz,
                                           ix, iy, iz);
                float4 v = sdf.table[vi];
                float c = fmaxf(eps - (v.x - rb->radius[p]), 0.f);     // likelihood.py:131-143
                float cs = c / rb->sigma_obs[p];
                acc = fmaf(cs, c, acc);                                // likelihood.py:99
                if (GRAD) {
                    vg_float3 gp = vg_make3(cs * v.y, cs * v.z, cs * v.w);   // d logp / d pos
                    F = vg_make3(F.x + gp.x, F.y + gp.y, F.z + gp.z);
                    vg_float3 m = vg_cross(pos, gp);
                    Mo = vg_make3(Mo.x + m.x, Mo.y + m.y, Mo.z + m.z);
                }
                ++p;
            }
            if (GRAD) { az[i] = T.cz; ao[i] = T.t; Fk[i] = F; Mk[i] = Mo; }
        }
    }
    if (GRAD) {
        vg_float3 Fs = vg_make3(0.f, 0.f, 0.f), Ms = vg_make3(0.f, 0.f, 0.f);
        const bool craig = rb->craig != 0;
#pragma unroll
        for (int i = DMAX; i >= 1; --i) {
            if (i <= D) {
                Fs = vg_make3(Fs.x + Fk[i].x, Fs.y + Fk[i].y, Fs.z + Fk[i].z);
                Ms = vg_make3(Ms.x + Mk[i].x, Ms.y + Mk[i].y, Ms.z + Mk[i].z);
                // joint i turns about z of frame i (Craig) / frame i-1 (classic), through that frame's origin
                vg_float3 z = craig ? az[i] : az[i - 1];
                vg_float3 o = craig ? ao[i] : ao[i - 1];
                vg_float3 oxF = vg_cross(o, Fs);
                dg[i - 1] = vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z));
            } else {
                dg[i - 1] = 0.f;
            }
        }
    }
    return -0.5f * acc;
}

// ---- stand-alone log_prob: g [n, dof] row major ------------------------------------------------
template <int DMAX, bool GRAD>
__global__ __launch_bounds__(kBlock) void log_prob_kernel(const vgpmp_robot* __restrict__ rb, vgpmp_sdf sdfh,
                                                           const float* __restrict__ gq, int64_t n,
                                                           float* __restrict__ logp, float* __restrict__ dlogp) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const vg_sdf_dev sdf = load_sdf(sdfh);
    const int D = rb->dof;
    float g[DMAX], dg[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) g[j] = j < D ? gq[i * D + j] : 0.f;
    float lp = loglik_config<DMAX, GRAD>(rb, sdf, g, dg);
    logp[i] = lp;
    if (GRAD) {
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
            if (j < D) dlogp[i * D + j] = dg[j];
    }
}

// ---- ELBO path: f [P,S,L,N] -> logp [P,S,N], G = dloss/df [P,S,L,N], block partial sums ----------
template <int DMAX>
__global__ __launch_bounds__(kBlock) void loglik_paths_kernel(const vgpmp_robot* __restrict__ rb, vgpmp_sdf sdfh,
                                                               const float* __restrict__ f, int S, int L, int N,
                                                               float scale, float* __restrict__ G,
                                                               float* __restrict__ logp,
                                                               float* __restrict__ lik_partial) {
    __shared__ float red[kBlock / VG_WAVE];
    const int pb = blockIdx.y;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const bool live = idx < S * N;
    float lp = 0.f;
    if (live) {
        const vg_sdf_dev sdf = load_sdf(sdfh);
        const int s = idx / N, n = idx - s * N;
        const size_t base = ((size_t)pb * S + s) * L * N + n;
        float g[DMAX], dg[DMAX], dgdf[DMAX];
#pragma unroll
        for (int j = 0; j < DMAX; ++j) {
            if (j < L) {
                float sg = 1.0f / (1.0f + expf(-f[base + (size_t)j * N]));      // likelihood.py:49-52
                float span = rb->high[j] - rb->low[j];
                g[j] = fmaf(span, sg, rb->low[j]);
                dgdf[j] = span * sg * (1.0f - sg);
            } else {
                g[j] = 0.f; dgdf[j] = 0.f;
            }
        }
        lp = loglik_config<DMAX, true>(rb, sdf, g, dg);
        logp[((size_t)pb * S + s) * N + n] = lp;
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
            if (j < L) G[base + (size_t)j * N] = scale * dg[j] * dgdf[j];
    }
    float w = vg_wave_sum(lp);
    if ((threadIdx.x & (VG_WAVE - 1)) == 0) red[threadIdx.x / VG_WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kBlock / VG_WAVE; ++k) t += red[k];
        lik_partial[(size_t)pb * gridDim.x + blockIdx.x] = t;
    }
}

// ---- stand-alone FK: q [n, dof] -> pos [n, P, 3], frames [n, dof+1, 12] ---------------------------
__global__ __launch_bounds__(kBlock) void fk_spheres_kernel(const vgpmp_robot* __restrict__ rb,
                                                             const float* __restrict__ q, int64_t n,
                                                             float* __restrict__ pos, float* __restrict__ frames) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int D = rb->dof, P = rb->num_spheres;
    Frame T = base_frame(rb);
    int p = 0;
    for (int k = 0; k <= D; ++k) {
        if (k > 0) dh_step(rb, k - 1, q[i * D + k - 1], T);
        if (frames) {
            float* o = frames + (i * (D + 1) + k) * 12;
            o[0] = T.cx.x; o[1] = T.cy.x; o[2] = T.cz.x; o[3] = T.t.x;
            o[4] = T.cx.y; o[5] = T.cy.y; o[6] = T.cz.y; o[7] = T.t.y;
            o[8] = T.cx.z; o[9] = T.cy.z; o[10] = T.cz.z; o[11] = T.t.z;
        }
        while (p < P && rb->sphere_frame[p] == k) {
            vg_float3 x = axpy(rb->sphere_off[p][0], T.cx,
                               axpy(rb->sphere_off[p][1], T.cy, axpy(rb->sphere_off[p][2], T.cz, T.t)));
            if (pos) {
                float* o = pos + (i * P + p) * 3;
                o[0] = x.x; o[1] = x.y; o[2] = x.z;
            }
            ++p;
        }
    }
}

// ---- stand-alone SDF query on float64 relative positions -----------------------------------------
__global__ __launch_bounds__(kBlock) void sdf_query_kernel(vgpmp_sdf sdfh, const double* __restrict__ rel, int64_t n,
                                                            int32_t* __restrict__ idx, float* __restrict__ dist,
                                                            float* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const vg_sdf_dev sdf = load_sdf(sdfh);
    int ix, iy, iz;
    size_t vi = vg_voxel_index(sdf, rel[3 * i], rel[3 * i + 1], rel[3 * i + 2], ix, iy, iz);
    float4 v = sdf.table[vi];
    if (idx) { idx[3 * i] = ix; idx[3 * i + 1] = iy; idx[3 * i + 2] = iz; }
    if (dist) dist[i] = v.x;
    if (grad) { grad[3 * i] = v.y; grad[3 * i + 1] = v.z; grad[3 * i + 2] = v.w; }
}

// ---- voxel table: {d, gx, gy, gz} from the float64 grid (utils/sdf_utils.py:100-136) --------------
__global__ __launch_bounds__(kBlock) void sdf_pack_kernel(const double* __restrict__ grid, int nx, int ny, int nz,
                                                           double delta, float4* __restrict__ table) {
    const size_t total = (size_t)nx * ny * nz;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < total; v += (size_t)gridDim.x * kBlock) {
        const int iz = (int)(v % nz);
        const int iy = (int)((v / nz) % ny);
        const int ix = (int)(v / ((size_t)nz * ny));
        auto at = [&](int x, int y, int z) { return grid[((size_t)x * ny + y) * nz + z]; };
        const int xp = min(ix + 1, nx - 1), xm = max(ix - 1, 0);
        const int yp = min(iy + 1, ny - 1), ym = max(iy - 1, 0);
        const int zp = min(iz + 1, nz - 1), zm = max(iz - 1, 0);
        double gx = (at(xp, iy, iz) - at(xm, iy, iz)) / (2.0 * delta);
        double gy = (at(ix, yp, iz) - at(ix, ym, iz)) / (2.0 * delta);
        double gz = (at(ix, iy, zp) - at(ix, iy, zm)) / (2.0 * delta);
        gx = gx == 0.0 ? 0.1 : gx;
        gy = gy == 0.0 ? 0.1 : gy;
        gz = gz == 0.0 ? 0.1 : gz;
        table[v] = make_float4((float)grid[v], (float)gx, (float)gy, (float)gz);
    }
}

}  // namespace

int vg_loglik_blocks_per_problem(int S, int N) { return (S * N + kBlock - 1) / kBlock; }

int vg_launch_sdf_pack(const double* grid, int nx, int ny, int nz, double delta, float4* table, hipStream_t st) {
    size_t total = (size_t)nx * ny * nz;
    unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock);
    if (blocks > 8192u) blocks = 8192u;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sdf_pack_kernel, dim3(blocks), dim3(kBlock), 0, st, grid, nx, ny, nz, delta, table);
    return (int)hipGetLastError();
}

int vg_launch_fk_spheres(const vgpmp_robot* rb, const float* q, int64_t n, float* pos, float* frames, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fk_spheres_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rb, q, n,
                       pos, frames);
    return (int)hipGetLastError();
}

int vg_launch_sdf_query(const vgpmp_sdf* sdf, const double* rel, int64_t n, int32_t* idx, float* dist, float* grad,
                        hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sdf_query_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, *sdf, rel, n,
                       idx, dist, grad);
    return (int)hipGetLastError();
}

// dof <= 8 and dof <= 16 instantiations (register arrays are sized by the template bound)
int vg_launch_log_prob_impl(const vgpmp_robot* rb, int dof, const vgpmp_sdf* sdf, const float* g, int64_t n,
                            float* logp, float* dlogp, hipStream_t st) {
    if (n == 0) return 0;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (dof <= 8) {
        if (dlogp) hipLaunchKernelGGL((log_prob_kernel<8, true>), grid, block, 0, st, rb, *sdf, g, n, logp, dlogp);
        else hipLaunchKernelGGL((log_prob_kernel<8, false>), grid, block, 0, st, rb, *sdf, g, n, logp, dlogp);
    } else {
        if (dlogp) hipLaunchKernelGGL((log_prob_kernel<16, true>), grid, block, 0, st, rb, *sdf, g, n, logp, dlogp);
        else hipLaunchKernelGGL((log_prob_kernel<16, false>), grid, block, 0, st, rb, *sdf, g, n, logp, dlogp);
    }
    return (int)hipGetLastError();
}

int vg_launch_loglik_paths(const vgpmp_robot* rb, const vgpmp_sdf* sdf, const float* f, int P, int S, int L, int N,
                           float scale, float* G, float* logp, float* lik_partial, int* nblk_out, hipStream_t st) {
    const int nblk = vg_loglik_blocks_per_problem(S, N);
    if (nblk_out) *nblk_out = nblk;
    if (P == 0 || nblk == 0) return 0;
    dim3 grid(nblk, P), block(kBlock);
    if (L <= 8)
        hipLaunchKernelGGL((loglik_paths_kernel<8>), grid, block, 0, st, rb, *sdf, f, S, L, N, scale, G, logp,
                           lik_partial);
    else
        hipLaunchKernelGGL((loglik_paths_kernel<16>), grid, block, 0, st, rb, *sdf, f, S, L, N, scale, G, logp,
                           lik_partial);
    return (int)hipGetLastError();
}
