// Shared device helpers of the vGPMP HIP kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgpmp.h"

#define VG_WAVE 64

#define VG_CHECK_HIP(expr)                      \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

struct vg_float3 {
    float x, y, z;
};

__device__ __forceinline__ vg_float3 vg_make3(float x, float y, float z) { return vg_float3{x, y, z}; }
__device__ __forceinline__ vg_float3 vg_cross(vg_float3 a, vg_float3 b) {
    return vg_float3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float vg_dot(vg_float3 a, vg_float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ float vg_wave_sum(float v) {
#pragma unroll
    for (int o = VG_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, VG_WAVE);
    return v;
}
__device__ __forceinline__ double vg_wave_sum(double v) {
#pragma unroll
    for (int o = VG_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, VG_WAVE);
    return v;
}

// ---- nearest-voxel signed distance lookup (utils/sdf_utils.py:62-66,73-76) ------------------
// Index arithmetic in float64 in the reference's operation order ((p - offset) - origin) / delta,
// truncate, clamp, so that indices are bit-identical to the float64 reference on equal inputs.
struct vg_sdf_dev {
    const float4* table;
    int nx, ny, nz;
    double ox, oy, oz, delta;
};

__device__ __forceinline__ int vg_voxel_axis(double rel, double origin, double delta, int n) {
    double q = (rel - origin) / delta;
    int hi = n - 1;
    return q < 0.0 ? 0 : (q > (double)hi ? hi : (int)q);
}

__device__ __forceinline__ size_t vg_voxel_index(const vg_sdf_dev& s, double rx, double ry, double rz, int& ix,
                                                 int& iy, int& iz) {
    ix = vg_voxel_axis(rx, s.ox, s.delta, s.nx);
    iy = vg_voxel_axis(ry, s.oy, s.delta, s.ny);
    iz = vg_voxel_axis(rz, s.oz, s.delta, s.nz);
    return ((size_t)ix * s.ny + iy) * s.nz + iz;
}

// ---- Philox-4x32-10 (same schedule as oracle/vgpmp_oracle.py::philox4x32) --------------------
__device__ __forceinline__ uint4 vg_philox(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float vg_u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }

// four standard normals of counter (i, stream, 0, 0): Box-Muller on lanes (0,1) and (2,3)
// Hardware transcendentals: v_log_f32 (log2) and v_sin/v_cos_f32, whose argument is in REVOLUTIONS --
// exactly the uniform of Box-Muller, so no range reduction and very little code.
__device__ __forceinline__ float4 vg_normal4(uint32_t i, uint32_t stream, uint2 key) {
    uint4 r = vg_philox(make_uint4(i, stream, 0u, 0u), key);
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.z)));
    const float u1 = vg_u01(r.y), u3 = vg_u01(r.w);
    return make_float4(r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1),
                       r1 * __builtin_amdgcn_cosf(u3), r1 * __builtin_amdgcn_sinf(u3));
}
__device__ __forceinline__ float vg_lane(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ float vg_normal1(uint32_t e, uint32_t stream, uint2 key) {
    return vg_lane(vg_normal4(e >> 2, stream, key), (int)(e & 3u));
}
__device__ __forceinline__ uint2 vg_key(uint32_t seed, uint32_t problem, uint32_t step) {
    return make_uint2(seed ^ (problem * 0x9E3779B1u), step);
}

enum { VG_STREAM_OMEGA = 0, VG_STREAM_CHI = 1, VG_STREAM_BETA = 2, VG_STREAM_W = 3, VG_STREAM_EPS = 4, VG_STREAM_EPS2 = 5 };

// launchers implemented in the .hip files
int vg_launch_sdf_pack(const double* grid, int nx, int ny, int nz, double delta, float4* table, hipStream_t st);
int vg_launch_fk_spheres(const vgpmp_robot* rb, const float* q, int64_t n, float* pos, float* frames, hipStream_t st);
int vg_launch_sdf_query(const vgpmp_sdf* sdf, const double* rel, int64_t n, int32_t* idx, float* dist, float* grad,
                        hipStream_t st);
int vg_launch_log_prob_impl(const vgpmp_robot* rb, int dof, const vgpmp_sdf* sdf, const float* g, int64_t n,
                            float* logp, float* dlogp, hipStream_t st);
// ELBO-path likelihood: f [P,S,L,N] -> G [P,S,L,N] (dloss/df), logp [P,S,N], lik_partial [P, nblk]
int vg_launch_loglik_paths(const vgpmp_robot* rb, const vgpmp_sdf* sdf, const float* f, int P, int S, int L, int N,
                           float scale, float* G, float* logp, float* lik_partial, int* nblk_out, hipStream_t st);
int vg_loglik_blocks_per_problem(int S, int N);
int vg_launch_mesh_sdf(const double* tri, const int* part, int T, int nx, int ny, int nz, const double* origin,
                       double delta, double* grid, hipStream_t st);
