// Mesh -> signed distance grid on the device.  Replaces the external SDFGen binary the
// reference shells out to (gpflow_vgpmp/utils/gen_sdf.py:16-43): exact point-triangle distance by brute
// force (scene meshes have a few hundred triangles), sign from the generalized winding number of each
// closed part (|w| > 1/2 = inside), all in float64.  One lane per voxel; triangles staged through LDS.
#include "vgpmp_device.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTriTile = 128;        // triangles per LDS tile (9 doubles each)

struct d3 { double x, y, z; };
__device__ __forceinline__ d3 sub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ d3 madd(d3 a, d3 b, double t) { return d3{a.x + b.x * t, a.y + b.y * t, a.z + b.z * t}; }

// squared distance from p to triangle (a, b, c)   (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ double tri_dist2(d3 p, d3 a, d3 b, d3 c) {
    const d3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    d3 q;
    if (d1 <= 0.0 && d2 <= 0.0) q = a;
    else {
        const d3 bp = sub(p, b);
        const double d3_ = dot(ab, bp), d4 = dot(ac, bp);
        if (d3_ >= 0.0 && d4 <= d3_) q = b;
        else {
            const double vc = d1 * d4 - d3_ * d2;
            if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) q = madd(a, ab, d1 / (d1 - d3_));
            else {
                const d3 cp = sub(p, c);
                const double d5 = dot(ab, cp), d6 = dot(ac, cp);
                if (d6 >= 0.0 && d5 <= d6) q = c;
                else {
                    const double vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) q = madd(a, ac, d2 / (d2 - d6));
                    else {
                        const double va = d3_ * d6 - d5 * d4;
                        if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0)
                            q = madd(b, sub(c, b), (d4 - d3_) / ((d4 - d3_) + (d5 - d6)));
                        else {
                            const double den = 1.0 / (va + vb + vc);
                            q = madd(madd(a, ab, vb * den), ac, vc * den);
                        }
                    }
                }
            }
        }
    }
    const d3 r = sub(p, q);
    return dot(r, r);
}

__global__ __launch_bounds__(kBlock) void mesh_sdf_kernel(const double* __restrict__ tri, const int* __restrict__ part,
                                                           int T, int nx, int ny, int nz, double ox, double oy, double oz,
                                                           double delta, double* __restrict__ grid) {
    __shared__ double ts[kTriTile * 9];
    __shared__ int ps[kTriTile];
    const size_t total = (size_t)nx * ny * nz;
    const size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = v < total;
    const size_t vv = live ? v : total - 1;
    const int iz = (int)(vv % nz), iy = (int)((vv / nz) % ny), ix = (int)(vv / ((size_t)nz * ny));
    const d3 p = d3{ox + delta * ix, oy + delta * iy, oz + delta * iz};
    double best = 1e300, omega = 0.0;
    int cur = -1;
    bool inside = false;
    for (int t0 = 0; t0 < T; t0 += kTriTile) {
        const int nt = min(kTriTile, T - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 9; e += kBlock) ts[e] = tri[(size_t)t0 * 9 + e];
        for (int e = threadIdx.x; e < nt; e += kBlock) ps[e] = part[t0 + e];
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const d3 a = d3{ts[9 * t], ts[9 * t + 1], ts[9 * t + 2]};
            const d3 b = d3{ts[9 * t + 3], ts[9 * t + 4], ts[9 * t + 5]};
            const d3 c = d3{ts[9 * t + 6], ts[9 * t + 7], ts[9 * t + 8]};
            best = fmin(best, tri_dist2(p, a, b, c));
            if (ps[t] != cur) {                      // parts are contiguous: close the previous one
                inside = inside || fabs(omega) > 6.283185307179586;
                omega = 0.0;
                cur = ps[t];
            }
            // solid angle of the triangle seen from p (van Oosterom & Strackee)
            const d3 pa = sub(a, p), pb = sub(b, p), pc = sub(c, p);
            const double la = sqrt(dot(pa, pa)), lb = sqrt(dot(pb, pb)), lc = sqrt(dot(pc, pc));
            const double num = dot(pa, cross(pb, pc));
            const double den = la * lb * lc + dot(pa, pb) * lc + dot(pb, pc) * la + dot(pc, pa) * lb;
            omega += 2.0 * atan2(num, den);
        }
    }
    inside = inside || fabs(omega) > 6.283185307179586;
    if (live) grid[v] = (inside ? -1.0 : 1.0) * sqrt(best);
}

}  // namespace

int vg_launch_mesh_sdf(const double* tri, const int* part, int T, int nx, int ny, int nz, const double* origin,
                       double delta, double* grid, hipStream_t st) {
    const size_t total = (size_t)nx * ny * nz;
    if (total == 0 || T == 0) return 0;
    hipLaunchKernelGGL(mesh_sdf_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, tri, part, T,
                       nx, ny, nz, origin[0], origin[1], origin[2], delta, grid);
    return (int)hipGetLastError();
}
