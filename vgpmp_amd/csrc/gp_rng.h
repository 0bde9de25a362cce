// Philox draws of the step: omega, beta (basis) and w, eps, eps' (normals).
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

// =================================================================================================
// RNG
// =================================================================================================
struct RngArgs {
    int L, B, D;
    uint32_t nW, nE, wOff, eOff;
    float *omega, *beta, *w, *eps, *eps2;
    uint32_t seed, problem_base, step, bias;
    const uint32_t* ctr;      // device step counter: the key uses *ctr + bias instead of `step`
    float* epsT;              // optional second copy of eps as [P,L,S,Mz] (the rows stage B reads when it forms U = m + C eps)
    int Mz, S;
};

__device__ __forceinline__ uint32_t rng_step(const RngArgs& a) { return a.ctr ? *a.ctr + a.bias : a.step; }

// omega [P,L,B,D] (Student-t, nu = 5: N(0,1) * rsqrt(chi2_5 / 5)) and beta [P,L,B]
__device__ __forceinline__ void rng_basis_body(const RngArgs& a, int bx, int p) {
    const int L = a.L, B = a.B, D = a.D;
    VG_T(bx == 0 && p == 0, 310);
    const uint32_t lb = bx * kBlock + threadIdx.x;
    if (lb >= (uint32_t)(L * B)) return;
    const uint2 key = vg_key(a.seed, a.problem_base + p, rng_step(a));
    const uint32_t e0 = lb * (uint32_t)D, c_first = e0 >> 2, c_last = (e0 + D - 1) >> 2;
    float* om = a.omega + ((size_t)p * L * B + lb) * D;
    float gam = 0.f, sc = 0.f;
    // pass q = 0,1: chi-square counters (5 of 8 normals); then the omega counters of this row
#pragma nounroll
    for (uint32_t q = 0; q < 2u + (c_last - c_first + 1u); ++q) {
        const bool chi = q < 2u;
        const uint32_t c = chi ? 2u * lb + q : c_first + (q - 2u);
        const float4 v = vg_normal4(c, chi ? VG_STREAM_CHI : VG_STREAM_OMEGA, key);
        if (chi) {
            gam += q == 0u ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : v.x * v.x;
            if (q == 1u) sc = __builtin_amdgcn_rsqf(gam * 0.2f);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = 4u * c + k;
                if (e >= e0 && e < e0 + (uint32_t)D) om[e - e0] = vg_lane(v, k) * sc;      // (plain stores: a wave's rows are contiguous and merge in L2)
            }
        }
    }
    uint4 r = vg_philox(make_uint4(lb >> 2, VG_STREAM_BETA, 0u, 0u), key);
    uint32_t rb = (lb & 3u) == 0 ? r.x : (lb & 3u) == 1 ? r.y : (lb & 3u) == 2 ? r.z : r.w;
    vg_stream(a.beta + (size_t)p * L * B + lb, 6.283185307179586f * vg_u01(rb));
}

// w [P, nW]: counter i of the stream yields global elements 4i..4i+3 (wOff is a multiple of 4);
// eps, eps2 [P, nE]: one thread per element (their global offset need not be aligned).
__device__ __forceinline__ void rng_normals_body(const RngArgs& a, int bx, int p, uint32_t nW, uint32_t nE) {
    const uint32_t cW = nW >> 2;
    uint32_t c = bx * kBlock + threadIdx.x;
    VG_T(bx == 0 && p == 0, nW ? 320 : 120);
    if (c >= cW + 2u * nE) return;
    const uint2 key = vg_key(a.seed, a.problem_base + p, rng_step(a));
    if (c < cW) {
        const float4 v = vg_normal4((a.wOff >> 2) + c, VG_STREAM_W, key);
        vg_stream(reinterpret_cast<float4*>(a.w + (size_t)p * nW + 4u * c), v);
        VG_T(bx == 0 && p == 0, 321);
        VG_T(c + kBlock >= cW && p == 0, 325);
        return;
    }
    c -= cW;
    const bool second = c >= nE;
    if (second) c -= nE;
    const float v = vg_normal1(a.eOff + c, second ? VG_STREAM_EPS2 : VG_STREAM_EPS, key);
    vg_stream((second ? a.eps2 : a.eps) + (size_t)p * nE + c, v);
    if (a.epsT && !second) {                             // c = (s Mz + k) L + l
        const uint32_t l = c % (uint32_t)a.L, sk = c / (uint32_t)a.L, k = sk % (uint32_t)a.Mz, s = sk / (uint32_t)a.Mz;
        a.epsT[(((size_t)p * a.L + l) * a.S + s) * a.Mz + k] = v;
    }
}

__global__ __launch_bounds__(kBlock) void rng_basis_kernel(RngArgs a) { rng_basis_body(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(kBlock) void rng_normals_kernel(RngArgs a) {
    rng_normals_body(a, blockIdx.x, blockIdx.y, a.nW, a.nE);
}

}  // namespace
