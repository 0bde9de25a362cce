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
    float *epsT, *eps2T;      // optional second copies [P,L,S,Mz] (a latent's rows contiguous: what the per-latent consumers stage;
                              //  with epsT set, eps / eps' are drawn by rng_eps_t_body, not by rng_normals_body)
    int Mz, S;
    int eps_rows_log2;        // rng_eps_t_body: rows (s, k) per workgroup, 64 (few problems: many short workgroups) or 256 (batches: a
                              // 64-row workgroup is one counter on 112 of its 256 threads -- all launch cost; 16 384 of them at 896 latents)
};

__device__ __forceinline__ uint32_t rng_step(const RngArgs& a) { return a.ctr ? *a.ctr + a.bias : a.step; }

// omega [P,L,B,D] (Student-t, nu = 5: N(0,1) * rsqrt(chi2_5 / 5)) and beta [P,L,B].
// A workgroup owns kBlock consecutive rows (l, b): each thread draws its row's chi-square scale (two counters, 5 of 8
// normals) into LDS, then the workgroup walks the omega counters of its rows -- kBlock D / 4 of them, contiguous in memory --
// one counter per thread and pass: four normals, their rows' scales, ONE aligned 16-byte store.  (A thread per row wrote its
// D values by D scattered 4-byte stores and regenerated the counters that straddle two rows: 60 us at 64 x 14 latents, most
// of it in the store path.)  beta: the thread of every fourth row draws four.
__device__ __forceinline__ void rng_basis_body(const RngArgs& a, int bx, int p, float* sc_s) {      // sc_s: kBlock floats of LDS
    const int L = a.L, B = a.B, D = a.D, tid = threadIdx.x;
    VG_T(bx == 0 && p == 0, 310);
    const uint32_t rows = (uint32_t)(L * B), r0 = (uint32_t)bx * kBlock, lb = r0 + tid;
    const uint32_t nrow = min((uint32_t)kBlock, rows - r0);          // rows % 16 == 0 (B % 16 == 0): nrow D % 4 == 0
    const uint2 key = vg_key(a.seed, a.problem_base + p, rng_step(a));
    if (lb < rows) {
        const float4 v0 = vg_normal4(2u * lb, VG_STREAM_CHI, key), v1 = vg_normal4(2u * lb + 1u, VG_STREAM_CHI, key);
        const float gam = v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x;
        sc_s[tid] = __builtin_amdgcn_rsqf(gam * 0.2f);
        if ((lb & 3u) == 0u) {
            const uint4 r = vg_philox(make_uint4(lb >> 2, VG_STREAM_BETA, 0u, 0u), key);
            vg_stream(reinterpret_cast<float4*>(a.beta + (size_t)p * rows + lb),
                      make_float4(6.283185307179586f * vg_u01(r.x), 6.283185307179586f * vg_u01(r.y),
                                  6.283185307179586f * vg_u01(r.z), 6.283185307179586f * vg_u01(r.w)));
        }
    }
    __syncthreads();
    const uint32_t e0 = r0 * (uint32_t)D, nq = nrow * (uint32_t)D >> 2;      // e0 % 4 == 0: kBlock % 4 == 0
    float4* om = reinterpret_cast<float4*>(a.omega + (size_t)p * rows * D + e0);
    for (uint32_t q = tid; q < nq; q += kBlock) {
        const float4 v = vg_normal4((e0 >> 2) + q, VG_STREAM_OMEGA, key);
        uint32_t row = (4u * q) / (uint32_t)D, rem = 4u * q - row * (uint32_t)D;      // local row of element 4 q, its column
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = sc_s[row];
            if (++rem == (uint32_t)D) { rem = 0u; ++row; }
        }
        om[q] = make_float4(v.x * s[0], v.y * s[1], v.z * s[2], v.w * s[3]);      // (plain stores: the features read them back through L2)
    }
}

// w [P, nW]: counter i of the stream yields global elements 8i..8i+7 (vg_w8: float16-valued weights out of the bin-mean table; wOff and nW are multiples of 16: B is);
// eps, eps2 [P, nE]: a thread per COUNTER of the stream as well -- elements eOff .. eOff + nE - 1 of the global sample axis,
// which need not start on a counter (sample-sharded ranks): the first and last counters of a rank are partly its neighbours'.
__host__ __device__ __forceinline__ uint32_t rng_eps_quads(uint32_t nE, uint32_t eOff) {
    return nE ? ((eOff + nE - 1u) >> 2) - (eOff >> 2) + 1u : 0u;
}
__host__ __device__ __forceinline__ uint32_t rng_normal_threads(uint32_t nW, uint32_t nE, uint32_t eOff) {
    return (nW >> 3) + 2u * rng_eps_quads(nE, eOff);
}
__device__ __forceinline__ void rng_normals_body(const RngArgs& a, int bx, int p, uint32_t nW, uint32_t nE) {
    const uint32_t cW = nW >> 3, nq = rng_eps_quads(nE, a.eOff);
    uint32_t c = bx * kBlock + threadIdx.x;
    VG_T(bx == 0 && p == 0, nW ? 320 : 120);
    if (c >= cW + 2u * nq) return;
    const uint2 key = vg_key(a.seed, a.problem_base + p, rng_step(a));
    if (c < cW) {
        float z[8];
        vg_w8_f32((a.wOff >> 3) + c, key, kWTable, z);
        float4* dst = reinterpret_cast<float4*>(a.w + (size_t)p * nW + 8u * c);
        vg_stream(dst, make_float4(z[0], z[1], z[2], z[3]));
        vg_stream(dst + 1, make_float4(z[4], z[5], z[6], z[7]));
        VG_T(bx == 0 && p == 0, 321);
        VG_T(c + kBlock >= cW && p == 0, 325);
        return;
    }
    c -= cW;
    const bool second = c >= nq;
    if (second) c -= nq;
    const uint32_t q = (a.eOff >> 2) + c;                 // counter of the stream
    const float4 v = vg_normal4(q, second ? VG_STREAM_EPS2 : VG_STREAM_EPS, key);
    float* dst = (second ? a.eps2 : a.eps) + (size_t)p * nE;
    const uint32_t first = 4u * q - a.eOff;               // local element of lane 0 (wraps below zero on a rank's first counter)
    if (4u * q >= a.eOff && first + 3u < nE && ((((size_t)p * nE + first) & 3u) == 0u)) {
        vg_stream(reinterpret_cast<float4*>(dst + first), v);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = first + (uint32_t)k;       // unsigned: an element below the rank's range compares >= nE
            if (e < nE) vg_stream(dst + e, vg_lane(v, k));
        }
    }
}

// eps, eps' in BOTH layouts: [P,S,Mz,L] (the interface's) and [P,L,S,Mz] (epsT / eps2T: the consumers work per latent -- from the
// first layout each of their 4-byte requests pulled a 64-byte sector shared by the L latents).  A workgroup owns 2^eps_rows_log2
// rows (s, k) of one of the two streams: L kEpsRows consecutive elements, their counters one per thread and pass; the normals
// go out in the first layout as they are drawn and through an LDS tile in the second, 64 consecutive floats per wave.
constexpr int kEpsRowsMax = 256;
__host__ __device__ __forceinline__ uint32_t rng_eps_t_blocks(uint32_t rows, int rows_log2) { return 2u * ((rows + (1u << rows_log2) - 1u) >> rows_log2); }
// (`tile`: eps_rows L floats of the launch's dynamic LDS -- static arrays here would add to every role of the merged launches)
__device__ __forceinline__ void rng_eps_t_body(const RngArgs& a, int bx, int p, float* tile) {
    const int tid = threadIdx.x, L = a.L, sh = a.eps_rows_log2;
    const uint32_t kEpsRows = 1u << sh;
    const uint32_t rows = (uint32_t)a.S * a.Mz, half = (rows + kEpsRows - 1u) >> sh, nE = rows * (uint32_t)L;
    const bool second = (uint32_t)bx >= half;
    const uint32_t r0 = ((uint32_t)bx - (second ? half : 0u)) * kEpsRows, nr = min(kEpsRows, rows - r0);
    const uint32_t e_lo = r0 * L, e_hi = e_lo + nr * L;            // local elements of this workgroup
    VG_T(bx == 0 && p == 0, 120);
    const uint2 key = vg_key(a.seed, a.problem_base + p, rng_step(a));
    float* dst = (second ? a.eps2 : a.eps) + (size_t)p * nE;
    float* dstT = second ? a.eps2T : a.epsT;
    const uint32_t q_lo = (a.eOff + e_lo) >> 2, q_hi = (a.eOff + e_hi - 1u) >> 2;
    for (uint32_t q = q_lo + tid; q <= q_hi; q += kBlock) {
        const float4 v = vg_normal4(q, second ? VG_STREAM_EPS2 : VG_STREAM_EPS, key);
        const uint32_t first = 4u * q - a.eOff;           // local element of lane 0 (wraps below zero on a rank's first counter)
        if (4u * q >= a.eOff + e_lo && first + 3u < e_hi && ((((size_t)p * nE + first) & 3u) == 0u)) {
            vg_stream(reinterpret_cast<float4*>(dst + first), v);
            if (dstT) {
                float* t = tile + (first - e_lo);
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = first + (uint32_t)k;   // unsigned: an element below the range compares >= e_hi
                if (e >= e_lo && e < e_hi) { vg_stream(dst + e, vg_lane(v, k)); if (dstT) tile[e - e_lo] = vg_lane(v, k); }
            }
        }
    }
    if (!dstT) return;                                    // (uniform)
    __syncthreads();
    // element (r, l) of the tile -> epsT[p][l][r0 + r]: lanes along r
    for (uint32_t idx = tid; idx < (uint32_t)L * kEpsRows; idx += kBlock) {
        const uint32_t l = idx >> sh, r = idx & (kEpsRows - 1u);
        if (r < nr) vg_stream(dstT + ((size_t)p * L + l) * rows + r0 + r, tile[r * L + l]);
    }
}

__global__ __launch_bounds__(kBlock) void rng_basis_kernel(RngArgs a) {
    __shared__ float sc_s[kBlock];
    rng_basis_body(a, blockIdx.x, blockIdx.y, sc_s);
}
__global__ __launch_bounds__(kBlock) void rng_eps_t_kernel(RngArgs a) {
    __shared__ float tile[kEpsRowsMax * VGPMP_MAX_DOF];
    rng_eps_t_body(a, blockIdx.x, blockIdx.y, tile);
}
__global__ __launch_bounds__(kBlock) void rng_normals_kernel(RngArgs a) {
    rng_normals_body(a, blockIdx.x, blockIdx.y, a.nW, a.nE);
}

}  // namespace
