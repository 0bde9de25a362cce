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

namespace {

constexpr int kBlock = 256;
constexpr int kMidMaxPL = 96;       // merged launches of the one-launch-per-kernel schedule up to this many pairs ...
constexpr int kMid2MaxPL = 192;     // ... and only cov_a | noise and hyper | final up to this many (16 problems: 333 -> 323 us)
constexpr int kFuseMaxPL = 32;      // measured on config 2: shared launches win up to 4 problems (x 7 latents), one launch per kernel from 5
__device__ __forceinline__ double matern52_dell(double t1, double t2, double ell, double var) {
    double r = fabs(t1 - t2) / ell;
    return var * exp(-kSqrt5 * r) * (5.0 * r * r / (3.0 * ell)) * (1.0 + kSqrt5 * r);
}

// e / n for 0 <= e < 2^21 via a float reciprocal (exact for n <= 4096, checked exhaustively): a 32-bit
// integer division expands to ~30 instructions on the critical path of every indexing loop
__device__ __forceinline__ int vg_div(int e, float inv_n) { return (int)(((float)e + 0.5f) * inv_n); }

// strided dot product with four independent accumulators (a dependent f64 FMA costs ~40 cycles)
__device__ __forceinline__ double dot4(const double* a, int sa, const double* b, int sb, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 < n; k += 4) {
        s0 = fma(a[k * sa], b[k * sb], s0);
        s1 = fma(a[(k + 1) * sa], b[(k + 1) * sb], s1);
        s2 = fma(a[(k + 2) * sa], b[(k + 2) * sb], s2);
        s3 = fma(a[(k + 3) * sa], b[(k + 3) * sb], s3);
    }
    for (; k < n; ++k) s0 = fma(a[k * sa], b[k * sb], s0);
    return (s0 + s1) + (s2 + s3);
}

// the same dot product on EIGHT adjacent lanes (sub = lane & 7): each takes every eighth term -- four loads per
// operand in flight per 32 terms, issued unconditionally on clamped indices and masked afterwards -- then three
// butterfly steps; every lane of the group returns the sum.  A 32-term row costs ~0.25 us instead of ~0.9.
__device__ __forceinline__ double dot8(const double* a, int sa, const double* b, int sb, int n, int sub) {
    double s0 = 0.0, s1 = 0.0;
    for (int k0 = 0; k0 < n; k0 += 32) {
        double av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(k0 + sub + 8 * j, n - 1);
            av[j] = a[k * sa]; bv[j] = b[k * sb];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double t = k0 + sub + 8 * j < n ? av[j] : 0.0;
            if (j & 1) s1 = fma(t, bv[j], s1); else s0 = fma(t, bv[j], s0);
        }
    }
    double s = s0 + s1;
    s += __shfl_xor(s, 1, VG_WAVE); s += __shfl_xor(s, 2, VG_WAVE); s += __shfl_xor(s, 4, VG_WAVE);
    return s;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = vg_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & (VG_WAVE - 1)) == 0) red[threadIdx.x / VG_WAVE] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < (int)(blockDim.x / VG_WAVE); ++k) t += red[k];
    return t;
}

// =================================================================================================
// RNG
// =================================================================================================
struct RngArgs {
    int L, B, D;
    uint32_t nW, nE, wOff, eOff;
    float *omega, *beta, *w, *eps, *eps2;
    uint32_t seed, problem_base, step, bias;
    const uint32_t* ctr;      // device step counter: the key uses *ctr + bias instead of `step`
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
    vg_stream((second ? a.eps2 : a.eps) + (size_t)p * nE + c, vg_normal1(a.eOff + c, second ? VG_STREAM_EPS2 : VG_STREAM_EPS, key));
}

__global__ __launch_bounds__(kBlock) void rng_basis_kernel(RngArgs a) { rng_basis_body(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(kBlock) void rng_normals_kernel(RngArgs a) {
    rng_normals_body(a, blockIdx.x, blockIdx.y, a.nW, a.nE);
}

struct PathArgs {
    int S, N, Mz, L, SK, NC;
    size_t slab, part_len;
    float sqrt_jitter;
    const float4* A4;
    const float *AT, *C, *CT, *CT_ell, *CT_var, *m, *F0, *H, *eps, *eps2;
    int nsplit;               // 2: paths_fwd on two workgroups per (chunk, latent), halves of the time axis
    float *R, *f;
    const float* G;
    float* part;
    int want_dell;
    int stop;
};

// sum of the SK split-K slabs: SK unconditional loads issued together, then a fixed-order tree sum
template <int SK>
__device__ __forceinline__ float read_slabs(const float* base, size_t off, size_t slab) {
    float v[SK];
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = base[off + (size_t)k * slab];
#pragma unroll
    for (int w = SK / 2; w > 0; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) v[k] += v[k + w];
    return v[0];
}

// All operands of a workgroup are staged into LDS by ONE wave of independent coalesced loads (these
// launches are latency bound: every dependent global access costs ~0.3-0.7 us), then the loops run
// out of LDS.
// sum of the SK split-K slabs of an LDS image [SK][n]: fixed-order tree
template <int SK>
__device__ __forceinline__ float sum_slabs_lds(const float* raw, int e, int n) {
    float v[SK];
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = raw[k * n + e];
#pragma unroll
    for (int w = SK / 2; w > 0; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) v[k] += v[k + w];
    return v[0];
}

// Every operand of a workgroup goes global -> LDS by DMA (vg_stage_*), all requests in flight together,
// then the loops run out of LDS.  Launches of this size are latency bound: what counts is the number of
// dependent memory round trips, here one.
// RAW: the split-K slabs are staged as they are ([SK][SC][J] of LDS) and summed from LDS; otherwise (LDS
// too small for that) they are summed from registers while the other operands arrive.
template <int SK, int SC, bool RAW>
__device__ __forceinline__ void paths_fwd_body(const PathArgs& a, float* smf, int ch, int l, int p) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L, J = N + Mz, ld = Mz + 1;
    const float iMz = 1.0f / (float)Mz, iN = 1.0f / (float)N, iJ = 1.0f / (float)J, ild = 1.0f / (float)ld;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };      // 16-byte aligned regions
    float* Cs = take(Mz * ld);         // [Mz][ld]
    float* ATs = take(Mz * N);         // [Mz][N]
    float* es = take(2 * SC * Mz);     // [SC][Mz] eps, then eps2
    float* e2s = es + SC * Mz;
    float* ms = take(Mz);              // [Mz]
    float* rs = take(SC * Mz);         // [SC][Mz]
    float* f0s = take(SC * J);         // [SC][J]   prior draws (split-K slabs summed)
    float* raw = take(0);              // [SK][SC][J] the slabs as they arrive
    const int s_base = ch * SC;
    VG_T(ch == 0 && l == 0 && p == 0, 300);
    {
        const float* Cg = a.C + pl * Mz * Mz;
        vg_stage_words(Cs, Mz * ld, tid, nt, [&](int i) -> const void* {
            const int r = vg_div(i, ild), c = i - r * ld;
            return Cg + r * Mz + min(c, Mz - 1);                       // the pad column repeats the last one
        });
        const float* ATg = a.AT + pl * N * Mz;
        vg_stage_rows(ATs, Mz, N, tid, nt, [&](int r) -> const float* { return ATg + (size_t)r * N; });
        vg_stage_words(es, 2 * SC * Mz, tid, nt, [&](int i) -> const void* {
            const int second = i >= SC * Mz, e = second ? i - SC * Mz : i;
            const int sl = vg_div(e, iMz), k = e - sl * Mz, s = min(s_base + sl, S - 1);
            return (second ? a.eps2 : a.eps) + (((size_t)p * S + s) * Mz + k) * L + l;
        });
        vg_stage_words(ms, Mz, tid, nt, [&](int i) -> const void* { return a.m + pl * Mz + i; });
        if (RAW)      // one slab: straight into its final place
            vg_stage_rows(SK == 1 ? f0s : raw, SK * SC, J, tid, nt, [&](int r) -> const float* {
                const int k = r / SC, s = min(s_base + (r - k * SC), S - 1);
                return a.F0 + (size_t)k * a.slab + (((size_t)p * S + s) * L + l) * J;
            });
        else
            for (int e = tid; e < SC * J; e += nt) {
                const int sl = vg_div(e, iJ), j = e - sl * J, s = min(s_base + sl, S - 1);
                f0s[e] = read_slabs<SK>(a.F0, (((size_t)p * S + s) * L + l) * J + j, a.slab);
            }
    }
    vg_dma_wait();
    __syncthreads();
    VG_T(ch == 0 && l == 0 && p == 0, 301);
    if (RAW && SK > 1) {
        for (int e = tid; e < SC * J; e += nt) f0s[e] = sum_slabs_lds<SK>(raw, e, SC * J);
        __syncthreads();
    }
    if (Mz == 32 && SC == 8 && nt == 256) {
        // Mz = 32: u = m + eps C^T (two 16-column tiles) and f = F0 + R A^T (one 16-point tile per wave and round) on the
        // f32 MFMA pipe, 8 of 16 rows used, instead of 32-long scalar chains per thread
        const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
        if (wv < 2) {
            const int mi = 16 * wv + i;
            vg_f32x4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = ms[mi];
            const float* ep = es + min(i, SC - 1) * 32;
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(i < SC ? ep[k + kk] : 0.f, Cs[mi * ld + k + kk], acc, 0, 0, 0);
            if (kk < SC / 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = 4 * kk + q, e = sl * 32 + mi, s = s_base + sl;
                    const float r = acc[q] - f0s[sl * J + N + mi] - a.sqrt_jitter * e2s[e];
                    rs[e] = r;
                    if (s < S) vg_stream(a.R + (((size_t)p * S + s) * L + l) * 32 + mi, r);
                }
            }
        }
        __syncthreads();
        for (int t = wv; 16 * t < N; t += 4) {
            const int n = min(16 * t + i, N - 1);
            vg_f32x4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = kk < SC / 4 ? f0s[(4 * kk + q) * J + n] : 0.f;
            const float* rp = rs + min(i, SC - 1) * 32;
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(i < SC ? rp[k + kk] : 0.f, ATs[(k + kk) * N + n], acc, 0, 0, 0);
            if (kk < SC / 4 && 16 * t + i < N) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = s_base + 4 * kk + q;
                    if (s < S) vg_stream(a.f + (((size_t)p * S + s) * L + l) * N + n, acc[q]);
                }
            }
        }
    } else {
        for (int e = tid; e < SC * Mz; e += nt) {
            const int sl = vg_div(e, iMz), mi = e - sl * Mz, s = s_base + sl;
            float u = ms[mi];
            for (int k = 0; k <= mi; ++k) u = fmaf(Cs[mi * ld + k], es[sl * Mz + k], u);
            const float r = u - f0s[sl * J + N + mi] - a.sqrt_jitter * e2s[e];
            rs[e] = r;
            if (s < S) vg_stream(a.R + (((size_t)p * S + s) * L + l) * Mz + mi, r);
        }
        __syncthreads();
        for (int e = tid; e < SC * N; e += nt) {
            const int sl = vg_div(e, iN), n = e - sl * N, s = s_base + sl;
            float v = f0s[sl * J + n];
            for (int k = 0; k < Mz; ++k) v = fmaf(ATs[k * N + n], rs[sl * Mz + k], v);
            if (s < S) vg_stream(a.f + (((size_t)p * S + s) * L + l) * N + n, v);
        }
    }
    VG_T(ch == 0 && l == 0 && p == 0, 302);
    VG_T(ch == a.NC - 1 && l == L - 1 && p == 0, 305);
}

// The same on TWO workgroups per (sample chunk, latent) for launches that leave half the chip idle (see
// paths_bwd_split): both halves form R (half 0 stores it), each assembles f on its half of the time points, so a
// workgroup stages ~25 instead of ~36 KB.  q_sqrt comes transposed (16-byte rows, conflict-free reads).  With a
// run-time Mz the arithmetic and its order are those of paths_fwd_body (identical bits); the Mz = 32 instance runs
// the two products on the MFMA pipe.  Needs SK > 1, N % 4 == 0, Mz % 4 == 0.
// MZ = 32 fixes the inducing extent at compile time: loops with a run-time trip count stay rolled (load, wait, one
// FMA per iteration), with a constant one their operands are requested together.
template <int SK, int MZ = 0>
__device__ __forceinline__ void paths_fwd_split_body(const PathArgs& a, float* smf, int ch2, int l, int p) {
    constexpr int SC = 8;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = MZ ? MZ : a.Mz, L = a.L, J = N + Mz;
    const int ch = ch2 >> 1, half = ch2 & 1;
    const int Nh = (N >> 1) & ~3, n0 = half ? Nh : 0, nx = half ? N - Nh : Nh;
    const float iMz = 1.0f / (float)Mz, inx = 1.0f / (float)nx;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };
    float* CTs = take(Mz * Mz);        // [Mz][Mz] q_sqrt^T
    float* ATs = take(Mz * nx);        // [Mz][nx]
    float* es = take(2 * SC * Mz);     // [SC][Mz] eps, then eps2
    float* e2s = es + SC * Mz;
    float* ms = take(Mz);              // [Mz]
    float* rs = take(SC * Mz);         // [SC][Mz]
    float* f0x = take(SC * nx);        // [SC][nx] prior draws at the time points
    float* f0z = take(SC * Mz);        // [SC][Mz] ... at the inducing points
    float* rawx = take(SK * SC * nx);  // the split-K slabs as they arrive
    float* rawz = take(SK * SC * Mz);
    const int s_base = ch * SC;
    VG_T(ch2 == 0 && l == 0 && p == 0, 300);
    {
        const float* CTg = a.CT + pl * Mz * Mz;
        vg_stage_rows(CTs, Mz, Mz, tid, nt, [&](int r) -> const float* { return CTg + (size_t)r * Mz; });
        const float* ATg = a.AT + pl * N * Mz + n0;
        vg_stage_rows(ATs, Mz, nx, tid, nt, [&](int r) -> const float* { return ATg + (size_t)r * N; });
        vg_stage_words(es, 2 * SC * Mz, tid, nt, [&](int i) -> const void* {
            const int second = i >= SC * Mz, e = second ? i - SC * Mz : i;
            const int sl = vg_div(e, iMz), k = e - sl * Mz, s = min(s_base + sl, S - 1);
            return (second ? a.eps2 : a.eps) + (((size_t)p * S + s) * Mz + k) * L + l;
        });
        vg_stage_words(ms, Mz, tid, nt, [&](int i) -> const void* { return a.m + pl * Mz + i; });
        auto slab_row = [&](int r) -> const float* {
            const int k = r / SC, s = min(s_base + (r - k * SC), S - 1);
            return a.F0 + (size_t)k * a.slab + (((size_t)p * S + s) * L + l) * J;
        };
        vg_stage_rows(rawx, SK * SC, nx, tid, nt, [&](int r) -> const float* { return slab_row(r) + n0; });
        vg_stage_rows(rawz, SK * SC, Mz, tid, nt, [&](int r) -> const float* { return slab_row(r) + N; });
    }
    vg_dma_wait();
    __syncthreads();
    VG_T(ch2 == 0 && l == 0 && p == 0, 301);
    for (int e = tid; e < SC * nx; e += nt) f0x[e] = sum_slabs_lds<SK>(rawx, e, SC * nx);
    for (int e = tid; e < SC * Mz; e += nt) f0z[e] = sum_slabs_lds<SK>(rawz, e, SC * Mz);
    __syncthreads();
    if (MZ == 32) {
        // Mz = 32: both products as 16 x 16 tiles (8 sample rows used) on the f32 MFMA pipe -- u = m + eps C^T on two
        // waves (16 columns each), f = F0 + R A^T on one wave per 16 time points -- instead of 32-long scalar chains.
        // (The accumulation order inside a product differs from the scalar form: f is no longer bit-identical to
        // paths_fwd_body, only to float32 rounding.)
        const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
        if (wv < 2) {
            const int mi = 16 * wv + i;
            vg_f32x4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = ms[mi];                      // C operand: m broadcast over the rows
            const float* ep = es + min(i, SC - 1) * Mz;
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(i < SC ? ep[k + kk] : 0.f, CTs[(k + kk) * Mz + mi], acc, 0, 0, 0);
            if (kk < SC / 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = 4 * kk + q, e = sl * Mz + mi, s = s_base + sl;
                    const float r = acc[q] - f0z[e] - a.sqrt_jitter * e2s[e];
                    rs[e] = r;
                    if (half == 0 && s < S) vg_stream(a.R + (((size_t)p * S + s) * L + l) * Mz + mi, r);
                }
            }
        }
        __syncthreads();
        for (int t = wv; 16 * t < nx; t += 4) {
            const int j = min(16 * t + i, nx - 1);
            vg_f32x4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = kk < SC / 4 ? f0x[(4 * kk + q) * nx + j] : 0.f;
            const float* rp = rs + min(i, SC - 1) * Mz;
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(i < SC ? rp[k + kk] : 0.f, ATs[(k + kk) * nx + j], acc, 0, 0, 0);
            if (kk < SC / 4 && 16 * t + i < nx) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = s_base + 4 * kk + q;
                    if (s < S) vg_stream(a.f + (((size_t)p * S + s) * L + l) * N + n0 + j, acc[q]);
                }
            }
        }
    } else {
        for (int e = tid; e < SC * Mz; e += nt) {
            const int sl = vg_div(e, iMz), mi = e - sl * Mz, s = s_base + sl;
            float u = ms[mi];
            for (int k = 0; k <= mi; ++k) u = fmaf(CTs[k * Mz + mi], es[sl * Mz + k], u);
            const float r = u - f0z[e] - a.sqrt_jitter * e2s[e];
            rs[e] = r;
            if (half == 0 && s < S) vg_stream(a.R + (((size_t)p * S + s) * L + l) * Mz + mi, r);
        }
        __syncthreads();
        for (int e = tid; e < SC * nx; e += nt) {
            const int sl = vg_div(e, inx), j = e - sl * nx, s = s_base + sl;
            float v = f0x[e];
            for (int k = 0; k < Mz; ++k) v = fmaf(ATs[k * nx + j], rs[sl * Mz + k], v);
            if (s < S) vg_stream(a.f + (((size_t)p * S + s) * L + l) * N + n0 + j, v);
        }
    }
    VG_T(ch2 == 0 && l == 0 && p == 0, 302);
    VG_T(ch2 == 2 * a.NC - 1 && l == L - 1 && p == 0, 305);
}

// Reverse of the path assembly over one chunk of samples.  With G = dloss/df:
//   dR = G A,  dm = sum_s dR,  dC = dR^T eps                     (-> q_mu, q_sqrt)
//   hyper-parameters by dot products with the forward-mode tangents of the covariance kernels:
//   s_ell = <R, G A_ell> + <dR, C_ell eps> + <G, H_X> - <dR, H_Z>
//   s_var = <R, G A_var> + <dR, C_var eps> ;  s_rff = <G, F0_X> - <dR, F0_Z>   (x 1/(2 var) later)
template <int SK, bool RAW>
__global__ __launch_bounds__(kBlock) void paths_bwd_sc8(PathArgs a) {
    constexpr int SC = 8;
    extern __shared__ float smf[];
    __shared__ float red[3][kBlock / VG_WAVE];
    const int ch = blockIdx.x, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L, J = N + Mz;
    const float iMz = 1.0f / (float)Mz, iN = 1.0f / (float)N, iJ = 1.0f / (float)J;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };      // 16-byte aligned regions
    float4* A4s = reinterpret_cast<float4*>(take(4 * N * Mz));      // [N][Mz] {A, A_ell, A_var, -}
    float* Ces = take(2 * Mz * Mz);                  // [Mz][Mz] (dC/dell)^T, then (dC/dvar)^T
    float* Cvs = Ces + Mz * Mz;
    float* Gs = take(SC * N);                        // [SC][N]
    float* f0s = take(2 * SC * J);                   // [SC][J] prior draws, then [SC][J] their d/dell
    float* hs = f0s + SC * J;
    float* Rs = take(SC * Mz);                       // [SC][Mz]
    float* Es = take(SC * Mz);                       // [SC][Mz]
    float* dRs = take(SC * Mz);                      // [SC][Mz]
    float* dGA = take(5 * SC * Mz);                  // [5][SC][Mz] G A, G A_ell, G A_var, eps C_var^T, eps C_ell^T (MFMA form)
    float* raw = take(0);                            // [2][SK][SC][J] slabs of F0 and H as they arrive (RAW)
    const int s_base = ch * SC;
    VG_T(ch == 0 && l == 0 && p == 0, 500);
    {
        vg_stage_16(A4s, a.A4 + pl * N * Mz, N * Mz, tid, nt);
        const float* Ce = a.CT_ell + pl * Mz * Mz;
        const float* Cv = a.CT_var + pl * Mz * Mz;
        const bool dell = a.want_dell != 0;
        vg_stage_rows(Ces, 2 * Mz, Mz, tid, nt, [&](int r) -> const float* {
            return r < Mz ? (dell ? Ce + (size_t)r * Mz : nullptr) : Cv + (size_t)(r - Mz) * Mz;
        });
        vg_stage_rows(Gs, SC, N, tid, nt, [&](int r) -> const float* {
            const int s = s_base + r;
            return s < S ? a.G + (((size_t)p * S + s) * L + l) * N : nullptr;              // zero beyond S
        });
        vg_stage_rows(Rs, SC, Mz, tid, nt, [&](int r) -> const float* {
            const int s = s_base + r;
            return s < S ? a.R + (((size_t)p * S + s) * L + l) * Mz : nullptr;
        });
        vg_stage_words(Es, SC * Mz, tid, nt, [&](int i) -> const void* {
            const int sl = vg_div(i, iMz), mi = i - sl * Mz, s = s_base + sl;
            return s < S ? a.eps + (((size_t)p * S + s) * Mz + mi) * L + l : nullptr;
        });
        if (RAW && SK == 1) {      // one slab: straight into its final place (f0s and hs are adjacent)
            vg_stage_rows(f0s, 2 * SC, J, tid, nt, [&](int r) -> const float* {
                const int second = r >= SC, s = min(s_base + (second ? r - SC : r), S - 1);
                if (second && !dell) return nullptr;
                return (second ? a.H : a.F0) + (((size_t)p * S + s) * L + l) * J;
            });
        } else if (RAW) {
            vg_stage_rows(raw, (dell ? 2 : 1) * SK * SC, J, tid, nt, [&](int r) -> const float* {
                const int second = r >= SK * SC, rr = second ? r - SK * SC : r;
                const int k = rr / SC, s = min(s_base + (rr - k * SC), S - 1);
                return (second ? a.H : a.F0) + (size_t)k * a.slab + (((size_t)p * S + s) * L + l) * J;
            });
        } else {
            for (int e = tid; e < SC * J; e += nt) {
                const int sl = vg_div(e, iJ), j = e - sl * J, s = min(s_base + sl, S - 1);
                const size_t fo = (((size_t)p * S + s) * L + l) * J + j;
                f0s[e] = read_slabs<SK>(a.F0, fo, a.slab);
                hs[e] = dell ? read_slabs<SK>(a.H, fo, a.slab) : 0.f;
            }
        }
    }
    vg_dma_wait();
    __syncthreads();
    if (RAW && SK > 1) {
        const int nsl = SK * SC * J;
        for (int e = tid; e < SC * J; e += nt) {
            f0s[e] = sum_slabs_lds<SK>(raw, e, SC * J);
            hs[e] = a.want_dell ? sum_slabs_lds<SK>(raw + nsl, e, SC * J) : 0.f;
        }
        __syncthreads();
    }
    VG_T(ch == 0 && l == 0 && p == 0, 501);
    VG_STOP(a, 1);
    float se = 0.f, sv = 0.f, sr = 0.f;
    // Mz = 32: the five small products as 16 x 16 MFMA tiles (8 sample rows used) -- waves 0..2 one component of G A each
    // (both column halves), wave 3 the two triangular products -- instead of N-long scalar chains per thread
    const bool tiles = Mz == 32 && (N & 3) == 0 && nt == 256;
    if (tiles) {
        const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
        if (wv < 3) {
            vg_f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const float* gp = Gs + min(i, SC - 1) * N;
            const float* ap = reinterpret_cast<const float*>(A4s) + wv;      // component wv of the float4 at [n][mi]
            for (int n = 0; n < N; n += 4) {
                const float a0 = i < SC ? gp[n + kk] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, ap[((n + kk) * 32 + 16 * h + i) * 4], acc[h], 0, 0, 0);
            }
            if (kk < SC / 4) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dGA[(wv * SC + 4 * kk + q) * 32 + 16 * h + i] = acc[h][q];
            }
        } else {
            vg_f32x4_t accv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acce[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const float* ep = Es + min(i, SC - 1) * 32;
#pragma unroll
            for (int k = 0; k < 32; k += 4) {
                const float a0 = i < SC ? ep[k + kk] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    accv[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, Cvs[(k + kk) * 32 + 16 * h + i], accv[h], 0, 0, 0);
                    acce[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, Ces[(k + kk) * 32 + 16 * h + i], acce[h], 0, 0, 0);
                }
            }
            if (kk < SC / 4) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dGA[(3 * SC + 4 * kk + q) * 32 + 16 * h + i] = accv[h][q];
                        dGA[(4 * SC + 4 * kk + q) * 32 + 16 * h + i] = acce[h][q];
                    }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < SC * Mz; e += nt) {
        const int sl = vg_div(e, iMz), mi = e - sl * Mz;
        const float* g = Gs + sl * N;
        float d = 0.f, de = 0.f, dv = 0.f;
        if (tiles) { d = dGA[e]; de = dGA[SC * 32 + e]; dv = dGA[2 * SC * 32 + e]; }
        for (int n = 0; !tiles && n < N; ++n) {
            const float4 av = A4s[n * Mz + mi];
            const float gv = g[n];
            d = fmaf(gv, av.x, d);
            de = fmaf(gv, av.y, de);
            dv = fmaf(gv, av.z, dv);
        }
        dRs[e] = d;
        float ue = 0.f, uv = 0.f;
        if (tiles) { uv = dGA[3 * SC * 32 + e]; ue = dGA[4 * SC * 32 + e]; }
        for (int k = 0; !tiles && k <= mi; ++k) {
            const float ev = Es[sl * Mz + k];
            uv = fmaf(Cvs[k * Mz + mi], ev, uv);
            ue = fmaf(Ces[k * Mz + mi], ev, ue);
        }
        const float rv = Rs[e];
        sv += rv * dv + d * uv;
        se += rv * de + d * ue - d * hs[sl * J + N + mi];
        sr -= d * f0s[sl * J + N + mi];
    }
    for (int e = tid; e < SC * N; e += nt) {
        const int sl = vg_div(e, iN), n = e - sl * N;
        const float gv = Gs[e];             // zero for samples beyond S
        sr = fmaf(gv, f0s[sl * J + n], sr);
        se = fmaf(gv, hs[sl * J + n], se);
    }
    __syncthreads();
    VG_T(ch == 0 && l == 0 && p == 0, 502);
    VG_STOP(a, 3);
    float* out = a.part + (pl * a.NC + ch) * a.part_len;
    for (int mi = tid; mi < Mz; mi += nt) {
        float t = 0.f;
        for (int sl = 0; sl < SC; ++sl) t += dRs[sl * Mz + mi];
        vg_stream(out + mi, t);
    }
    float* oC = out + Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int mi = vg_div(e, iMz), k = e - mi * Mz;
        float t = 0.f;
        for (int sl = 0; sl < SC; ++sl) t = fmaf(dRs[sl * Mz + mi], Es[sl * Mz + k], t);
        vg_stream(oC + e, t);
    }
    se = vg_wave_sum(se); sv = vg_wave_sum(sv); sr = vg_wave_sum(sr);
    if ((tid & 63) == 0) { red[0][tid >> 6] = se; red[1][tid >> 6] = sv; red[2][tid >> 6] = sr; }
    __syncthreads();
    if (tid == 0) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < (int)(nt >> 6); ++k) { t0 += red[0][k]; t1 += red[1][k]; t2 += red[2][k]; }
        float* os = oC + (size_t)Mz * Mz;
        os[0] = t0; os[1] = t1; os[2] = t2; os[3] = 0.f;
        os[4] = 0.f; os[5] = 0.f; os[6] = 0.f; os[7] = 0.f;      // second set: paths_bwd_split only
    }
    VG_T(ch == 0 && l == 0 && p == 0, 503);
    VG_T(ch == a.NC - 1 && l == L - 1 && p == 0, 505);
}

// The same reverse pass on TWO workgroups per (sample chunk, latent) for launches that leave half the chip idle:
// a workgroup's time here is the ~100 KB it stages at the ~25 KB/us one CU can pull, and everything downstream is
// linear in G, so the work splits by COLUMNS of the inducing axis: half h owns columns [h Mz/2, (h+1) Mz/2) of
// A / dR / dm / dC (disjoint outputs, no extra partials) and the time points [n0, n0 + nx) of the two prior-draw dot
// products (a second set of the three scalars, added by hyper_update).  Two threads per (sample, column) halve the
// N-long chains.  Needs Mz % 8 == 0, N % 4 == 0 (16-byte rows), SK > 1.
template <int SK, int MZ = 0>      // MZ = 32: inducing extent fixed at compile time (see paths_fwd_split_body)
__global__ __launch_bounds__(kBlock) void paths_bwd_split(PathArgs a) {
    constexpr int SC = 8;
    extern __shared__ float smf[];
    __shared__ float red[3][kBlock / VG_WAVE];
    const int ch = blockIdx.x >> 1, half = blockIdx.x & 1, l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = MZ ? MZ : a.Mz, L = a.L, J = N + Mz;
    const int Mh = Mz >> 1, m0 = half * Mh;
    const int Nh = (N >> 1) & ~3, n0 = half ? Nh : 0, nx = half ? N - Nh : Nh;
    const float iMh = 1.0f / (float)Mh, iMz = 1.0f / (float)Mz;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };
    float4* A4s = reinterpret_cast<float4*>(take(4 * N * Mh));      // [N][Mh] {A, A_ell, A_var, -}
    float* Ces = take(2 * Mz * Mh);                  // [Mz][Mh] columns of (dC/dell)^T, then of (dC/dvar)^T
    float* Cvs = Ces + Mz * Mh;
    float* Gs = take(SC * N);                        // [SC][N]
    float* Rs = take(SC * Mh);                       // [SC][Mh]
    float* Es = take(SC * Mz);                       // [SC][Mz]
    float* dRs = take(SC * Mh);                      // [SC][Mh]
    float* fx = take(2 * SC * nx);                   // [SC][nx] prior draws at the time points, then their d/dell
    float* hx = fx + SC * nx;
    float* fz = take(2 * SC * Mh);                   // [SC][Mh] ... at the inducing points
    float* hz = fz + SC * Mh;
    float* rawx = take(2 * SK * SC * nx);            // the split-K slabs as they arrive
    float* rawz = take(2 * SK * SC * Mh);
    float* dGA = take(5 * SC * Mh);                  // [5][SC][Mh] G A, G A_ell, G A_var, eps C_var^T, eps C_ell^T (MFMA form)
    const int s_base = ch * SC;
    const bool dell = a.want_dell != 0;
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 500);
    {
        const float* A4g = reinterpret_cast<const float*>(a.A4 + pl * N * Mz);
        vg_stage_rows(A4s, N, 4 * Mh, tid, nt, [&](int r) -> const float* { return A4g + ((size_t)r * Mz + m0) * 4; });
        const float* Ce = a.CT_ell + pl * Mz * Mz + m0;
        const float* Cv = a.CT_var + pl * Mz * Mz + m0;
        vg_stage_rows(Ces, 2 * Mz, Mh, tid, nt, [&](int r) -> const float* {
            return r < Mz ? (dell ? Ce + (size_t)r * Mz : nullptr) : Cv + (size_t)(r - Mz) * Mz;
        });
        vg_stage_rows(Gs, SC, N, tid, nt, [&](int r) -> const float* {
            const int s = s_base + r;
            return s < S ? a.G + (((size_t)p * S + s) * L + l) * N : nullptr;              // zero beyond S
        });
        vg_stage_rows(Rs, SC, Mh, tid, nt, [&](int r) -> const float* {
            const int s = s_base + r;
            return s < S ? a.R + (((size_t)p * S + s) * L + l) * Mz + m0 : nullptr;
        });
        vg_stage_words(Es, SC * Mz, tid, nt, [&](int i) -> const void* {
            const int sl = vg_div(i, iMz), mi = i - sl * Mz, s = s_base + sl;
            return s < S ? a.eps + (((size_t)p * S + s) * Mz + mi) * L + l : nullptr;
        });
        const int nrow = (dell ? 2 : 1) * SK * SC;
        auto slab_row = [&](int r) -> const float* {
            const int second = r >= SK * SC, rr = second ? r - SK * SC : r;
            const int k = rr / SC, s = min(s_base + (rr - k * SC), S - 1);
            return (second ? a.H : a.F0) + (size_t)k * a.slab + (((size_t)p * S + s) * L + l) * J;
        };
        vg_stage_rows(rawx, nrow, nx, tid, nt, [&](int r) -> const float* { return slab_row(r) + n0; });
        vg_stage_rows(rawz, nrow, Mh, tid, nt, [&](int r) -> const float* { return slab_row(r) + N + m0; });
    }
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 506);
    vg_dma_wait();
    __syncthreads();
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 507);
    {
        const int nsx = SK * SC * nx, nsz = SK * SC * Mh;
        for (int e = tid; e < SC * nx; e += nt) {
            fx[e] = sum_slabs_lds<SK>(rawx, e, SC * nx);
            hx[e] = dell ? sum_slabs_lds<SK>(rawx + nsx, e, SC * nx) : 0.f;
        }
        for (int e = tid; e < SC * Mh; e += nt) {
            fz[e] = sum_slabs_lds<SK>(rawz, e, SC * Mh);
            hz[e] = dell ? sum_slabs_lds<SK>(rawz + nsz, e, SC * Mh) : 0.f;
        }
        __syncthreads();
    }
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 501);
    VG_STOP(a, 1);
    float se = 0.f, sv = 0.f, sr = 0.f;
    const int par = tid & 1;
    // Mz = 32 (16 columns per half): the three products G A, G A_ell, G A_var are 16 x 16 tiles (8 sample rows used) over
    // K = N on the f32 MFMA pipe, one component per wave, instead of N-long scalar chains: [3][SC][16] into LDS
    if (MZ == 32) {
        const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
        if (wv < 3) {
            vg_f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            const float* gp = Gs + min(i, SC - 1) * N;
            const float* ap = reinterpret_cast<const float*>(A4s) + wv;      // component wv of the float4 at [n][ml]
            int n = 0;
            for (; n + 8 <= N; n += 8) {
                const float a0 = i < SC ? gp[n + kk] : 0.f, a1 = i < SC ? gp[n + 4 + kk] : 0.f;
                const float b0 = ap[((n + kk) * Mh + i) * 4], b1 = ap[((n + 4 + kk) * Mh + i) * 4];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            }
            for (; n < N; n += 4) {
                const float a0 = i < SC ? gp[n + kk] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, ap[((n + kk) * Mh + i) * 4], acc0, 0, 0, 0);
            }
            // D layout: col = lane & 15 (column ml), row = (lane >> 4) * 4 + reg (sample)
            if (kk < SC / 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dGA[(wv * SC + 4 * kk + q) * Mh + i] = acc0[q] + acc1[q];
            }
        } else {
            // fourth wave: eps (dC/dvar)^T and eps (dC/dell)^T, [SC x Mz] [Mz x 16] each (the factors are triangular:
            // terms beyond the diagonal are exact zeros, no mask)
            vg_f32x4_t accv = {0.f, 0.f, 0.f, 0.f}, acce = accv;
            const float* ep = Es + min(i, SC - 1) * Mz;
#pragma unroll
            for (int k = 0; k < MZ; k += 4) {
                const float a0 = i < SC ? ep[k + kk] : 0.f;
                accv = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, Cvs[(k + kk) * Mh + i], accv, 0, 0, 0);
                acce = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, Ces[(k + kk) * Mh + i], acce, 0, 0, 0);
            }
            if (kk < SC / 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dGA[(3 * SC + 4 * kk + q) * Mh + i] = accv[q];
                    dGA[(4 * SC + 4 * kk + q) * Mh + i] = acce[q];
                }
            }
        }
        __syncthreads();
    }
    for (int it = tid >> 1; it < SC * Mh; it += nt >> 1) {      // uniform trip count for the two lanes of a pair
        const int sl = vg_div(it, iMh), ml = it - sl * Mh, mi = m0 + ml;
        const float* g = Gs + sl * N;
        float d = 0.f, de = 0.f, dv = 0.f;
        if (MZ == 32) {      // (the pair's two lanes add their halves below: the second lane contributes zero)
            d = par == 0 ? dGA[it] : 0.f;
            de = par == 0 ? dGA[SC * Mh + it] : 0.f;
            dv = par == 0 ? dGA[2 * SC * Mh + it] : 0.f;
        }
        // passes of 8 time points per lane with constant bounds (operands of a pass requested together; the tail is
        // read on clamped indices and masked)
        for (int nb = 0; MZ != 32 && nb < N; nb += 16) {
            float4 av[8];
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = min(nb + par + 2 * u, N - 1);
                av[u] = A4s[n * Mh + ml];
                gv[u] = g[n];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float gm = nb + par + 2 * u < N ? gv[u] : 0.f;
                d = fmaf(gm, av[u].x, d);
                de = fmaf(gm, av[u].y, de);
                dv = fmaf(gm, av[u].z, dv);
            }
        }
        float ue = 0.f, uv = 0.f;
        if (MZ == 32) {
            uv = par == 0 ? dGA[3 * SC * Mh + it] : 0.f;
            ue = par == 0 ? dGA[4 * SC * Mh + it] : 0.f;
        } else if (MZ) {      // (dC/dtheta)^T is upper triangular: the terms beyond the diagonal add exact zeros
#pragma unroll
            for (int k2 = 0; k2 < (MZ ? MZ / 2 : 1); ++k2) {
                const int k = par + 2 * k2;
                const float ev = k <= mi ? Es[sl * Mz + k] : 0.f;
                uv = fmaf(Cvs[k * Mh + ml], ev, uv);
                ue = fmaf(Ces[k * Mh + ml], ev, ue);
            }
        } else {
            for (int k = par; k <= mi; k += 2) {
                const float ev = Es[sl * Mz + k];
                uv = fmaf(Cvs[k * Mh + ml], ev, uv);
                ue = fmaf(Ces[k * Mh + ml], ev, ue);
            }
        }
        d += __shfl_xor(d, 1, VG_WAVE); de += __shfl_xor(de, 1, VG_WAVE); dv += __shfl_xor(dv, 1, VG_WAVE);
        ue += __shfl_xor(ue, 1, VG_WAVE); uv += __shfl_xor(uv, 1, VG_WAVE);
        if (par == 0) {
            dRs[it] = d;
            const float rv = Rs[it];
            sv += rv * dv + d * uv;
            se += rv * de + d * ue - d * hz[it];
            sr -= d * fz[it];
        }
    }
    {
        const float inx = 1.0f / (float)nx;
        for (int e = tid; e < SC * nx; e += nt) {
            const int sl = vg_div(e, inx), j = e - sl * nx;
            const float gv = Gs[sl * N + n0 + j];             // zero for samples beyond S
            sr = fmaf(gv, fx[e], sr);
            se = fmaf(gv, hx[e], se);
        }
    }
    __syncthreads();
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 502);
    VG_STOP(a, 3);
    float* out = a.part + (pl * a.NC + ch) * a.part_len;
    for (int ml = tid; ml < Mh; ml += nt) {
        float t = 0.f;
        for (int sl = 0; sl < SC; ++sl) t += dRs[sl * Mh + ml];
        vg_stream(out + m0 + ml, t);
    }
    float* oC = out + Mz;
    if (MZ == 32) {      // dC rows of this half = dR^T eps: [16 x SC] [SC x 32], two MFMA tiles (waves 0 and 1)
        const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
        if (wv < 2) {
            vg_f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < SC; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dRs[(k + kk) * Mh + i], Es[(k + kk) * Mz + 16 * wv + i], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) vg_stream(oC + (size_t)(m0 + 4 * kk + q) * Mz + 16 * wv + i, acc[q]);
        }
    } else {
        for (int e = tid; e < Mh * Mz; e += nt) {
            const int ml = vg_div(e, iMz), k = e - ml * Mz;
            float t = 0.f;
            for (int sl = 0; sl < SC; ++sl) t = fmaf(dRs[sl * Mh + ml], Es[sl * Mz + k], t);
            vg_stream(oC + (size_t)m0 * Mz + e, t);
        }
    }
    se = vg_wave_sum(se); sv = vg_wave_sum(sv); sr = vg_wave_sum(sr);
    if ((tid & 63) == 0) { red[0][tid >> 6] = se; red[1][tid >> 6] = sv; red[2][tid >> 6] = sr; }
    __syncthreads();
    if (tid == 0) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < (int)(nt >> 6); ++k) { t0 += red[0][k]; t1 += red[1][k]; t2 += red[2][k]; }
        float* os = oC + (size_t)Mz * Mz + 4 * half;
        os[0] = t0; os[1] = t1; os[2] = t2; os[3] = 0.f;
    }
    VG_T(ch == 0 && half == 0 && l == 0 && p == 0, 503);
    VG_T(ch == a.NC - 1 && half == 1 && l == L - 1 && p == 0, 505);
}

// =================================================================================================
// Gradient assembly + Adam, in two launches so that the next step can start early:
//   lengthscale / variance of every latent (a handful of scalars): the covariance and feature kernels of
//                    the NEXT step depend only on these, so the reverse pass itself updates them (HyperArgs);
//   final_kernel  -- q_mu / q_sqrt of one (latent, problem) per workgroup, and the ELBO pieces.
// =================================================================================================
struct FinalArgs {
    int M, L, NC, nblk;
    size_t part_len;
    const float *part, *Lk32, *lik_partial;
    const double *gkl_qmu, *gkl_Q, *kl_l;
    double kl_scale, lik_scale;
    const double* alpha_fin;  // [P] per-problem alpha / S (trainable likelihood constants), else lik_scale
    double *out_lik, *out_kl;
    double *g_qmu, *g_qsqrt;
    int do_adam, trainable;
    int dma;                  // the chunk partials fit in LDS: stage them by DMA
    const double* lr_dev;     // [1] step size stored at the counter tick (device counter form)
    double lr_t;              // host form
    int use_lr_dev;
    double *mq_mu, *mq_sqrt;  // Adam moments
    double *vq_mu, *vq_sqrt;
    double *pq_mu, *pq_sqrt;  // parameters (updated in place)
    int stop;
};

// Hyper-parameter update of one (problem, latent): gradient of the loss wrt (raw lengthscale, raw variance) from the
// reverse-pass sums and the KL tangents, chain rule through the softplus, Adam.  Two forms with identical arithmetic:
// hyper_kernel (its own launch) and a PROLOGUE of the stage-1 roles that need the new values (small batches, steps
// after the first of a call): every workgroup of the latent repeats the ~100 operations, only the cov_a role stores
// (to a staging row that role 0 of stage 2 copies to the parameter / Adam tensors, which nobody reads in between).
// Everything slow is prepared earlier: the step size at the counter tick, var and the softplus slopes by cov_a.
// (A ticket scheme that let the last workgroup of the reverse pass do the update was measured and rejected: with
// __threadfence() the agent-scope fences cost ~16 us on this 8-XCD part, with atomics only it is a wash.)
struct HyperArgs {
    int L, Mz, NC, want_dell;
    size_t part_len;
    const float* part;
    const double *gkl_ell, *gkl_var, *var, *sig_ell, *sig_var;      // var / slopes of the step being finished
    double kl_scale, lr_t;
    const double* lr_dev;    // [1] step size stored at the counter tick (device counter form), else lr_t
    const uint32_t* ctr;     // hyper_kernel only: derive the step size from the (ticked) counter and store it
    double lr;
    double* lr_store;
    double *g_ell, *g_var;
    double *m_ell, *m_var, *v_ell, *v_var, *p_ell, *p_var;
    double* next;            // [P,L,6] staging of {raw_ell, raw_var, m_ell, v_ell, m_var, v_var} (prologue form)
    int do_adam, trainable, use_lr_dev;
};

struct HyperState { double raw_ell, raw_var, m_ell, v_ell, m_var, v_var, g_ell, g_var; };

// The three sums over the sample chunks are loaded in ONE round (16 chunks x 3 values per pass, clamped + masked)
// and added in the order of sum_chunks().
// `which`: 1 = lengthscale, 2 = variance, 3 = both (the two halves are independent: cov_a runs them on two waves)
__device__ __forceinline__ HyperState hyper_update(const HyperArgs& h, size_t pl, bool own_lr = false, double lr_own = 0.0,
                                                   int which = 3) {
    const float* part = h.part + pl * h.NC * h.part_len + (h.Mz + h.Mz * h.Mz);
    // every operand requested in one go, unconditionally (null Adam pointers fall back to a valid address): the
    // prologue form sits on the critical chain and a second dependent round trip costs ~2 us
    const double* mell = h.do_adam ? h.m_ell : h.p_ell;
    const double* vell = h.do_adam ? h.v_ell : h.p_ell;
    const double* mvar = h.do_adam ? h.m_var : h.p_var;
    const double* vvar = h.do_adam ? h.v_var : h.p_var;
    float v0[16][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float* q = part + (size_t)min(k, h.NC - 1) * h.part_len;
        v0[k][0] = q[0] + q[4]; v0[k][1] = q[1] + q[5]; v0[k][2] = q[2] + q[6];      // the two halves of paths_bwd_split
    }
    HyperState o;
    o.raw_ell = h.p_ell[pl]; o.raw_var = h.p_var[pl];
    o.m_ell = mell[pl]; o.v_ell = vell[pl]; o.m_var = mvar[pl]; o.v_var = vvar[pl];
    const double gkl_ell = h.gkl_ell[pl], gkl_var = h.gkl_var[pl], var = h.var[pl];
    const double sig_ell = h.sig_ell[pl], sig_var = h.sig_var[pl];
    const double lr_dev = h.lr_dev[0];
    const double lr_t = own_lr ? lr_own : ((h.do_adam && h.use_lr_dev) ? lr_dev : h.lr_t);
    if (!h.do_adam) o.m_ell = o.v_ell = o.m_var = o.v_var = 0.0;
    double s3[3] = {0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < h.NC; c0 += 16) {
        float v[16][3];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (c0 == 0) { v[k][0] = v0[k][0]; v[k][1] = v0[k][1]; v[k][2] = v0[k][2]; continue; }
            const float* q = part + (size_t)min(c0 + k, h.NC - 1) * h.part_len;
            v[k][0] = q[0] + q[4]; v[k][1] = q[1] + q[5]; v[k][2] = q[2] + q[6];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!(which & (j == 0 ? 1 : 2))) continue;
            double d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = c0 + k < h.NC ? (double)v[k][j] : 0.0;
            s3[j] += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
            s3[j] += ((d[8] + d[9]) + (d[10] + d[11])) + ((d[12] + d[13]) + (d[14] + d[15]));
        }
    }
    o.g_ell = o.g_var = 0.0;
    if (which & 1) {
        const double s_ell = h.want_dell ? s3[0] : 0.0;
        o.g_ell = (s_ell + h.kl_scale * gkl_ell) * sig_ell;
        if (h.do_adam && (h.trainable & VGPMP_TRAIN_LENGTHSCALES)) adam_update(&o.raw_ell, &o.m_ell, &o.v_ell, o.g_ell, lr_t);
    }
    if (which & 2) {
        o.g_var = (s3[1] + s3[2] / (2.0 * var) + h.kl_scale * gkl_var) * sig_var;
        if (h.do_adam && (h.trainable & VGPMP_TRAIN_KERNEL_VARIANCE)) adam_update(&o.raw_var, &o.m_var, &o.v_var, o.g_var, lr_t);
    }
    return o;
}

// =================================================================================================
// Covariance path (float64).
//
// cov_fwd_kernel -- one workgroup per (latent, problem): Kuu, chol, inverse, q_sqrt, KL and its
//   gradient, plus the FORWARD-MODE tangents of chol/q_sqrt/KL wrt the latent's two kernel
//   hyper-parameters (lengthscale, variance).
// cov_rows_kernel -- row tiles of A = Kfu (Kuu + jI)^-1 and of its two tangents, spread over
//   N/8 workgroups per latent.
// With the tangents available the sample-dependent reverse pass needs only dot products of its
// upstream gradients with them -- no Cholesky adjoint, no N-sized float64 reductions -- and both
// kernels sit off the critical path (side stream) next to the noise/feature/GEMM branch.
// =================================================================================================
struct CovArgs {
    int N, M, L, D;
    const double *X, *Zy, *y_u;
    size_t zy_stride;        // doubles between the Zy of consecutive problems (0: one shared set)
    double jitter;
    const double *q_mu, *q_sqrt, *raw_ell, *raw_var;
    int want_dell;
    int stop;
    int elim_wave;           // Mz <= 32: the elimination on one wave (chol_inverse_wave)
    uint32_t* tick;          // device step counter, ticked by one row-tile workgroup of stage 2 (or null)
    double lr;               // with the tick: the step size of this step's update goes to lr_dev[0]
    double* lr_dev;
    // hyper-parameter update of the previous step as a prologue (stage 1) / its commit (stage 2, role 0)
    int prologue, commit, keep_prev;
    HyperArgs hy;
    vg_workspace ws;
};

constexpr int kCovThreads = 256;      // == kBlock: the covariance roles share launches with other kernels
constexpr int kRowTile = 8;

// ---- float64 matrix-core tiles ---------------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] (one
// double each); it receives D[row = (l >> 4) + 4 q][col = l & 15] in accumulator element q = 0..3.
// All matrices live in LDS with dimension Mp = roundup(Mz, 16) (zero padded), so no edge handling.
typedef double vg_f64x4 __attribute__((ext_vector_type(4)));

struct MatView {            // element (r, c) at p[r * sr + c * sc]
    const double* p;
    int sr, sc;
};

__device__ __forceinline__ vg_f64x4 mfma_tile_f64(MatView A, MatView B, int K, int lane, int i0, int j0) {
    vg_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int r = lane & 15, g = lane >> 4;
    const double* ap = A.p + (i0 + r) * A.sr + g * A.sc;
    const double* bp = B.p + g * B.sr + (j0 + r) * B.sc;
    // K is a multiple of 16: passes of four k-steps with constant bounds, so that a pass's eight operands are
    // requested together and its products chain in the accumulator registers (a loop with a run-time trip count is
    // left rolled by the compiler: load, wait, move the accumulator in, multiply, move it out -- 3x slower)
    auto pass = [&](int k0) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = ap[(k0 + 4 * u) * A.sc]; bv[u] = bp[(k0 + 4 * u) * B.sr]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    };
    if (K == 32) { pass(0); pass(16); }
    else for (int k0 = 0; k0 < K; k0 += 16) pass(k0);
    return acc;
}

// D = A B over all 16x16 tiles of an Mp x Mp result, tiles dealt round-robin to the waves; `emit(r, c, v)`
// receives every element.
template <typename Emit>
__device__ __forceinline__ void matmul_f64(MatView A, MatView B, int Mp, int tid, int nt, Emit emit) {
    const int lane = tid & 63, nT = Mp >> 4;
    for (int t = tid >> 6; t < nT * nT; t += nt >> 6) {
        const int i0 = (t / nT) << 4, j0 = (t % nT) << 4;
        const vg_f64x4 acc = mfma_tile_f64(A, B, Mp, lane, i0, j0);
#pragma unroll
        for (int q = 0; q < 4; ++q) emit(i0 + (lane >> 4) + 4 * q, j0 + (lane & 15), acc[q]);
    }
}

// Cholesky factor and its inverse of the SPD matrix held in La (LDS), by forward elimination of the
// augmented matrix [K | I] without pivoting (K = L~ D L~^T): after Mz pivots the left half holds
// U = D L~^T and the right half L~^-1, so  Lk = L~ D^1/2  and  Lk^-1 = D^-1/2 L~^-1.  Every pivot is
// one rank-1 update spread over the whole workgroup and ONE barrier (any Mz; chol_inverse_regs below is the
// faster form for Mz <= 32).
__device__ __forceinline__ void chol_inverse_block(double* La, double* Li, double* Aug, double* rsd, int Mz, int ld,
                                                   int tid, int nt) {
    const int la = 2 * Mz + 1;
    const float iMz = 1.0f / (float)Mz, i2Mz = 0.5f / (float)Mz;
    for (int e = tid; e < Mz * 2 * Mz; e += nt) {
        const int i = vg_div(e, i2Mz), j = e - i * 2 * Mz;
        Aug[i * la + j] = j < Mz ? La[i * ld + j] : (j - Mz == i ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = 0; k < Mz; ++k) {
        const double r = 1.0 / Aug[k * la + k];
        const int h = Mz - k - 1;                   // rows k+1 .. Mz-1, columns k+1 .. Mz+k
        for (int e = tid; e < h * Mz; e += nt) {
            const int qi = vg_div(e, iMz);
            const int i = k + 1 + qi, j = k + 1 + (e - qi * Mz);
            Aug[i * la + j] = fma(-(Aug[i * la + k] * r), Aug[k * la + j], Aug[i * la + j]);
        }
        __syncthreads();
    }
    for (int k = tid; k < Mz; k += nt) rsd[k] = rsqrt(Aug[k * la + k]);
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int i = vg_div(e, iMz), j = e - i * Mz;
        La[i * ld + j] = j <= i ? Aug[j * la + i] * rsd[j] : 0.0;
        Li[i * ld + j] = j <= i ? Aug[i * la + Mz + j] * rsd[i] : 0.0;
    }
    __syncthreads();
}

// The same elimination with the augmented matrix in REGISTERS (Mz <= 32, 256 threads): thread (row i = tid & 31,
// column block jb = tid >> 5) keeps columns [8 jb, 8 jb + 8) of [K | I] laid out as 32 + 32 columns.  Per pivot
// the owners publish the pivot row and the pivot column through LDS (double buffered: one barrier per pivot),
// everyone reads its 8 + 2 values in one LDS round and does 8 FMAs out of registers; measured 340 ns per pivot
// against 420 ns for the LDS-resident loop above (three LDS reads and a write per element).
__device__ __forceinline__ void chol_inverse_regs(double* La, double* Li, double* Aug, double* rsd, int Mz, int ld,
                                                  int tid, int nt) {
    const int i = tid & 31, jb = tid >> 5, la = 2 * Mz + 1;
    const float iMz = 1.0f / (float)Mz;
    double* prow = Aug;                  // [2][64] pivot row, both halves
    double* pcol = Aug + 128;            // [2][32] pivot column
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = 8 * jb + c;      // < 32: column of K;  >= 32: column col - 32 of I
        a[c] = i < Mz ? (col < 32 ? (col < Mz ? La[i * ld + col] : 0.0) : (col - 32 == i ? 1.0 : 0.0)) : 0.0;
    }
    __syncthreads();
#pragma nounroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = 8 * kb + c;
            if (k >= Mz) break;
            const int buf = k & 1;
            if (i == k) {
#pragma unroll
                for (int q = 0; q < 8; ++q) prow[buf * 64 + 8 * jb + q] = a[q];
            }
            if (jb == kb) pcol[buf * 32 + i] = a[c];
            __syncthreads();
            // one LDS round for everything this thread needs of pivot k (read unconditionally, used conditionally)
            const double piv = pcol[buf * 32 + k], aik = pcol[buf * 32 + i];
            double pr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) pr[q] = prow[buf * 64 + 8 * jb + q];
            if (tid == 0) rsd[k] = piv;          // rsqrt after the loop, off the chain
            // 1 / piv sits on the dependency chain of every pivot: hardware estimate + two Newton steps (to the
            // last bit or two) instead of the ~10-instruction IEEE division sequence
            double r = __builtin_amdgcn_rcp(piv);
            r = fma(fma(-piv, r, 1.0), r, r);
            r = fma(fma(-piv, r, 1.0), r, r);
            const double m = (i > k && i < Mz) ? aik * r : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fma(-m, pr[q], a[q]);
        }
    }
    __syncthreads();
    if (tid < Mz) rsd[tid] = rsqrt(rsd[tid]);
    __syncthreads();
    // registers -> the [Mz][2 Mz + 1] image the tail expects: left half U = D L~^T, right half L~^-1
    double* Img = Aug;                   // the exchange buffers are dead now
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = 8 * jb + c;
        if (i < Mz) {
            if (col < Mz) Img[i * la + col] = a[c];
            else if (col >= 32 && col - 32 < Mz) Img[i * la + Mz + (col - 32)] = a[c];
        }
    }
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int r = vg_div(e, iMz), j = e - r * Mz;
        La[r * ld + j] = j <= r ? Img[j * la + r] * rsd[j] : 0.0;
        Li[r * ld + j] = j <= r ? Img[r * la + Mz + j] * rsd[r] : 0.0;
    }
    __syncthreads();
}

// The same elimination on ONE wave without LDS or barriers in the loop (Mz <= 32): lane c keeps column c of
// [K | I] (32 + 32 columns, 32 rows = 64 registers); the pivot and the pivot column reach the other lanes as
// scalar broadcasts (v_readlane), the loops are fully unrolled so that every row index is a register name.  What
// is left of a pivot's cost is its dependency chain (reciprocal + two Newton steps + the update of the next pivot),
// the trailing updates of the previous pivot fill its gaps.  (The multiplier is applied as a_ik (row_k / d_k) instead
// of (a_ik / d_k) row_k: one independent FMA per row; results differ from the forms above in the last bit.)
__device__ __forceinline__ double vg_bcast_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void chol_inverse_wave(double* La, double* Li, double* Aug, double* rsd, int Mz, int ld,
                                                  int tid, int nt) {
    const int la = 2 * Mz + 1;
    const float iMz = 1.0f / (float)Mz;
    double* Img = Aug;
    if (tid < VG_WAVE) {
        const int c = tid;
        double a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double kv = La[min(i, Mz - 1) * ld + min(c & 31, Mz - 1)];      // loads first, selects afterwards
            const double id = (i == (c & 31)) ? 1.0 : 0.0;
            a[i] = c < 32 ? ((i < Mz && c < Mz) ? kv : id) : id;
        }
        // software pipelined: pivot k first finishes row k + 1 -- the next pivot row -- so that the next reciprocal
        // (the long dependent chain) is in flight while the remaining rows of pivot k are updated
        double piv = vg_bcast_f64(a[0], 0);
        double r = __builtin_amdgcn_rcp(piv);
        r = fma(fma(-piv, r, 1.0), r, r);
        r = fma(fma(-piv, r, 1.0), r, r);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            double rn = 0.0;
            // row_i -= (a_ik / d_k) row_k as ONE product per row: the scalar a_ik times w = row_k / d_k (per lane)
            const double w = -a[k] * r;
            if (k + 1 < 32) {
                a[k + 1] = fma(vg_bcast_f64(a[k + 1], k), w, a[k + 1]);
                const double pn = vg_bcast_f64(a[k + 1], k + 1);
                rn = __builtin_amdgcn_rcp(pn);
                rn = fma(fma(-pn, rn, 1.0), rn, rn);
                rn = fma(fma(-pn, rn, 1.0), rn, rn);
            }
#pragma unroll
            for (int i = k + 2; i < 32; ++i) a[i] = fma(vg_bcast_f64(a[i], k), w, a[i]);
            r = rn;
        }
        // pivot k is the diagonal entry lane k ends with (row k is final once pivot k - 1 is done): rsqrt after the
        // loop, off the chain
        double mine = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) mine = c == i ? a[i] : mine;
        if (c < Mz) rsd[c] = mine;
        // registers -> the [Mz][2 Mz + 1] image the tail expects: left half U = D L~^T, right half L~^-1
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < Mz) {
                if (c < Mz) Img[i * la + c] = a[i];
                else if (c >= 32 && c - 32 < Mz) Img[i * la + Mz + (c - 32)] = a[i];
            }
        }
    }
    __syncthreads();
    if (tid < Mz) rsd[tid] = rsqrt(rsd[tid]);
    __syncthreads();
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int r = vg_div(e, iMz), j = e - r * Mz;
        La[r * ld + j] = j <= r ? Img[j * la + r] * rsd[j] : 0.0;
        Li[r * ld + j] = j <= r ? Img[r * la + Mz + j] * rsd[r] : 0.0;
    }
    __syncthreads();
}

// ---- stage A: Kuu, factorisation, inverse --------------------------------------------------------
__device__ __forceinline__ void cov_a_body(const CovArgs& a, double* sm, int l, int p) {
    __shared__ double scal[2];
    const int tid = threadIdx.x, nt = blockDim.x;
    VG_T(l == 0 && p == 0, 100);
    const int M = a.M, Mz = M + 2, L = a.L, D = a.D;
    const int Mp = (Mz + 15) & ~15, ld = Mp + 1;
    const float iMz = 1.0f / (float)Mz;
    const size_t pl = (size_t)p * L + l;
    double* La = sm;                 // Kuu + jI -> Cholesky factor Lk      (Mp x ld, zero padded)
    double* Li = La + Mp * ld;       // Lk^-1
    double* Sc = Li + Mp * ld;       // 2 x (Mp x ld) scratch: augmented matrix of the elimination
    double* zs = Sc + 2 * Mp * ld;   // [Mp]
    double* rsd = zs + Mp;           // [Mp]
    // the latent's two scalars, each a chain of float64 exp / log / sqrt / division (~1 us): lengthscale on the first
    // lane of wave 0, variance on the first lane of wave 1 (different waves run side by side, lanes of one do not)
    if (tid == 0 || tid == VG_WAVE) {
        const bool is_ell = tid == 0;
        double raw;
        if (a.prologue) {
            const HyperState o = hyper_update(a.hy, pl, false, 0.0, is_ell ? 1 : 2);
            double* nx = a.hy.next + 6 * pl;
            if (is_ell) { a.hy.g_ell[pl] = o.g_ell; nx[0] = o.raw_ell; nx[2] = o.m_ell; nx[3] = o.v_ell; raw = o.raw_ell; }
            else { a.hy.g_var[pl] = o.g_var; nx[1] = o.raw_var; nx[4] = o.m_var; nx[5] = o.v_var; raw = o.raw_var; }
        } else {
            raw = is_ell ? a.raw_ell[pl] : a.raw_var[pl];
        }
        if (is_ell) { scal[0] = softplus_d(raw); a.ws.sig_ell[pl] = sigmoid_d(raw); }
        else { scal[1] = kVarFloor + softplus_d(raw); a.ws.sig_var[pl] = sigmoid_d(raw); }
    }
    for (int e = tid; e < 2 * Mp * ld; e += nt) sm[e] = 0.0;
    for (int i = tid; i < Mz; i += nt) zs[i] = a.Zy[(size_t)p * a.zy_stride + (size_t)i * D + l];
    __syncthreads();
    const double ell = scal[0], var = scal[1], jit = a.jitter;
    if (tid == 0) { a.ws.ell[pl] = ell; a.ws.var[pl] = var; }
    // Kuu and dKuu/dell share the exponential; symmetric: evaluate the lower triangle only
    double* Kg = a.ws.Ks64 + pl * Mz * Mz;
    double* Kdg = a.ws.Kd_ell + pl * Mz * Mz;
    // lower triangle only, in triangular order (e -> row i, column j <= i): Mz (Mz + 1) / 2 evaluations of the exponential
    // spread evenly over the workgroup; one division per thread instead of two per element
    const double inv_ell = 1.0 / ell, c3 = 5.0 / (3.0 * ell);
    // (the diagonal needs no exponential: Mz (Mz - 1) / 2 = 496 evaluations at Mz = 32 are two rounds of the
    // workgroup, with the diagonal among them it was three)
    for (int e = tid; e < Mz * (Mz - 1) / 2; e += nt) {
        int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        if (i * (i + 1) / 2 > e) --i;                    // float rounding at the row boundaries
        if ((i + 1) * (i + 2) / 2 <= e) ++i;
        const int j = e - i * (i + 1) / 2;               // strictly lower entry (i + 1, j)
        ++i;
        double r = fabs(zs[i] - zs[j]) * inv_ell;
        double ex = exp(-kSqrt5 * r);
        double k = var * (1.0 + kSqrt5 * r + (5.0 / 3.0) * r * r) * ex;
        double dk = var * ex * (r * r * c3) * (1.0 + kSqrt5 * r);
        La[i * ld + j] = k; La[j * ld + i] = k;
        Kg[(size_t)i * Mz + j] = k; Kg[(size_t)j * Mz + i] = k;
        Kdg[(size_t)i * Mz + j] = dk; Kdg[(size_t)j * Mz + i] = dk;
    }
    for (int i = tid; i < Mz; i += nt) {                 // r = 0: k = var exp(-0) = var, dk = 0
        La[i * ld + i] = var + jit;
        Kg[(size_t)i * Mz + i] = var;
        Kdg[(size_t)i * Mz + i] = 0.0;
    }
    __syncthreads();
    VG_T(l == 0 && p == 0, 101);
    if (Mz <= 32 && a.elim_wave) chol_inverse_wave(La, Li, Sc, rsd, Mz, ld, tid, nt);
    else if (Mz <= 32 && nt == 256) chol_inverse_regs(La, Li, Sc, rsd, Mz, ld, tid, nt);
    else chol_inverse_block(La, Li, Sc, rsd, Mz, ld, tid, nt);
    VG_T(l == 0 && p == 0, 102);
    double* Kig = a.ws.Kinv + pl * Mz * Mz;
    matmul_f64(MatView{Li, 1, ld}, MatView{Li, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
        if (r < Mz && c < Mz) Kig[(size_t)r * Mz + c] = v;
    });
    double* Lkg = a.ws.Lk64 + pl * Mz * Mz;
    double* Lig = a.ws.Li64 + pl * Mz * Mz;
    for (int e = tid; e < Mz * Mz; e += nt) {
        const int i = vg_div(e, iMz), j = e - i * Mz;
        Lkg[e] = La[i * ld + j];
        Lig[e] = Li[i * ld + j];
    }
    VG_T(l == 0 && p == 0, 103);
}

__global__ __launch_bounds__(kCovThreads) void cov_a_kernel(CovArgs a) {
    extern __shared__ double sm[];
    cov_a_body(a, sm, blockIdx.x, blockIdx.y);
}

// ---- stage B: heterogeneous launch, role = blockIdx.x ---------------------------------------------
//   0            q_sqrt = Lk pad(Q) + jitter, q_mu, KL and its gradient wrt q_mu / q_sqrt
//   1, 2         forward-mode tangent wrt lengthscale / variance:  dC = (Lk Phi(Lk^-1 dK Lk^-T)) pad(Q), dKL
//   3 + t        row tile t of A = Kfu (Kuu + jI)^-1 and its tangents (cov_rows_body)
__device__ void cov_rows_body(const CovArgs& a, double* sm, int tile, int l, int p, int tid, int nt);

template <bool TANGENTS>
__device__ __forceinline__ void cov_b_body(const CovArgs& a, double* sm, int role, int l, int p) {
    __shared__ double red[kCovThreads / VG_WAVE];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (role >= 3) {
        // The step counter ticks where no kernel that reads it runs alongside: noise drawn before this launch
        // saw the old value, the noise of the next step and the Adam count see the new one.
        if (a.tick && role == 3 && l == 0 && p == 0 && tid == 0) {
            const uint32_t t = *a.tick + 1u;          // = 1-based Adam count of this step's update
            *a.tick = t;
            a.lr_dev[0] = adam_step_size(a.lr, (double)t);
        }
        cov_rows_body(a, sm, role - 3, l, p, tid, nt);
        return;
    }
    if (role > 0 && (!TANGENTS || (role == 1 && !a.want_dell))) return;
    VG_T(l == 0 && p == 0, 200 + 10 * role);
    const int M = a.M, Mz = M + 2, L = a.L;
    const int Mp = (Mz + 15) & ~15, ld = Mp + 2;      // even: LDS rows start on 16 bytes
    const float iMz = 1.0f / (float)Mz, iM = 1.0f / (float)M;
    const size_t pl = (size_t)p * L + l;
    double* La = sm;                 // Lk, later pad(q_sqrt) for the tangents   (all Mp x ld, zero padded)
    double* Li = La + Mp * ld;       // Lk^-1
    double* X1 = Li + Mp * ld;       // role 0: pad(q_sqrt), Q at [2:, 2:];  tangents: dK/dtheta, then W
    double* X2 = X1 + Mp * ld;       // tangents: scratch T
    double* dl = X2 + Mp * ld;       // [Mp] q_mu - p_mu
    double* af = dl + Mp;            // [Mp] Lk^-1 (q_mu - p_mu)
    double* v1 = af + Mp;            // [Mp]
    double* k0 = v1 + Mp;            // [Mp] first two columns of Kuu + jI
    double* k1 = k0 + Mp;
    double* kd0 = k1 + Mp;           // [Mp] first two columns of dK/dtheta
    double* kd1 = kd0 + Mp;
    double* Qp = X1;
    double* Kd = X1;
    double* T = X2;
    double* qm = kd1 + Mp;           // [Mp] q_mu behind the two conditioned points
    const double jit = a.jitter, var = a.ws.var[pl];
    const double y0 = a.y_u[((size_t)p * 2 + 0) * L + l], y1 = a.y_u[((size_t)p * 2 + 1) * L + l];
    const double* Kg = a.ws.Ks64 + pl * Mz * Mz;
    constexpr int kQRegs = (VGPMP_MAX_MZ - 2) * (VGPMP_MAX_MZ - 2) / kCovThreads + 1;
    double qreg[kQRegs];
    {
        // every operand by DMA, all requests in flight together (zero padding written directly)
        const double* Qg = a.q_sqrt + pl * M * M;
        auto all = [](int, int) { return true; };
        const bool square = Mz == Mp;      // no zero padding needed: whole rows in 16-byte units
        if (square) {
            vg_stage_f64_square(La, ld, a.ws.Lk64 + pl * Mz * Mz, Mz, tid, nt);
            vg_stage_f64_square(Li, ld, a.ws.Li64 + pl * Mz * Mz, Mz, tid, nt);
        } else {
            vg_stage_f64(La, Mp, ld, a.ws.Lk64 + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
            vg_stage_f64(Li, Mp, ld, a.ws.Li64 + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
        }
        if (role == 0) {
            vg_stage_f64(Qp, Mp, ld, Qg, M, M, 2, 2, tid, nt, [](int r, int c) { return c <= r; });
        } else {      // tangents: this thread's share of Q waits in registers until Lk's LDS space is free
            const double* Kdg = role == 1 ? a.ws.Kd_ell + pl * Mz * Mz : Kg;
            if (square) vg_stage_f64_square(Kd, ld, Kdg, Mz, tid, nt);
            else vg_stage_f64(Kd, Mp, ld, Kdg, Mz, Mz, 0, 0, tid, nt, all);
#pragma unroll
            for (int k = 0; k < kQRegs; ++k) qreg[k] = Qg[min(tid + k * nt, M * M - 1)];
        }
        // k0 | k1: the first two columns of Kuu;  qm: q_mu at [2:]
        vg_stage_words(k0, 4 * Mp, tid, nt, [&](int w) -> const void* {
            const int d = w >> 1, col = d >= Mp, i = d - col * Mp;
            return i < Mz ? reinterpret_cast<const uint32_t*>(Kg + (size_t)i * Mz + col) + (w & 1) : nullptr;
        });
        vg_stage_words(qm, 2 * Mp, tid, nt, [&](int w) -> const void* {
            const int i = w >> 1;
            return (i >= 2 && i < Mz) ? reinterpret_cast<const uint32_t*>(a.q_mu + pl * M + (i - 2)) + (w & 1) : nullptr;
        });
    }
    if (role == 0 && tid == 0) {      // behind the staging requests: these round trips overlap them
        if (a.commit && a.hy.do_adam) {      // staged hyper-parameters of the prologue -> their tensors
            const HyperArgs& h = a.hy;
            const double* nx = h.next + 6 * pl;
            h.p_ell[pl] = nx[0]; h.p_var[pl] = nx[1]; h.m_ell[pl] = nx[2]; h.v_ell[pl] = nx[3]; h.m_var[pl] = nx[4];
            h.v_var[pl] = nx[5];
        }
        if (a.keep_prev) {                   // this step's var / slopes for the prologue of the next step
            a.ws.prev_var[pl] = a.ws.var[pl];
            a.ws.prev_sig_ell[pl] = a.ws.sig_ell[pl];
            a.ws.prev_sig_var[pl] = a.ws.sig_var[pl];
        }
    }
    vg_dma_wait();
    __syncthreads();
    // prior mean through the two conditioned points and a = Lk^-1 (q_mu - p_mu)  (prior_kl.py:16-35); the jitter on
    // the two leading diagonal entries and the conditioned values are applied on the fly (no fix-up pass, no barrier)
    const double k00 = k0[0] + jit, k01 = k1[0], k11 = k1[1] + jit;
    const double det = k00 * k11 - k01 * k01;
    const double c0 = (k11 * y0 - k01 * y1) / det, c1 = (k00 * y1 - k01 * y0) / det;
    // (loads first, selects afterwards: a conditional load is a branch)
    auto K0 = [&](int i) { const double v = k0[i]; return i == 0 ? k00 : v; };
    auto K1 = [&](int i) { const double v = k1[i]; return i == 1 ? k11 : v; };
    for (int i = tid; i < Mz; i += nt) {
        const double qi = qm[i];
        const double mi = i == 0 ? y0 : (i == 1 ? y1 : qi);
        if (role == 0) a.ws.m[pl * Mz + i] = (float)mi;
        dl[i] = mi - (K0(i) * c0 + K1(i) * c1);
    }
    const double kd_scale = role == 2 ? 1.0 / var : 1.0;      // dK/dvar = K / var, applied to the products
    if (role == 0) {                         // float32 copy for the gradient assembly (written here, not in stage A:
        float* Lk32 = a.ws.Lk32 + pl * Mz * Mz;      // stage A of the next step may overlap that kernel)
        for (int e = tid; e < Mz * Mz; e += nt) {
            const int i = vg_div(e, iMz), j = e - i * Mz;
            Lk32[e] = (float)La[i * ld + j];
        }
    }
    __syncthreads();
    VG_T(l == 0 && p == 0, 201 + 10 * role);
    double klacc = 0.0;
    const int sub = tid & 7;
    for (int i = tid >> 3; i < Mz; i += nt >> 3) {      // af is read again only behind later barriers
        const double s = dot8(Li + i * ld, 1, dl, 1, i + 1, sub);
        if (sub == 0) {
            af[i] = s;
            if (i >= 2) klacc += s * s;
        }
    }
    if (role == 0) {
        float* C32 = a.ws.C + pl * Mz * Mz;
        float* C32T = a.ws.CT + pl * Mz * Mz;
        matmul_f64(MatView{La, ld, 1}, MatView{Qp, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
            if (r < Mz && c < Mz) {
                const float cv = (float)(v + (r == c && r < 2 ? jit : 0.0));
                C32[(size_t)r * Mz + c] = cv;
                C32T[(size_t)c * Mz + r] = cv;
            }
        });
        double* gklQ = a.ws.gkl_Q + pl * M * M;
        for (int e = tid; e < M * M; e += nt) {
            int r = vg_div(e, iM), c = e - r * M;
            double gq = 0.0;
            if (c <= r) {
                double q = Qp[(r + 2) * ld + (c + 2)];
                klacc += q * q;
                gq = q;
                if (c == r) { klacc -= log(q * q); gq -= 1.0 / q; }
            }
            gklQ[e] = gq;
        }
        const double kl = block_sum(klacc, red);
        if (tid == 0) a.ws.kl_l[pl] = 0.5 * (kl - (double)M);
        // d KL / d q_mu = (Lk^-T [0, 0, a])[2:]
        for (int k = (tid >> 3) + 2; k < Mz; k += nt >> 3) {
            const double g = dot8(Li + k * ld + k, ld, af + k, 1, Mz - k, sub);
            if (sub == 0) a.ws.gkl_qmu[pl * M + (k - 2)] = g;
        }
        VG_T(l == 0 && p == 0, 202);
        return;
    }
    // ---- tangent wrt theta: W = Phi(Lk^-1 dK Lk^-T), dLk = Lk W, dC = dLk pad(Q)   (64-bit MFMA products)
    // Four LDS matrices (35 KB at Mz = 32, so that these workgroups pack 4 per CU next to the prior GEMM):
    // W overwrites dK (its first two columns are kept), dLk overwrites T, and pad(Q) -- prefetched into
    // registers -- takes the place of Lk once Lk has been used.
    for (int i = tid; i < Mz; i += nt) { kd0[i] = Kd[i * ld + 0] * kd_scale; kd1[i] = Kd[i * ld + 1] * kd_scale; }
    VG_T(l == 0 && p == 0, 204 + 10 * role);
    matmul_f64(MatView{Li, ld, 1}, MatView{Kd, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) { T[r * ld + c] = v * kd_scale; });
    __syncthreads();
    VG_T(l == 0 && p == 0, 205 + 10 * role);
    double* W = X1;
    matmul_f64(MatView{T, ld, 1}, MatView{Li, 1, ld}, Mp, tid, nt, [&](int r, int c, double v) {
        W[r * ld + c] = c < r ? v : (c == r ? 0.5 * v : 0.0);
    });
    __syncthreads();
    VG_T(l == 0 && p == 0, 206 + 10 * role);
    matmul_f64(MatView{La, ld, 1}, MatView{W, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) { T[r * ld + c] = v; });
    __syncthreads();
    VG_T(l == 0 && p == 0, 207 + 10 * role);
    for (int e = tid; e < Mp * ld; e += nt) La[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kQRegs; ++k) {
        const int e = tid + k * nt;
        if (e < M * M) {
            const int r = vg_div(e, iM), c = e - r * M;
            if (c <= r) La[(r + 2) * ld + (c + 2)] = qreg[k];
        }
    }
    __syncthreads();
    float* CT = (role == 1 ? a.ws.CT_ell : a.ws.CT_var) + pl * Mz * Mz;
    matmul_f64(MatView{T, ld, 1}, MatView{La, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
        if (r < Mz && c < Mz) CT[(size_t)c * Mz + r] = (float)v;       // stored transposed
    });
    VG_T(l == 0 && p == 0, 202 + 10 * role);
    // KL tangent: a_dot = Lk^-1 (delta_dot - dLk a),  delta_dot = -d p_mu
    const double d00 = kd0[0], d01 = kd1[0], d11 = kd1[1];
    const double e0 = d00 * c0 + d01 * c1, e1 = d01 * c0 + d11 * c1;        // dKyy c
    const double cd0 = -(k11 * e0 - k01 * e1) / det, cd1 = -(k00 * e1 - k01 * e0) / det;
    for (int i = tid >> 3; i < Mz; i += nt >> 3) {
        const double s = dot8(T + i * ld, 1, af, 1, i + 1, sub);
        const double pd = kd0[i] * c0 + kd1[i] * c1 + K0(i) * cd0 + K1(i) * cd1;
        if (sub == 0) v1[i] = -pd - s;
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = (tid >> 3) + 2; i < Mz; i += nt >> 3) {
        const double s = dot8(Li + i * ld, 1, v1, 1, i + 1, sub);
        if (sub == 0) acc += af[i] * s;
    }
    acc = block_sum(acc, red);
    if (tid == 0) (role == 1 ? a.ws.gkl_ell : a.ws.gkl_var)[pl] = acc;
    VG_T(l == 0 && p == 0, 203 + 10 * role);
}

template <bool TANGENTS>
__global__ __launch_bounds__(kCovThreads) void cov_b_kernel(CovArgs a) {
    extern __shared__ double sm[];
    cov_b_body<TANGENTS>(a, sm, blockIdx.x, blockIdx.y, blockIdx.z);
}

// A = Kfu (Kuu + jI)^-1 and its tangents for a tile of kRowTile time points:
//   A_ell = (dKfu/dell - A dKuu/dell) Kinv,   A_var = (jitter / var) A Kinv
// Output float32: A4[n][m] = {A, A_ell, A_var, 0} (one 16-byte load per use in the reverse pass)
// and AT[m][n] for the forward path assembly.
__device__ void cov_rows_body(const CovArgs& a, double* sm, int tile, int l, int p, int tid, int nt) {
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D, ld = (Mz + 2) & ~1;
    VG_T(tile == 0 && l == 0 && p == 0, 230);
    const float iMz = 1.0f / (float)Mz;
    const size_t pl = (size_t)p * L + l;
    double* Ki = sm;                       // [Mz][ld]
    double* Kd = Ki + Mz * ld;             // [Mz][ld]
    double* kf = Kd + Mz * ld;             // [RT][Mz]  Kfu rows
    double* df = kf + kRowTile * Mz;       // [RT][Mz]  dKfu/dell rows
    double* ar = df + kRowTile * Mz;       // [RT][Mz]  A rows
    double* yr = ar + kRowTile * Mz;       // [RT][Mz]
    double* zs = yr + kRowTile * Mz;       // [Mz]
    double* xs = zs + Mz;                  // [RT] times of this tile
    const double ell = a.ws.ell[pl], var = a.ws.var[pl];
    const int n0 = tile * kRowTile;
    {
        auto all = [](int, int) { return true; };
        if ((Mz & 1) == 0) {
            vg_stage_f64_square(Ki, ld, a.ws.Kinv + pl * Mz * Mz, Mz, tid, nt);
            if (a.want_dell) vg_stage_f64_square(Kd, ld, a.ws.Kd_ell + pl * Mz * Mz, Mz, tid, nt);
            else for (int e = tid; e < Mz * ld; e += nt) Kd[e] = 0.0;
        } else {
            vg_stage_f64(Ki, Mz, ld, a.ws.Kinv + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
            vg_stage_f64(Kd, Mz, ld, a.ws.Kd_ell + pl * Mz * Mz, a.want_dell ? Mz : 0, Mz, 0, 0, tid, nt, all);
        }
        vg_stage_words(zs, 2 * (Mz + kRowTile), tid, nt, [&](int w) -> const void* {
            const int i = w >> 1;
            const double* src = i < Mz ? a.Zy + (size_t)p * a.zy_stride + (size_t)i * D + l
                                       : a.X + (size_t)min(n0 + i - Mz, N - 1) * D + l;
            return reinterpret_cast<const uint32_t*>(src) + (w & 1);
        });
    }
    vg_dma_wait();
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 232);
    for (int e = tid; e < kRowTile * Mz; e += nt) {
        int r = vg_div(e, iMz), m = e - r * Mz, n = n0 + r;
        double k = 0.0, dk = 0.0;
        if (n < N) {
            double rr = fabs(xs[r] - zs[m]) / ell;
            double ex = exp(-kSqrt5 * rr);
            k = var * (1.0 + kSqrt5 * rr + (5.0 / 3.0) * rr * rr) * ex;
            dk = var * ex * (5.0 * rr * rr / (3.0 * ell)) * (1.0 + kSqrt5 * rr);
        }
        kf[e] = k; df[e] = dk;
    }
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 233);
    for (int e = tid; e < kRowTile * Mz; e += nt) {
        int r = vg_div(e, iMz), m = e - r * Mz;
        ar[e] = dot4(kf + r * Mz, 1, Ki + m, ld, Mz);
    }
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 234);
    float4* A4 = reinterpret_cast<float4*>(a.ws.A4) + pl * N * Mz;
    float* AT = a.ws.AT + pl * N * Mz;
    float av_keep[2] = {0.f, 0.f};
    int cnt = 0;
    for (int e = tid; e < kRowTile * Mz; e += nt, ++cnt) {
        int r = vg_div(e, iMz), m = e - r * Mz;
        const double y = df[e] - dot4(ar + r * Mz, 1, Kd + m, ld, Mz);
        const double v = dot4(ar + r * Mz, 1, Ki + m, ld, Mz);
        yr[e] = y;
        if (cnt < 2) av_keep[cnt] = (float)(a.jitter / var * v);
    }
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 235);
    cnt = 0;
    for (int e = tid; e < kRowTile * Mz; e += nt, ++cnt) {
        int r = vg_div(e, iMz), m = e - r * Mz, n = n0 + r;
        if (n >= N) continue;
        const double s = a.want_dell ? dot4(yr + r * Mz, 1, Ki + m, ld, Mz) : 0.0;
        const float av = av_keep[cnt < 2 ? cnt : 1];
        vg_stream(A4 + (size_t)n * Mz + m, make_float4((float)ar[e], (float)s, av, 0.f));
        vg_stream(AT + (size_t)m * N + n, (float)ar[e]);
    }
    VG_T(tile == 0 && l == 0 && p == 0, 231);
}

// =================================================================================================
// Random Fourier features  Phi[l, j, b] = sqrt(2 var / B) cos(x_j . omega_lb / ell + beta_lb)
// and dPhi/dell.  Points j < N are rows of X, the rest rows of Zy.
// =================================================================================================
__device__ __forceinline__ float softplus_f(float x) { return x > 15.f ? x : __logf(1.f + __expf(x)); }

struct FeatArgs {
    int N, Mz, L, D, B, jchunk;
    const double *X, *Zy, *raw_ell, *raw_var;
    size_t zy_stride;
    const float *omega, *beta;
    float *Phi, *dPhi;
    uint32_t* tick;          // device step counter, ticked by the stand-alone launch of a training step (or null)
    HyperArgs hy;            // stage 1 of a chained step: the hyper-parameter update to repeat first (features_body<true>)
};

// PRO: derive this step's hyper-parameters from the previous reverse pass first (stage 1 of a chained step).  A
// compile-time switch: the update's registers would otherwise halve the occupancy of the stand-alone kernel.
template <bool PRO>
__device__ __forceinline__ void features_body(const FeatArgs& a, int bx, int by, int bz) {
    // one lane per (latent, basis): its frequency row stays in registers while it sweeps `jchunk` points;
    // the points are uniform across the workgroup (scalar loads), the stores are coalesced along b
    const int N = a.N, Mz = a.Mz, L = a.L, D = a.D, B = a.B;
    const float *omega = a.omega, *beta = a.beta;
    float *Phi = a.Phi, *dPhi = a.dPhi;
    VG_T(bx == 0 && by == 0 && bz == 0, 130);
    const int b = bx * kBlock + threadIdx.x;
    const int l = bz % L, p = bz / L;
    const double *X = a.X, *Zy = a.Zy + (size_t)p * a.zy_stride;
    const int J = N + Mz;
    const size_t pl = (size_t)p * L + l;
    if (b >= B) return;
    double re, rv;
    if constexpr (PRO) {
        const HyperState o = hyper_update(a.hy, pl);
        re = o.raw_ell; rv = o.raw_var;
    } else {
        re = a.raw_ell[pl]; rv = a.raw_var[pl];
    }
    const float ell = softplus_f((float)re);
    const float var = (float)kVarFloor + softplus_f((float)rv);
    const float inv_ell = 1.0f / ell, c = __builtin_amdgcn_sqrtf(2.0f * var / (float)B);
    float om[VGPMP_MAX_DOF];
#pragma unroll
    for (int d = 0; d < VGPMP_MAX_DOF; ++d) om[d] = d < D ? omega[(pl * B + b) * D + d] : 0.f;
    const float bt = beta[pl * B + b];
    const int j0 = by * a.jchunk, j1 = min(J, j0 + a.jchunk);
    for (int j = j0; j < j1; ++j) {
        const double* pt = j < N ? X + (size_t)j * D : Zy + (size_t)(j - N) * D;
        float proj = 0.f;
#pragma unroll
        for (int d = 0; d < VGPMP_MAX_DOF; ++d)
            if (d < D) proj = fmaf((float)pt[d], om[d], proj);
        // v_sin/v_cos take revolutions: reduce with fract (argument is a few tens of radians at most)
        const float rev = __builtin_amdgcn_fractf((proj * inv_ell + bt) * 0.15915494309189535f);
        const size_t o = (pl * J + j) * B + b;
        // streamed past the caches: 7.6 MB per problem that the next launch reads from another XCD anyway, and
        // dirty lines left in L2 lengthen the hand-over to that launch
        vg_stream(Phi + o, c * __builtin_amdgcn_cosf(rev));
        if (dPhi) vg_stream(dPhi + o, c * __builtin_amdgcn_sinf(rev) * proj * inv_ell * inv_ell);
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 131);
    VG_T(bx == 0 && j1 == J && bz == L - 1, 135);
}

__global__ __launch_bounds__(kBlock) void features_kernel(FeatArgs a) {
    // (the step size of this update is derived from the counter by hyper_kernel in this schedule: float64 exp /
    // sqrt code here would cost this bandwidth-bound kernel half its occupancy)
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *a.tick += 1u;
    features_body<false>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// =================================================================================================
// Prior draws  F0[s, l, j] = sum_b w[s, l, b] Phi[l, j, b]   (and H with dPhi) on the f32 MFMA pipe.
// v_mfma_f32_16x16x4_f32: lane -> A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15];
// each lane loads 4 consecutive k (16 B) per operand, so one load pair feeds 4 MFMAs (k = 4g + c).
// =================================================================================================
typedef float vg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kNT = 3;     // 16-column tiles per wave

struct GemmArgs {
    int S, L, J, B, SK, nsel;
    const float *W, *Phi, *dPhi;
    float *F0, *H;
    size_t slab;
    int dbg;                 // measurement builds: 1 no stores, 2 no loads, 3 no MFMA
};

// KS > 0: the K-slice of a workgroup is a multiple of KS steps of 16 and goes in passes of KS steps whose operands
// are ALL requested before the pass's first MFMA (one L2 round trip per pass instead of one per step -- at one
// problem these launches are latency bound, not bandwidth bound).  KS == 0: any slice length, next step
// prefetched while the MFMAs of this one run.
template <int KS>
__device__ __forceinline__ void prior_gemm_body(const GemmArgs& a, int bx, int by, int bz) {
    const int S = a.S, L = a.L, J = a.J, B = a.B, SK = a.SK, nsel = a.nsel;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int z = bz;
    const int sel = z % nsel; z /= nsel;
    const int sk = z % SK; z /= SK;
    const int l = z % L, p = z / L;
    const int s0 = (by * 4 + wave) * 16;
    const int j0 = bx * (16 * kNT);
    VG_T(bx == 0 && by == 0 && bz == 0, 240);
    if (s0 >= S) return;
    const float* Bm = sel == 0 ? a.Phi : a.dPhi;
    float* Out = (sel == 0 ? a.F0 : a.H) + (size_t)sk * a.slab;
    const int kchunk = B / SK, kbeg = sk * kchunk, kend = kbeg + kchunk;
    const int r = lane & 15, g = lane >> 4;
    const int srow = min(s0 + r, S - 1);
    const float* ap = a.W + (((size_t)p * S + srow) * L + l) * B + 4 * g;
    const float* bp[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        int jc = min(j0 + 16 * t + r, J - 1);
        bp[t] = Bm + (((size_t)p * L + l) * J + jc) * B + 4 * g;
    }
    vg_f32x4 acc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc[t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (KS > 0) {
        // the K-slice in passes of KS steps: every operand of a pass is requested before its first MFMA
        for (int k0 = kbeg; k0 < kend; k0 += 16 * KS) {
            float4 av[KS], bv[kNT][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                av[ks] = *reinterpret_cast<const float4*>(ap + k0 + 16 * ks);
#pragma unroll
                for (int t = 0; t < kNT; ++t) bv[t][ks] = *reinterpret_cast<const float4*>(bp[t] + k0 + 16 * ks);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].x, bv[t][ks].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].y, bv[t][ks].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].z, bv[t][ks].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks].w, bv[t][ks].w, acc[t], 0, 0, 0);
                }
        }
    } else {
        float4 a_cur = *reinterpret_cast<const float4*>(ap + kbeg);
        float4 b_cur[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) b_cur[t] = *reinterpret_cast<const float4*>(bp[t] + kbeg);
        for (int k = kbeg; k < kend; k += 16) {
            const int kn = (k + 16 < kend) ? k + 16 : k;      // prefetch next k-step while the MFMAs run
            float4 a_nxt = *reinterpret_cast<const float4*>(ap + kn);
            float4 b_nxt[kNT];
#pragma unroll
            for (int t = 0; t < kNT; ++t) b_nxt[t] = *reinterpret_cast<const float4*>(bp[t] + kn);
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.x, b_cur[t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.y, b_cur[t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.z, b_cur[t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.w, b_cur[t].w, acc[t], 0, 0, 0);
            }
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < kNT; ++t) b_cur[t] = b_nxt[t];
        }
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 241);
#ifdef VGPMP_BISECT
    if (a.dbg == 1) { if (acc[0][0] + acc[1][1] + acc[2][2] == 123.456f) Out[0] = 1.f; return; }
#endif
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        const int jc = j0 + 16 * t + r;
        if (jc >= J) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = s0 + g * 4 + q;
            if (s < S) vg_stream(Out + (((size_t)p * S + s) * L + l) * J + jc, acc[t][q]);
        }
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 242);
    VG_T(bx == 2 && by == 1 && l == L - 1 && sk == SK - 1 && sel == nsel - 1, 245);
}

template <int KS>
__global__ __launch_bounds__(kBlock) void prior_gemm_kernel(GemmArgs a) { prior_gemm_body<KS>(a, blockIdx.x, blockIdx.y, blockIdx.z); }

// The same tile with its operands staged through LDS by DMA in passes of 128 K (the form stage 2 uses): whole
// 512-byte rows per request instead of the 16 rows x 64 bytes a fragment-shaped load touches, all requests of a
// pass in flight together, fragments by ds_read_b128.  LDS rows are padded to 132 floats (33 units of 16 bytes;
// the pad unit repeats the row's last one): the 16-byte fragment reads of a 16-row group then fall on distinct
// bank slots.  LDS: (64 + 48) x 132 x 4 = 59 KB per workgroup.
constexpr int kGK = 128, kGLd = kGK + 4, kGRows = 64 + 16 * kNT;
constexpr size_t kGemmLds = (size_t)kGRows * kGLd * sizeof(float);

__device__ __forceinline__ void prior_gemm_lds_body(const GemmArgs& a, float* lds, int bx, int by, int bz) {
    const int S = a.S, L = a.L, J = a.J, B = a.B, SK = a.SK, nsel = a.nsel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int z = bz;
    const int sel = z % nsel; z /= nsel;
    const int sk = z % SK; z /= SK;
    const int l = z % L, p = z / L;
    const int s0 = by * 64, j0 = bx * (16 * kNT);
    VG_T(bx == 0 && by == 0 && bz == 0, 240);
    const float* Bm = sel == 0 ? a.Phi : a.dPhi;
    float* Out = (sel == 0 ? a.F0 : a.H) + (size_t)sk * a.slab;
    const int kchunk = B / SK, kbeg = sk * kchunk, kend = kbeg + kchunk;
    const int r = lane & 15, g = lane >> 4;
    constexpr int kUnits = kGLd / 4;                         // 33 units per padded row
    float* As = lds;                                         // [64][kGLd]
    float* Bs = lds + 64 * kGLd;                             // [16 kNT][kGLd]
    vg_f32x4 acc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc[t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += kGK) {
        if (k0 != kbeg) __syncthreads();                     // the previous pass has been read
        for (int c = (tid & ~63); c < kGRows * kUnits; c += kBlock) {
            const int i = c + lane;
            if (i < kGRows * kUnits) {
                const int row = i / kUnits, u = min(i - row * kUnits, kGK / 4 - 1);
                const float* src = row < 64
                    ? a.W + (((size_t)p * S + min(s0 + row, S - 1)) * L + l) * B + k0 + 4 * u
                    : Bm + (((size_t)p * L + l) * J + min(j0 + row - 64, J - 1)) * B + k0 + 4 * u;
                __builtin_amdgcn_global_load_lds((vg_gmem*)src, (vg_lmem*)(lds + 4 * (size_t)c), 16, 0, VG_DMA_AUX);
            }
        }
        vg_dma_wait();
        __syncthreads();
        const float* ap = As + (wave * 16 + r) * kGLd + 4 * g;
#pragma unroll
        for (int ks = 0; ks < kGK / 16; ++ks) {
            const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * ks);
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(Bs + (16 * t + r) * kGLd + 16 * ks + 4 * g);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc[t], 0, 0, 0);
            }
        }
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 241);
    const int sw = s0 + wave * 16;
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        const int jc = j0 + 16 * t + r;
        if (jc >= J) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = sw + g * 4 + q;
            if (s < S) vg_stream(Out + (((size_t)p * S + s) * L + l) * J + jc, acc[t][q]);
        }
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 242);
    VG_T(bx == 2 && by == 1 && l == L - 1 && sk == SK - 1 && sel == nsel - 1, 245);
}

__global__ __launch_bounds__(kBlock) void prior_gemm_lds_kernel(GemmArgs a) {
    extern __shared__ float gemm_lds[];
    prior_gemm_lds_body(a, gemm_lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// LDS-tiled variant for large batches (no split-K): a workgroup owns 64 samples x 144 columns, stages
// 32-deep K slices of W and Phi through double-buffered LDS (global -> registers -> LDS, next slice in
// flight while the MFMAs run) and every wave reads its fragments with ds_read_b128.  Row stride 36 floats
// keeps the 16-byte fragment reads of a 16-row group on distinct bank slots.  Raises flop per byte
// fetched from L2 from ~10 to ~22 compared with the direct kernel above.
constexpr int kTS = 64, kTJ = 144, kTK = 32, kTLd = 36;

// MT = 16-row tiles per wave: a workgroup owns 64 MT samples.  MT = 2 (128 samples x 144 columns) moves 35 % fewer
// operand bytes per flop than MT = 1 and keeps a slice's products long enough (144 per wave) to cover the next
// slice's loads; dynamic LDS 2 (64 MT + 144) 36 4 B = 60 / 78 KB, two workgroups per CU either way.
struct TiledGemmArgs {
    int S, L, J, B, nsel;
    const float *W, *Phi, *dPhi;
    float *F0, *H;
};
template <int MT>
__device__ __forceinline__ void prior_gemm_tiled_body(const TiledGemmArgs& ta, float* tg_lds, int bx, int by, int bz) {
    const int S = ta.S, L = ta.L, J = ta.J, B = ta.B, nsel = ta.nsel;
    const float* __restrict__ W = ta.W;
    const float* __restrict__ Phi = ta.Phi;
    const float* __restrict__ dPhi = ta.dPhi;
    float* __restrict__ F0 = ta.F0;
    float* __restrict__ H = ta.H;
    constexpr int TS = kTS * MT, NA = TS * 8 / kBlock;      // A: TS rows x 8 chunks of 16 bytes, NA per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int z = bz;
    const int sel = z % nsel; z /= nsel;
    const int l = z % L, p = z / L;
    const int s0 = by * TS, j0 = bx * kTJ;
    const float* Bm = sel == 0 ? Phi : dPhi;
    float* Out = sel == 0 ? F0 : H;
    // staging map: thread -> (row, 16-byte k-chunk); B: 144 rows x 8 chunks = 1152 (4.5 per thread -> 5 passes, last partial)
    // (macros, not lambdas: staging registers captured by a lambda end up in scratch memory)
    vg_f32x4 ra[NA], rbv[5];      // (ext_vector registers: HIP's float4 struct arrays are not always promoted out of scratch)
#define VG_TG_LOAD(k0)                                                                                              \
    {                                                                                                               \
        _Pragma("unroll") for (int q = 0; q < NA; ++q) {                                                            \
            const int c = tid + q * kBlock, row = c >> 3, ch = c & 7;                                               \
            const int srow = min(s0 + row, S - 1);                                                                  \
            ra[q] = *reinterpret_cast<const vg_f32x4*>(W + (((size_t)p * S + srow) * L + l) * B + (k0) + 4 * ch);     \
        }                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 5; ++q) {                                                             \
            const int c = min(tid + q * kBlock, kTJ * 8 - 1), row = c >> 3, ch = c & 7;                             \
            const int jrow = min(j0 + row, J - 1);                                                                  \
            rbv[q] = *reinterpret_cast<const vg_f32x4*>(Bm + (((size_t)p * L + l) * J + jrow) * B + (k0) + 4 * ch);   \
        }                                                                                                           \
    }
#define VG_TG_STORE(buf)                                                                                            \
    {                                                                                                               \
        float* as_ = tg_lds + (buf) * (TS * kTLd);                                                                  \
        float* bs_ = tg_lds + 2 * TS * kTLd + (buf) * (kTJ * kTLd);                                                 \
        _Pragma("unroll") for (int q = 0; q < NA; ++q) {                                                            \
            const int c = tid + q * kBlock, row = c >> 3, ch = c & 7;                                               \
            *reinterpret_cast<vg_f32x4*>(as_ + row * kTLd + 4 * ch) = ra[q];                                          \
        }                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 5; ++q) {                                                             \
            const int c = tid + q * kBlock;                                                                         \
            if (c < kTJ * 8) {                                                                                      \
                const int row = c >> 3, ch = c & 7;                                                                 \
                *reinterpret_cast<vg_f32x4*>(bs_ + row * kTLd + 4 * ch) = rbv[q];                                     \
            }                                                                                                       \
        }                                                                                                           \
    }
    vg_f32x4 acc[MT][kTJ / 16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) acc[m][t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;
    VG_TG_LOAD(0)
    VG_TG_STORE(0)
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < B; k0 += kTK) {
        const bool more = k0 + kTK < B;
        if (more) VG_TG_LOAD(k0 + kTK)
        const float* a_base = &(tg_lds + buf * (TS * kTLd))[(wave * 16 + r) * kTLd + 4 * g];
#pragma unroll
        for (int kk = 0; kk < kTK; kk += 16) {
            float4 a4[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const float4*>(a_base + m * 64 * kTLd + kk);
#pragma unroll
            for (int t = 0; t < kTJ / 16; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(&(tg_lds + 2 * TS * kTLd + buf * (kTJ * kTLd))[(t * 16 + r) * kTLd + kk + 4 * g]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].x, b4.x, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].y, b4.y, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].z, b4.z, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].w, b4.w, acc[m][t], 0, 0, 0);
                }
            }
        }
        if (more) VG_TG_STORE(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) {
            const int jc = j0 + 16 * t + r;
            if (jc >= J) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int srow = s0 + m * 64 + wave * 16 + g * 4 + q;
                if (srow < S) vg_stream(Out + (((size_t)p * S + srow) * L + l) * J + jc, acc[m][t][q]);
            }
        }
}
#undef VG_TG_LOAD
#undef VG_TG_STORE

template <int MT>
__global__ __launch_bounds__(kBlock) void prior_gemm_tiled_kernel(TiledGemmArgs ta) {
    extern __shared__ __attribute__((aligned(16))) float tg_lds[];
    prior_gemm_tiled_body<MT>(ta, tg_lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Large batches with device-generated noise: the prior draws as ONE kernel -- the weights W come out of the Philox
// generator and the features Phi / dPhi/dell out of sin / cos INSIDE the GEMM's K loop, straight into the LDS tiles the
// MFMAs read.  Neither W (S L B floats per problem: 470 MB at 64 problems of 14 joints, written by the generator and
// read back by the GEMM) nor Phi / dPhi (2 x L J B: 15 MB per problem each way) exist in memory any more; the generator
// and the feature kernel are gone as launches.  A workgroup owns 64 samples x 144 columns of BOTH products (F0 = W Phi^T
// and H = W dPhi^T share the W tile); per 16-deep K step a thread draws one Philox counter (4 normals of a W row) and
// forms 9 feature pairs, then the four waves run 72 MFMAs each.  The VALU work of one workgroup's generation phase
// overlaps the MFMA phase of the others on the CU (33 KB of LDS: four workgroups per CU).
// Same expressions, same accumulation order as rng_normals / features_kernel / prior_gemm_tiled_kernel: bit-identical.
constexpr int kFBK = 16, kFBLd = 20;        // K step, LDS row stride (16-row fragment reads fall on distinct banks)
struct FusedBatchArgs {
    int S, L, J, N, D, B, want_dell;
    const double *X, *Zy, *raw_ell, *raw_var;
    size_t zy_stride;
    const float *omega, *beta;
    float *F0, *H;
    uint32_t seed, problem_base, step, wOff;
    const uint32_t* ctr;
};
template <bool DELL, int DM>      // d/d ell wanted; joint-space extent padded to DM (8 or 16)
__global__ __launch_bounds__(kBlock) void prior_fused_batch_kernel(FusedBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.z % L, p = blockIdx.z / L;
    const int s0 = blockIdx.y * kTS, j0 = blockIdx.x * kTJ;
    const size_t pl = (size_t)p * L + l;
    float* As = fb_lds;                                  // [64][kFBLd]         W tile
    float* Bs = As + kTS * kFBLd;                        // [2][144][kFBLd]     Phi, dPhi tiles
    float* pts = Bs + 2 * kTJ * kFBLd;                   // [144][DM]           the tile's points (rows of X, then of Zy), zero padded
    float* oms = pts + kTJ * DM;                         // [2][16][DM + 4]     the K step's frequency rows (+ phase), double buffered
    constexpr int kOLd = DM + 4;
    for (int e = tid; e < kTJ * DM; e += kBlock) {
        const int jj = e / DM, d = e - jj * DM, j = min(j0 + jj, J - 1);
        const double* pt = j < N ? a.X + (size_t)j * D : a.Zy + (size_t)p * a.zy_stride + (size_t)(j - N) * D;
        pts[e] = d < D ? (float)pt[d] : 0.f;
    }
    for (int e = tid; e < 2 * kFBK * kOLd; e += kBlock) oms[e] = 0.f;
    const float ell = softplus_f((float)a.raw_ell[pl]);
    const float var = (float)kVarFloor + softplus_f((float)a.raw_var[pl]);
    const float inv_ell = 1.0f / ell, c = __builtin_amdgcn_sqrtf(2.0f * var / (float)B);
    const uint2 key = vg_key(a.seed, a.problem_base + p, a.ctr ? *a.ctr : a.step);
    // generation roles: W -- thread (row = tid / 4, quad = tid % 4) draws the 4 normals of columns 4 quad .. 4 quad + 3;
    // features -- thread (kcol = tid % 16, jg = tid / 16) forms rows jg, jg + 16, ... of column kcol;
    // frequencies -- thread t < 16 D fetches element t of the step's 16 contiguous rows of omega, t < 16 + 16 D a phase
    const int wrow = tid >> 2, wq = tid & 3;
    const int srow = min(s0 + wrow, S - 1);
    const uint32_t wbase = (a.wOff + ((uint32_t)srow * L + l) * (uint32_t)B) >> 2;      // counter of (row, column 0)
    const int kcol = tid & 15, jg = tid >> 4;
    const int nom = kFBK * D;
    const bool is_om = tid < nom, is_bt = tid >= nom && tid < nom + kFBK;
    const int orow = is_om ? tid / D : tid - nom, ocol = is_om ? tid - orow * D : DM;      // phase sits behind the row
    const float* osrc = is_om ? a.omega + pl * B * D + tid : a.beta + pl * B + (tid - nom);
    const int ostep = is_om ? kFBK * D : kFBK;
    float onext = (is_om || is_bt) ? osrc[0] : 0.f;
    vg_f32x4 accF[kTJ / 16], accH[kTJ / 16];
#pragma unroll
    for (int t = 0; t < kTJ / 16; ++t) { accF[t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f}; accH[t] = accF[t]; }
    const int r = lane & 15, g = lane >> 4;
    __syncthreads();
    if (is_om || is_bt) oms[orow * kOLd + ocol] = onext;
    __syncthreads();
    int ob = 0;
    for (int k0 = 0; k0 < B; k0 += kFBK) {
        // ---- generate the K step's operands (the next step's frequencies are requested first, stored last)
        {
            if ((is_om || is_bt) && k0 + kFBK < B) onext = osrc[(size_t)(k0 / kFBK + 1) * ostep];
            const float4 w4 = vg_normal4(wbase + (uint32_t)((k0 >> 2) + wq), VG_STREAM_W, key);
            *reinterpret_cast<float4*>(As + wrow * kFBLd + 4 * wq) = w4;
            float om[DM];
            const float* orowp = oms + (ob * kFBK + kcol) * kOLd;
#pragma unroll
            for (int d = 0; d < DM; d += 4) {
                const float4 o4 = *reinterpret_cast<const float4*>(orowp + d);
                om[d] = o4.x; om[d + 1] = o4.y; om[d + 2] = o4.z; om[d + 3] = o4.w;
            }
            const float bt = orowp[DM];
#pragma unroll
            for (int i = 0; i < kTJ / 16; ++i) {
                const int jj = jg + 16 * i;
                float proj = 0.f;
#pragma unroll
                for (int d = 0; d < DM; d += 4) {          // (zero padding: the products beyond D add exact zeros)
                    const float4 p4 = *reinterpret_cast<const float4*>(pts + jj * DM + d);
                    proj = fmaf(p4.x, om[d], proj); proj = fmaf(p4.y, om[d + 1], proj);
                    proj = fmaf(p4.z, om[d + 2], proj); proj = fmaf(p4.w, om[d + 3], proj);
                }
                const float rev = __builtin_amdgcn_fractf((proj * inv_ell + bt) * 0.15915494309189535f);
                Bs[jj * kFBLd + kcol] = c * __builtin_amdgcn_cosf(rev);
                if (DELL) Bs[(kTJ + jj) * kFBLd + kcol] = c * __builtin_amdgcn_sinf(rev) * proj * inv_ell * inv_ell;
            }
            if ((is_om || is_bt) && k0 + kFBK < B) oms[((ob ^ 1) * kFBK + orow) * kOLd + ocol] = onext;
        }
        __syncthreads();
        // ---- 2 x 9 tiles of 16 x 16, four k-interleaved MFMAs each
        {
            const float4 a4 = *reinterpret_cast<const float4*>(As + (wave * 16 + r) * kFBLd + 4 * g);
#pragma unroll
            for (int t = 0; t < kTJ / 16; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(Bs + (t * 16 + r) * kFBLd + 4 * g);
                accF[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, accF[t], 0, 0, 0);
                accF[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, accF[t], 0, 0, 0);
                accF[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, accF[t], 0, 0, 0);
                accF[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, accF[t], 0, 0, 0);
                if (DELL) {
                    const float4 d4 = *reinterpret_cast<const float4*>(Bs + (kTJ + t * 16 + r) * kFBLd + 4 * g);
                    accH[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, d4.x, accH[t], 0, 0, 0);
                    accH[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, d4.y, accH[t], 0, 0, 0);
                    accH[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, d4.z, accH[t], 0, 0, 0);
                    accH[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, d4.w, accH[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        ob ^= 1;
    }
#pragma unroll
    for (int t = 0; t < kTJ / 16; ++t) {
        const int jc = j0 + 16 * t + r;
        if (jc >= J) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = s0 + wave * 16 + g * 4 + q;
            if (s < S) {
                vg_stream(a.F0 + (((size_t)p * S + s) * L + l) * J + jc, accF[t][q]);
                if (DELL) vg_stream(a.H + (((size_t)p * S + s) * L + l) * J + jc, accH[t][q]);
            }
        }
    }
}
__global__ void tick_kernel(uint32_t* ctr) { *ctr += 1u; }

// Few samples (S <= 32): the prior draws are bound by Phi / dPhi themselves -- every feature is used by only S
// products, so writing the two matrices (features_kernel) and reading them back (GEMM) is the cost: 157 MB each way
// for 36 problems of the reference's default shape.  Here a wave forms its feature fragments in registers and feeds
// them straight to the MFMAs; Phi / dPhi are never stored.  One workgroup per (problem, latent, group of 5 column
// tiles), wave w = K-slice w of the SK = 4 slabs the path kernels sum anyway.
struct FusedPriorArgs {
    int S, L, J, N, D, B, want_dell;
    const double *X, *Zy, *raw_ell, *raw_var;
    size_t zy_stride;
    const float *omega, *beta, *W;
    float *F0, *H;
    size_t slab;
    uint32_t* tick;
};
constexpr int kFNT = 5;      // column tiles per workgroup
template <int MT, int DM, bool DELL>    // 16-row sample tiles; joint-space extent padded to DM (8 or 16); d/d ell wanted
__global__ __launch_bounds__(kBlock) void prior_fused_small_kernel(FusedPriorArgs a) {
    __shared__ float pts[kFNT * 16][DM];
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, sk = tid >> 6;      // 4 waves = 4 K-slices
    const int pl = blockIdx.x, l = pl % L, p = pl / L, j0 = blockIdx.y * (kFNT * 16);
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.tick += 1u;
    for (int e = tid; e < kFNT * 16 * DM; e += kBlock) {
        const int jj = e / DM, d = e - jj * DM, j = min(j0 + jj, J - 1);
        const double* pt = j < N ? a.X + (size_t)j * D : a.Zy + (size_t)p * a.zy_stride + (size_t)(j - N) * D;
        pts[jj][d] = d < D ? (float)pt[d] : 0.f;
    }
    __syncthreads();
    const float ell = softplus_f((float)a.raw_ell[pl]);
    const float var = (float)kVarFloor + softplus_f((float)a.raw_var[pl]);
    const float inv_ell = 1.0f / ell, c = __builtin_amdgcn_sqrtf(2.0f * var / (float)B);
    const int r = lane & 15, g = lane >> 4;
    const int kchunk = B / 4, kbeg = sk * kchunk;
    vg_f32x4 accF[MT][kFNT], accH[MT][kFNT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kFNT; ++t) { accF[m][t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f}; accH[m][t] = accF[m][t]; }
    const float* wrow[MT];
    bool wlive[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int s = 16 * m + r;
        wrow[m] = a.W + (((size_t)p * S + min(s, S - 1)) * L + l) * B;
        wlive[m] = s < S;
    }
    // operands of a pass (4 bases per lane: 28 frequencies, 4 phases, the W fragments) are requested one pass ahead
    float om[4][DM], bt[4], om_n[4][DM], bt_n[4];
    vg_f32x4 a4[MT], a4_n[MT];
    auto fetch = [&](int k0, float (&o)[4][DM], float (&bb)[4], vg_f32x4 (&aa)[MT]) {
        const int b0 = min(k0, B - 16) + 4 * g;          // (the look-ahead of the last pass re-reads it)
        // every load unconditional on a clamped index, masked afterwards (a conditional load is a branch)
        const float* op = a.omega + ((size_t)pl * B + b0) * D;
        const float* bp = a.beta + (size_t)pl * B + b0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bb[q] = bp[q];
#pragma unroll
            for (int d = 0; d < DM; ++d) {
                const float v = op[q * D + min(d, D - 1)];
                o[q][d] = d < D ? v : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const vg_f32x4 v = *reinterpret_cast<const vg_f32x4*>(wrow[m] + b0);
            aa[m] = wlive[m] ? v : (vg_f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(kbeg, om_n, bt_n, a4_n);
    for (int k0 = kbeg; k0 < kbeg + kchunk; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bt[q] = bt_n[q];
#pragma unroll
            for (int d = 0; d < DM; ++d) om[q][d] = om_n[q][d];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) a4[m] = a4_n[m];
        fetch(k0 + 16, om_n, bt_n, a4_n);
        // one wave per SIMD issues in order: the features of column tile t + 1 are formed between the products of
        // tile t (independent work next to each other in the instruction stream), not after them
        float ph[2][4], dh[2][4];
        auto feats = [&](int t, float (&pc)[4], float (&dc)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float proj = 0.f;
#pragma unroll
                for (int d = 0; d < DM; ++d) proj = fmaf(pts[16 * t + r][d], om[q][d], proj);
                const float rev = __builtin_amdgcn_fractf((proj * inv_ell + bt[q]) * 0.15915494309189535f);
                pc[q] = c * __builtin_amdgcn_cosf(rev);
                dc[q] = c * __builtin_amdgcn_sinf(rev) * proj * inv_ell * inv_ell;
            }
        };
        feats(0, ph[0], dh[0]);
#pragma unroll
        for (int t = 0; t < kFNT; ++t) {
            const int cb = t & 1;
            if (t + 1 < kFNT) feats(t + 1, ph[cb ^ 1], dh[cb ^ 1]);
            // (k outermost: consecutive products go to different accumulators; the order per accumulator is unchanged)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][q], ph[cb][q], accF[m][t], 0, 0, 0);
                    if (DELL) accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][q], dh[cb][q], accH[m][t], 0, 0, 0);
                }
        }
    }
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* F0 = a.F0 + (size_t)sk * a.slab;
    float* H = a.H + (size_t)sk * a.slab;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kFNT; ++t) {
            const int jc = j0 + 16 * t + r;
            if (jc >= J) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 16 * m + g * 4 + q;
                if (s >= S) continue;
                const size_t o = (((size_t)p * S + s) * L + l) * J + jc;
                vg_stream(F0 + o, accF[m][t][q]);
                if (DELL) vg_stream(H + o, accH[m][t][q]);
            }
        }
}

// =================================================================================================
// Path assembly  (decoupled / Matheron update, vgpmp.py:281-282):
//   u = m + C eps;  r = u - F0(Z) - sqrt(jitter) eps2;  f = F0(X) + A r
// One workgroup per (chunk of VG_SC samples, latent, problem).
// =================================================================================================
// the same update with the state already in registers
__device__ __forceinline__ void adam_apply(double* x, double* m, double* v, double x0, double m0, double v0, double g,
                                           double lr_t) {
    const double mm = m0 + (g - m0) * (1.0 - 0.8);
    const double vv = v0 + (g * g - v0) * (1.0 - 0.95);
    *m = mm; *v = vv;
    *x = x0 - lr_t * mm / (sqrt(vv) + 1e-7);
}

// sum over the NC sample chunks of one reverse-pass partial, 8 independent loads in flight per pass
// (unconditional clamped loads, masked afterwards); fixed order: deterministic
__device__ __forceinline__ double sum_chunks(const float* part, size_t part_len, int NC, int e) {
    double s = 0.0;
    for (int c0 = 0; c0 < NC; c0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k < NC ? c0 + k : NC - 1;
            v[k] = part[(size_t)c * part_len + e];
        }
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = c0 + k < NC ? (double)v[k] : 0.0;
        s += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    }
    return s;
}

// one wave per problem, one lane per latent
__global__ __launch_bounds__(64) void hyper_kernel(HyperArgs h) {
    const int p = blockIdx.x, l = threadIdx.x;
    VG_T(p == 0, 600);
    const bool own = h.ctr && h.do_adam;
    double lr_own = 0.0;
    if (own) {
        lr_own = adam_step_size(h.lr, (double)*h.ctr);
        if (p == 0 && l == 0) h.lr_store[0] = lr_own;         // final_kernel reads it
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
    VG_T(p == 0, 601);
}

// alpha / S * sum of the per-workgroup log-likelihood sums and the KL total of one problem: whole workgroup,
// fixed order (thread-strided partial sums, wave sums, then the waves in order) -- shared by the forward-only
// epilogue so that both entry points return identical numbers
__device__ __forceinline__ void elbo_pieces(const float* lik_partial, int nblk, const double* kl_l, int L, int p,
                                            double lik_scale, double kls, double* out_lik, double* out_kl) {
    __shared__ double red2[2][kBlock / VG_WAVE];
    const int tid = threadIdx.x, nt = blockDim.x;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = lik_partial[(size_t)p * nblk + min(tid + k * nt, nblk - 1)];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += tid + k * nt < nblk ? (double)v[k] : 0.0;
    for (int k = tid + 4 * nt; k < nblk; k += nt) s += (double)lik_partial[(size_t)p * nblk + k];
    double kk = tid < L ? kl_l[(size_t)p * L + tid] : 0.0;
    s = vg_wave_sum(s);
    kk = vg_wave_sum(kk);
    if ((tid & (VG_WAVE - 1)) == 0) { red2[0][tid / VG_WAVE] = s; red2[1][tid / VG_WAVE] = kk; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, c = 0.0;
        for (int k = 0; k < (int)(nt / VG_WAVE); ++k) { a += red2[0][k]; c += red2[1][k]; }
        out_lik[p] = lik_scale * a;
        out_kl[p] = kls * c;
    }
}

// Gradient assembly of one (latent, problem).  Everything it reads is requested up front -- the chunk partials
// and the Cholesky factor by DMA into LDS (when `dma`), the KL gradients and the Adam state of this thread's
// elements into registers -- so the kernel waits for memory once, not once per loop iteration.
constexpr int kFinRegs = ((VGPMP_MAX_MZ - 2) * (VGPMP_MAX_MZ - 1) + kBlock - 1) / kBlock;      // elements per thread
__device__ __forceinline__ void final_body(const FinalArgs& b, double* sm, int l, int p) {
    VG_STOP(b, 7);
    const int tid = threadIdx.x, nt = blockDim.x;
    VG_T(l == 0 && p == 0, 110);
    const int M = b.M, Mz = M + 2, L = b.L, nq = M + M * M, np = Mz + Mz * Mz;
    const float iM = 1.0f / (float)M;
    const size_t pl = (size_t)p * L + l;
    double* dC = sm;                 // [Mz][Mz]
    double* dmv = dC + Mz * Mz;      // [Mz]
    float* Lks = reinterpret_cast<float*>(dmv + Mz + (Mz & 1));   // [Mz][Mz] chol factor (16-byte aligned)
    float* raw = Lks + ((Mz * Mz + 3) & ~3);                      // [NC][np] chunk partials as they arrive (dma)
    const float* part = b.part + pl * b.NC * b.part_len;
    const double lr_t = b.do_adam ? (b.use_lr_dev ? b.lr_dev[0] : b.lr_t) : 0.0;
    vg_stage_rows(Lks, 1, Mz * Mz, tid, nt, [&](int) -> const float* { return b.Lk32 + pl * Mz * Mz; });
    if (b.dma) vg_stage_rows(raw, b.NC, np, tid, nt, [&](int c) -> const float* { return part + (size_t)c * b.part_len; });
    // this thread's elements k = tid + j * nt of  q_mu | q_sqrt:  KL gradient and Adam state
    double kg[kFinRegs], xs[kFinRegs], mo[kFinRegs], vo[kFinRegs];
#pragma unroll
    for (int j = 0; j < kFinRegs; ++j) {
        const int k = min(tid + j * nt, nq - 1);
        const bool mu = k < M;
        const size_t o = mu ? pl * M + k : pl * M * M + (k - M);
        kg[j] = (mu ? b.gkl_qmu : b.gkl_Q)[o];
        if (b.do_adam) {
            xs[j] = (mu ? b.pq_mu : b.pq_sqrt)[o];
            mo[j] = (mu ? b.mq_mu : b.mq_sqrt)[o];
            vo[j] = (mu ? b.vq_mu : b.vq_sqrt)[o];
        }
    }
    VG_STOP(b, 5);
    if (!b.dma)
        for (int e = tid; e < np; e += nt) {
            const double s = sum_chunks(part, b.part_len, b.NC, e);
            if (e < Mz) dmv[e] = s;
            else dC[e - Mz] = s;
        }
    VG_T(l == 0 && p == 0, 115);
    vg_dma_wait();
    __syncthreads();
    VG_T(l == 0 && p == 0, 114);
    if (b.dma) {
        for (int e = tid; e < np; e += nt) {
            double s = 0.0;
            for (int c0 = 0; c0 < b.NC; c0 += 8) {       // the order of sum_chunks()
                double d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = c0 + k < b.NC ? (double)raw[(size_t)min(c0 + k, b.NC - 1) * np + e] : 0.0;
                s += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
            }
            if (e < Mz) dmv[e] = s;
            else dC[e - Mz] = s;
        }
        __syncthreads();
    }
    VG_STOP(b, 6);
    VG_T(l == 0 && p == 0, 111);
    VG_STOP(b, 1);
    const double kls = b.kl_scale;
    double* gQ = b.g_qsqrt + pl * M * M;
    double* gm = b.g_qmu + pl * M;
#pragma unroll
    for (int j = 0; j < kFinRegs; ++j) {
        const int k = tid + j * nt;
        if (k >= nq) continue;
        double g;
        if (k < M) {
            g = dmv[k + 2] + kls * kg[j];
            gm[k] = g;
            if (b.do_adam && (b.trainable & VGPMP_TRAIN_Q_MU))
                adam_apply(b.pq_mu + pl * M + k, b.mq_mu + pl * M + k, b.vq_mu + pl * M + k, xs[j], mo[j], vo[j], g, lr_t);
        } else {
            const int e = k - M, r = vg_div(e, iM), c = e - r * M;
            g = 0.0;
            if (c <= r) {
                // tril(Lk^T dC)[2:, 2:]
                double s0 = 0.0, s1 = 0.0;
                int i = r + 2;
                for (; i + 1 < Mz; i += 2) {
                    s0 = fma((double)Lks[i * Mz + (r + 2)], dC[i * Mz + (c + 2)], s0);
                    s1 = fma((double)Lks[(i + 1) * Mz + (r + 2)], dC[(i + 1) * Mz + (c + 2)], s1);
                }
                if (i < Mz) s0 = fma((double)Lks[i * Mz + (r + 2)], dC[i * Mz + (c + 2)], s0);
                g = s0 + s1 + kls * kg[j];
            }
            gQ[e] = g;
            if (c <= r && b.do_adam && (b.trainable & VGPMP_TRAIN_Q_SQRT))
                adam_apply(b.pq_sqrt + pl * M * M + e, b.mq_sqrt + pl * M * M + e, b.vq_sqrt + pl * M * M + e, xs[j], mo[j],
                           vo[j], g, lr_t);
        }
    }
    VG_STOP(b, 2);
    VG_T(l == 0 && p == 0, 112);
    if (l == 0) elbo_pieces(b.lik_partial, b.nblk, b.kl_l, L, p, b.alpha_fin ? b.alpha_fin[p] : b.lik_scale, kls, b.out_lik, b.out_kl);
    VG_T(l == 0 && p == 0, 113);
}

__global__ __launch_bounds__(kBlock) void final_kernel(FinalArgs b) {
    extern __shared__ double sm[];
    final_body(b, sm, blockIdx.x, blockIdx.y);
}

// forward-only epilogue: ELBO pieces without the reverse pass
__global__ __launch_bounds__(kBlock) void elbo_pieces_kernel(int L, int nblk, const float* __restrict__ lik_partial,
                                                              const double* __restrict__ kl_l, double lik_scale, double kls,
                                                              double* __restrict__ out_lik, double* __restrict__ out_kl,
                                                              const double* __restrict__ alpha_fin) {
    elbo_pieces(lik_partial, nblk, kl_l, L, blockIdx.x, alpha_fin ? alpha_fin[blockIdx.x] : lik_scale, kls, out_lik, out_kl);
}

// stand-alone Adam over the packed variables (sample-sharded mode, after the gradient all-reduce)
__global__ __launch_bounds__(kBlock) void adam_kernel(size_t n, double* __restrict__ x, const double* __restrict__ g,
                                                       double* __restrict__ m, double* __restrict__ v, double lr_t,
                                                       int tril_M) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (tril_M > 0) {
        int e = (int)(i % ((size_t)tril_M * tril_M));
        if (e % tril_M > e / tril_M) return;
    }
    adam_update(x + i, m + i, v + i, g[i], lr_t);
}

template <int SK, bool RAW>
__global__ __launch_bounds__(kBlock) void paths_fwd_sc8(PathArgs a) {
    extern __shared__ float smf[];
    if (SK > 1 && a.nsplit == 2) {
        if constexpr (SK > 1) {
            if (a.Mz == 32) paths_fwd_split_body<SK, 32>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
            else paths_fwd_split_body<SK>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
        }
        return;
    }
    paths_fwd_body<SK, 8, RAW>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
}

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
    int skip;                 // measurement builds: bit mask of roles that return at once
};
template <bool PRO>
__global__ __launch_bounds__(kBlock) void stage1_kernel(Stage1Args a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) { if (!(a.skip & 1)) cov_a_body(a.cov, sm, b % a.cov.L, b / a.cov.L); return; }
    b -= a.n_cov;
    if (b < a.n_fin) { if (!(a.skip & 2)) final_body(a.fin, sm, b % a.fin.L, b / a.fin.L); return; }
    b -= a.n_fin;
    if (b < a.n_eps) { if (!(a.skip & 4)) rng_normals_body(a.rng, b % a.eps_gx, b / a.eps_gx, 0u, a.rng.nE); return; }
    b -= a.n_eps;
    if (a.skip & 8) return;
    const int bx = b % a.feat_gx;
    b /= a.feat_gx;
    features_body<PRO>(a.feat, bx, b % a.feat_gy, b / a.feat_gy);
}

struct Stage2Args {
    CovArgs cov; GemmArgs gemm;
    int n_cov, cov_roles, gemm_gx, gemm_gy, gemm_per_xcd;
    int skip;
};
template <bool TANGENTS, int KS>
__global__ __launch_bounds__(kBlock) void stage2_kernel(Stage2Args a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) {
        if (a.skip & 1) return;
        const int role = b % a.cov_roles;
        b /= a.cov_roles;
        cov_b_body<TANGENTS>(a.cov, sm, role, b % a.cov.L, b / a.cov.L);
        return;
    }
    b -= a.n_cov;
    if (a.skip & 2) return;
    // Workgroups go to the 8 XCDs round robin (n_cov is a multiple of 8 or the remap is off): give each XCD a
    // CONTIGUOUS range of GEMM tiles, so the column / row tiles that share operands share an L2 as well.
    if (a.gemm_per_xcd > 0) b = (b & 7) * a.gemm_per_xcd + (b >> 3);
    const int bx = b % a.gemm_gx;
    b /= a.gemm_gx;
    if constexpr (KS < 0) prior_gemm_lds_body(a.gemm, reinterpret_cast<float*>(sm), bx, b % a.gemm_gy, b / a.gemm_gy);
    else prior_gemm_body<KS>(a.gemm, bx, b % a.gemm_gy, b / a.gemm_gy);
}

struct Stage3Args {
    PathArgs path; RngArgs rng;
    int n_path, n_basis, basis_gx, w_gx;
    int skip;
};
template <int SK, bool RAW>
__global__ __launch_bounds__(kBlock) void stage3_kernel(Stage3Args a) {
    extern __shared__ float smf[];
    int b = blockIdx.x;
    if (b < a.n_path) {
        if (a.skip & 1) return;
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
    if (b < a.n_basis) { rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx); return; }
    b -= a.n_basis;
    rng_normals_body(a.rng, b % a.w_gx, b / a.w_gx, a.rng.nW, 0u);
}

// ---- medium batches (5 problems up, one launch per kernel): independent kernels of a dependency level share a
// launch here too.  The latency-bound ones (cov_a, cov_b, hyper-parameter update, final) then run beside the
// throughput-bound ones (noise draws, tiled GEMM) instead of in front of them: 7 launches per step instead of 11.
struct MidAArgs {            // cov_a | omega, beta | w, eps, eps'
    CovArgs cov; RngArgs rng;
    int n_cov, n_basis, basis_gx, n_gx;
};
__global__ __launch_bounds__(kCovThreads) void mid_cov_a_rng_kernel(MidAArgs a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) { cov_a_body(a.cov, sm, b % a.cov.L, b / a.cov.L); return; }
    b -= a.n_cov;
    if (b < a.n_basis) { rng_basis_body(a.rng, b % a.basis_gx, b / a.basis_gx); return; }
    b -= a.n_basis;
    rng_normals_body(a.rng, b % a.n_gx, b / a.n_gx, a.rng.nW, a.rng.nE);
}

struct MidCArgs {            // cov_b | tiled prior GEMM
    CovArgs cov; TiledGemmArgs gemm;
    int cov_roles, n_cov, gemm_gx, gemm_gy;
};
template <bool TANGENTS>
__global__ __launch_bounds__(kBlock) void mid_cov_b_gemm_kernel(MidCArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int b = blockIdx.x;
    if (b < a.n_cov) {
        const int role = b % a.cov_roles;
        b /= a.cov_roles;
        cov_b_body<TANGENTS>(a.cov, sm, role, b % a.cov.L, b / a.cov.L);
        return;
    }
    b -= a.n_cov;
    const int bx = b % a.gemm_gx;
    b /= a.gemm_gx;
    prior_gemm_tiled_body<1>(a.gemm, reinterpret_cast<float*>(sm), bx, b % a.gemm_gy, b / a.gemm_gy);
}

struct MidGArgs {            // hyper-parameter update | q_mu, q_sqrt update + ELBO pieces
    HyperArgs hy; FinalArgs fin;
    int n_hyper;             // = problems
};
__global__ __launch_bounds__(kBlock) void mid_hyper_final_kernel(MidGArgs a) {
    extern __shared__ double sm[];
    int b = blockIdx.x;
    const HyperArgs& h = a.hy;
    const bool own = h.ctr && h.do_adam;
    const double lr_own = own ? adam_step_size(h.lr, (double)*h.ctr) : 0.0;
    if (b < a.n_hyper) {
        const int p = b, l = threadIdx.x;
        if (own && p == 0 && l == 0) h.lr_store[0] = lr_own;
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
    final_body(fb, sm, b % fb.L, b / fb.L);
}

// ---- likelihood constants as trainable variables (vgpmp_lik_params) -----------------------------------------
constexpr double kAlphaFloor = 1e-4, kSigmaFloor = 1e-5;      // models/vgpmp.py:82, likelihoods/likelihood.py:31,41

struct LikConstArgs {
    const double *raw_alpha, *raw_sigma;
    vg_lik_scratch sc;
    double inv_s;          // 1 / S_total
};
// effective constants from the raw variables (start of every call): one wave per problem, one lane per sphere
__global__ __launch_bounds__(VGPMP_MAX_SPHERES) void lik_consts_kernel(LikConstArgs a) {
    const int p = blockIdx.x, q = threadIdx.x;
    if (q == 0) {
        const double al = (kAlphaFloor + softplus_d(a.raw_alpha[p])) * a.inv_s;
        a.sc.alpha_fin[p] = al;
        a.sc.alpha_eff[p] = (float)al;
    }
    a.sc.sigma_eff[(size_t)p * VGPMP_MAX_SPHERES + q] = (float)(kSigmaFloor + softplus_d(a.raw_sigma[(size_t)p * VGPMP_MAX_SPHERES + q]));
}

struct LikUpdArgs {
    const vgpmp_robot* rb;
    const float *lik_partial, *sig_partial;
    int nblk;
    double inv_s;          // 1 / S_total
    double *raw_alpha, *raw_sigma, *m_alpha, *v_alpha, *m_sigma, *v_sigma, *g_alpha, *g_sigma;
    vg_lik_scratch sc;
    int do_adam, trainable;
    const uint32_t* ctr;   // ticked device counter (then the step size comes from it), else lr_t
    double lr, lr_t;
};
// gradient of the training loss wrt (raw_alpha, raw_sigma) of one problem, Adam, and the constants of the next step.
//   loss = -(ELBO + log sigmoid(raw_alpha) + sum_q log sigmoid(raw_sigma_q))      (vgpmp.h: vgpmp_lik_params)
//   d ELBO / d alpha = (1/S) sum_{s,n} logp,   d ELBO / d sigma_q = (alpha/S) 1/2 sum_{s,n} c_q^2 / sigma_q^2
// One wave per problem, lane q = sphere q; sums over the likelihood's workgroups in fixed order.
__global__ __launch_bounds__(VGPMP_MAX_SPHERES) void lik_update_kernel(LikUpdArgs a) {
    const int p = blockIdx.x, q = threadIdx.x, nsph = a.rb->num_spheres;
    double ls = 0.0;
    for (int b = q; b < a.nblk; b += VGPMP_MAX_SPHERES) ls += (double)a.lik_partial[(size_t)p * a.nblk + b];
    ls = vg_wave_sum(ls);                                        // sum_{s,n} logp
    double c2 = 0.0;
    const float* sp = a.sig_partial + (size_t)p * a.nblk * VGPMP_MAX_SPHERES + q;
    int b = 0;
    for (; b + 7 < a.nblk; b += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sp[(size_t)(b + k) * VGPMP_MAX_SPHERES];
#pragma unroll
        for (int k = 0; k < 8; ++k) c2 += (double)v[k];
    }
    for (; b < a.nblk; ++b) c2 += (double)sp[(size_t)b * VGPMP_MAX_SPHERES];      // sum_{s,n} c_q^2 / sigma_q
    const size_t pq = (size_t)p * VGPMP_MAX_SPHERES + q;
    const double lr_t = a.ctr ? adam_step_size(a.lr, (double)*a.ctr) : a.lr_t;
    double ra = a.raw_alpha[p], rs = a.raw_sigma[pq];
    const double alpha = kAlphaFloor + softplus_d(ra), sigma = kSigmaFloor + softplus_d(rs);
    const double gs = q < nsph ? -(alpha * a.inv_s * 0.5 * c2 / sigma * sigmoid_d(rs) + sigmoid_d(-rs)) : 0.0;
    a.g_sigma[pq] = gs;
    if (a.do_adam && (a.trainable & VGPMP_TRAIN_SIGMA_OBS) && q < nsph) adam_update(&rs, a.m_sigma + pq, a.v_sigma + pq, gs, lr_t);
    if (a.do_adam && (a.trainable & VGPMP_TRAIN_SIGMA_OBS)) a.raw_sigma[pq] = rs;
    a.sc.sigma_eff[pq] = (float)(kSigmaFloor + softplus_d(rs));
    if (q == 0) {
        const double ga = -(ls * a.inv_s * sigmoid_d(ra) + sigmoid_d(-ra));
        a.g_alpha[p] = ga;
        a.sc.alpha_fin[p] = alpha * a.inv_s;
        if (a.do_adam && (a.trainable & VGPMP_TRAIN_ALPHA)) {
            adam_update(&ra, a.m_alpha + p, a.v_alpha + p, ga, lr_t);
            a.raw_alpha[p] = ra;
        }
        a.sc.alpha_eff[p] = (float)((kAlphaFloor + softplus_d(ra)) * a.inv_s);
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
    r.nW = (uint32_t)d->S * d->L * d->B;                 // nW % 4 == 0 since B % 16 == 0
    r.nE = (uint32_t)d->S * vg_mz(d) * d->L;
    r.wOff = (uint32_t)d->sample_offset * d->L * d->B;
    r.eOff = (uint32_t)d->sample_offset * vg_mz(d) * d->L;
    r.omega = nz->omega; r.beta = nz->beta; r.w = nz->w; r.eps = nz->eps; r.eps2 = nz->eps2;
    r.seed = seed; r.problem_base = problem_base; r.step = step; r.bias = bias; r.ctr = ctr;
    return r;
}

int vg_launch_rng(const vgpmp_dims* d, const vgpmp_noise* nz, uint32_t seed, uint32_t problem_base, uint32_t step,
                  const uint32_t* ctr, hipStream_t st) {
    const int P = d->num_problems;
    RngArgs r = make_rng_args(d, nz, seed, problem_base, step, ctr, 0u);
    hipLaunchKernelGGL(rng_basis_kernel, dim3((r.L * r.B + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
    const uint32_t nthr = (r.nW >> 2) + 2 * r.nE;
    hipLaunchKernelGGL(rng_normals_kernel, dim3((nthr + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
    return (int)hipGetLastError();
}

static double adam_lr_t(double lr, int t) { return lr * sqrt(1.0 - pow(0.95, t)) / (1.0 - pow(0.8, t)); }

int vg_launch_adam(const vgpmp_dims* d, const vgpmp_params* x, const vgpmp_params* g, const vgpmp_params* am,
                   const vgpmp_params* av, int trainable, double lr, int t, hipStream_t st) {
    const size_t P = d->num_problems, L = d->L, M = d->M;
    const double lr_t = adam_lr_t(lr, t);
    auto go = [&](size_t n, double* xx, const double* gg, double* mm, double* vv, int tril) {
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, xx, gg, mm,
                           vv, lr_t, tril);
    };
    if (trainable & VGPMP_TRAIN_Q_MU) go(P * L * M, x->q_mu, g->q_mu, am->q_mu, av->q_mu, 0);
    if (trainable & VGPMP_TRAIN_Q_SQRT) go(P * L * M * M, x->q_sqrt, g->q_sqrt, am->q_sqrt, av->q_sqrt, (int)M);
    if (trainable & VGPMP_TRAIN_LENGTHSCALES) go(P * L, x->raw_ell, g->raw_ell, am->raw_ell, av->raw_ell, 0);
    if (trainable & VGPMP_TRAIN_KERNEL_VARIANCE) go(P * L, x->raw_var, g->raw_var, am->raw_var, av->raw_var, 0);
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
    const bool fused = !ev && !(what & VGPMP_NO_FUSE) && SC == 8 && P * L <= kFuseMaxPL && !pb->ind;
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
    ca.tick = (fused && do_adam) ? ctr : nullptr;
    ca.lr = lr; ca.lr_dev = ws->lr_t;
    ca.prologue = 0; ca.commit = 0; ca.keep_prev = (fused && backward) ? 1 : 0;
    ca.ws = *ws;
    FeatArgs fe;
    fe.N = N; fe.Mz = Mz; fe.L = L; fe.D = L; fe.B = B;
    // few problems: few points per workgroup (more parallelism); many: sweep 16 points per lane (omega reuse)
    // (shared launches: 8 for one problem -- the role is off the pole either way --, 16 from two: 105 -> 95 us per step)
    fe.jchunk = fused ? (P > 1 ? 16 : 8) : (P * L >= 16 ? 16 : 4);
    fe.X = pb->X; fe.Zy = zy; fe.zy_stride = zy_stride; fe.raw_ell = params->raw_ell; fe.raw_var = params->raw_var;
    fe.omega = nz->omega; fe.beta = nz->beta; fe.Phi = ws->Phi; fe.dPhi = want_dell ? ws->dPhi : nullptr;
    fe.tick = (!fused && do_adam) ? ctr : nullptr;
    const dim3 feat_grid((B + kBlock - 1) / kBlock, (J + fe.jchunk - 1) / fe.jchunk, P * L);
    const size_t slab = (size_t)P * S * L * J;
    GemmArgs ga;
    ga.S = S; ga.L = L; ga.J = J; ga.B = B; ga.SK = SK; ga.nsel = want_dell ? 2 : 1;
    ga.W = nz->w; ga.Phi = ws->Phi; ga.dPhi = ws->dPhi; ga.F0 = ws->F0; ga.H = ws->H; ga.slab = slab;
    ga.dbg = 0;
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
    pa.eps = nz->eps; pa.eps2 = nz->eps2; pa.R = ws->R; pa.f = out->f; pa.G = ws->G; pa.part = ws->part;
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
    const double lik_scale = pb->alpha / (double)d->S_total;
    FinalArgs fa;
    fa.M = M; fa.L = L; fa.NC = NC; fa.nblk = P ? vg_loglik_blocks_per_problem(S, N) : 0; fa.part_len = vg_part_len(d);
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
    const size_t lds_rows = ((size_t)2 * Mz * ((Mz + 2) & ~1) + (size_t)4 * kRowTile * Mz + Mz + kRowTile) * sizeof(double);
    const size_t lds_cov_b = lds_cov > lds_rows ? lds_cov : lds_rows;
    // path kernels: operands + (when it fits) the raw split-K slabs of the prior draws
    const size_t raw_f = SK == 1 ? 0 : (size_t)SK * SC * J * sizeof(float);      // one slab lands in place
    size_t lds_pf = ((size_t)Mz * (Mz + 1) + (size_t)Mz * N + (size_t)3 * SC * Mz + Mz + (size_t)SC * J + 6 * 4) * sizeof(float);
    size_t lds_pb = ((size_t)4 * N * Mz + (size_t)2 * Mz * Mz + (size_t)SC * N + (size_t)2 * SC * J +
                     (size_t)8 * SC * Mz + 10 * 4) * sizeof(float);
    const bool raw_fwd = lds_pf + raw_f <= 64 * 1024, raw_bwd = lds_pb + 2 * raw_f <= 160 * 1024;
    if (raw_fwd) lds_pf += raw_f;
    if (raw_bwd) lds_pb += 2 * raw_f;
    size_t lds_fin = ((size_t)Mz * Mz + Mz + 1) * sizeof(double) + ((size_t)Mz * Mz + 4) * sizeof(float);
    const size_t raw_fin = (size_t)NC * (Mz + Mz * Mz) * sizeof(float);
    const bool fin_dma = lds_fin + raw_fin <= 96 * 1024;
    if (fin_dma) lds_fin += raw_fin;
    fa.dma = fin_dma ? 1 : 0;
    const size_t lds_s1 = lds_cov_a > lds_fin ? lds_cov_a : lds_fin;
    const void* fn_cov_b = backward ? (const void*)cov_b_kernel<true> : (const void*)cov_b_kernel<false>;
    const bool k8 = (B / SK) % 128 == 0;      // K-slice in passes of 8 steps of 16: a pass's operands in one request
    // K-slices of a multiple of 128 and enough samples: operands through LDS by DMA (needs 59 KB per workgroup)
    const bool glds = (B / SK) % kGK == 0 && S >= 48 && !(what & VGPMP_GEMM_DIRECT);      // 64-row tiles: few samples waste them
    const void* fn_s2 = glds ? (backward ? (const void*)stage2_kernel<true, -1> : (const void*)stage2_kernel<false, -1>)
                      : backward ? (k8 ? (const void*)stage2_kernel<true, 8> : (const void*)stage2_kernel<true, 0>)
                                 : (k8 ? (const void*)stage2_kernel<false, 8> : (const void*)stage2_kernel<false, 0>);
    const size_t lds_s2 = glds && kGemmLds > lds_cov_b ? kGemmLds : lds_cov_b;
    if (SC != 8) return VGPMP_E_SHAPE;
#define VG_PICK(kernel, raw)                                                                                        \
    (SK == 1 ? (raw ? (const void*)kernel<1, true> : (const void*)kernel<1, false>)                                 \
     : SK == 2 ? (raw ? (const void*)kernel<2, true> : (const void*)kernel<2, false>)                               \
     : SK == 4 ? (raw ? (const void*)kernel<4, true> : (const void*)kernel<4, false>)                               \
               : (raw ? (const void*)kernel<8, true> : (const void*)kernel<8, false>))
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
    if (split_bwd) {
        fn_pb = Mz == 32 ? (SK == 2 ? (const void*)paths_bwd_split<2, 32> : SK == 4 ? (const void*)paths_bwd_split<4, 32>
                                                                                        : (const void*)paths_bwd_split<8, 32>)
                         : (SK == 2 ? (const void*)paths_bwd_split<2> : SK == 4 ? (const void*)paths_bwd_split<4>
                                                                                  : (const void*)paths_bwd_split<8>);
        lds_pb = lds_pbs;
    }
    if (backward && (rc = set_dyn_lds(fn_pb, lds_pb))) return rc;      // forward-only calls never launch the reverse pass
    if (fused) {
        if ((rc = set_dyn_lds((const void*)stage1_kernel<false>, lds_s1))) return rc;
        if ((rc = set_dyn_lds((const void*)stage1_kernel<true>, lds_s1))) return rc;
        if ((rc = set_dyn_lds(fn_s2, lds_s2))) return rc;
        if ((rc = set_dyn_lds(fn_s3, lds_pf))) return rc;
    } else {
        if ((rc = set_dyn_lds((const void*)cov_a_kernel, lds_cov_a))) return rc;
        if (glds && (rc = set_dyn_lds((const void*)prior_gemm_lds_kernel, kGemmLds))) return rc;
        if ((rc = set_dyn_lds(fn_cov_b, lds_cov_b))) return rc;
        if ((rc = set_dyn_lds(fn_pf, lds_pf))) return rc;
    }
    if ((rc = set_dyn_lds((const void*)final_kernel, lds_fin))) return rc;
    // medium batches: merged launches (not while profiling stage by stage, not with the shared stage launches)
    // measured on config 2 shapes: 5 problems 197 -> 182 us per step, 9: 218 -> 205, 16: equal, 24 and 64: 2-6 % slower
    // (the chip is full by then, the merged kernels only cost registers) -- hence the bound
    // few samples, four K-slices: features inside the GEMM (prior_fused_small_kernel)
    const bool fused_small = !fused && SK == 4 && S <= 32 && (B % 64) == 0 && !(what & VGPMP_GEMM_DIRECT);
    // (bounds measured at S = 128; the work per problem scales with the samples)
    const long long pls = (long long)P * L * S;
    const bool mid = !fused && !ev && !pb->ind && (tiled_gemm || fused_small) && backward && !(what & VGPMP_NO_FUSE) &&
                     pls <= (long long)kMid2MaxPL * 128;
    const bool mid_gemm = tiled_gemm && pls <= (long long)kMidMaxPL * 128;      // cov_b beside the GEMM only while the chip is not full
    const size_t lds_tg1 = (size_t)2 * (kTS + kTJ) * kTLd * sizeof(float);
    const size_t lds_midC = lds_cov_b > lds_tg1 ? lds_cov_b : lds_tg1;
    const void* fn_midC = backward ? (const void*)mid_cov_b_gemm_kernel<true> : (const void*)mid_cov_b_gemm_kernel<false>;
    if (mid) {
        if ((rc = set_dyn_lds((const void*)mid_cov_a_rng_kernel, lds_cov_a))) return rc;
        if ((rc = set_dyn_lds(fn_midC, lds_midC))) return rc;
        if ((rc = set_dyn_lds((const void*)mid_hyper_final_kernel, lds_fin))) return rc;
    }
    const dim3 cov_b_grid(3 + (N + kRowTile - 1) / kRowTile, L, P);
    const uint32_t eps_gx = (2u * (uint32_t)S * Mz * L + kBlock - 1) / kBlock;
    const uint32_t basis_gx = ((uint32_t)L * B + kBlock - 1) / kBlock;
    const uint32_t w_gx = (((uint32_t)S * L * B >> 2) + kBlock - 1) / kBlock;
    auto launch = [&](const void* fn, dim3 grid, void* arg, size_t lds) -> int {
        void* kargs[] = {arg};
        return (int)hipLaunchKernel(fn, grid, dim3(kBlock), kargs, lds, st);
    };
    auto launch_fused_small = [&](hipEvent_t g0, hipEvent_t g1) {
        FusedPriorArgs fp;
        fp.S = S; fp.L = L; fp.J = J; fp.N = N; fp.D = L; fp.B = B; fp.want_dell = want_dell ? 1 : 0;
        fp.X = pb->X; fp.Zy = zy; fp.zy_stride = zy_stride; fp.raw_ell = params->raw_ell; fp.raw_var = params->raw_var;
        fp.omega = nz->omega; fp.beta = nz->beta; fp.W = nz->w; fp.F0 = ws->F0; fp.H = ws->H; fp.slab = slab;
        fp.tick = fe.tick;
        const dim3 fgrid(P * L, (J + kFNT * 16 - 1) / (kFNT * 16));
#define VG_FUSED_SMALL(MT_, DM_)                                            \
    (want_dell ? (void)hipExtLaunchKernelGGL((prior_fused_small_kernel<MT_, DM_, true>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp)                                  \
               : (void)hipExtLaunchKernelGGL((prior_fused_small_kernel<MT_, DM_, false>), fgrid, dim3(kBlock), 0, st, g0, g1, \
                             0, fp))
        if (L <= 8) { if (S <= 16) VG_FUSED_SMALL(1, 8); else VG_FUSED_SMALL(2, 8); }
        else { if (S <= 16) VG_FUSED_SMALL(1, 16); else VG_FUSED_SMALL(2, 16); }
#undef VG_FUSED_SMALL
    };
    auto launch_final = [&]() -> int { return launch((const void*)final_kernel, dim3(L, P), &fa, lds_fin); };
    // likelihood constants as variables (vgpmp_lik_params): effective values from the raw ones at the start of the call
    LikUpdArgs lu;
    if (lk) {
        LikConstArgs lc;
        lc.raw_alpha = lk->raw_alpha; lc.raw_sigma = lk->raw_sigma; lc.sc = lsc; lc.inv_s = 1.0 / (double)d->S_total;
        hipLaunchKernelGGL(lik_consts_kernel, dim3(P), dim3(VGPMP_MAX_SPHERES), 0, st, lc);
        lu.rb = rb; lu.lik_partial = ws->lik_partial; lu.sig_partial = lsc.sig_partial; lu.nblk = 0;
        lu.inv_s = lc.inv_s;
        lu.raw_alpha = lk->raw_alpha; lu.raw_sigma = lk->raw_sigma; lu.m_alpha = lk->m_alpha; lu.v_alpha = lk->v_alpha;
        lu.m_sigma = lk->m_sigma; lu.v_sigma = lk->v_sigma; lu.g_alpha = lk->g_alpha; lu.g_sigma = lk->g_sigma;
        lu.sc = lsc; lu.do_adam = do_adam ? 1 : 0; lu.trainable = trainable;
        lu.ctr = do_adam ? ctr : nullptr; lu.lr = lr; lu.lr_t = 0.0;
    }

    for (int i = 0; i < num_steps; ++i) {
        const bool first = i == 0, more = i + 1 < num_steps;
        const uint32_t step_i = step + (uint32_t)i;
        if (ind && (rc = vg_launch_inducing_build(d, ind, st))) return rc;       // Zy = [0; 1; Z(raw_Z)] of every problem
        if (fused) {
            // noise of the first step of a call: everything up front; afterwards eps rides in stage 1 and the
            // prior noise of step i was drawn by stage 3 of step i-1
            if (gen && first && (rc = vg_launch_rng(d, nz, seed, problem_base, step_i, ctr, st))) return rc;
            // steps after the first of a call: the hyper-parameter update of the previous step is a prologue of the
            // cov_a and feature roles, its q_mu / q_sqrt update (final) another role of the same launch
            const bool prologue = !first && backward;
            const double lr_prev = do_adam ? adam_lr_t(lr, adam_t + i - 1 > 0 ? adam_t + i - 1 : 1) : 0.0;
            hyp.lr_t = lr_prev;
            Stage1Args s1;
            s1.skip = skip1;
            s1.cov = ca; s1.fin = fa; s1.feat = fe;
            s1.fin.lr_t = lr_prev;
            s1.cov.hy = hyp; s1.feat.hy = hyp;
            s1.cov.prologue = prologue ? 1 : 0;
            s1.rng = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
            s1.n_cov = L * P;
            s1.n_fin = first ? 0 : L * P;
            s1.eps_gx = (int)eps_gx;
            s1.n_eps = (gen && !first) ? (int)eps_gx * P : 0;
            s1.feat_gx = (int)feat_grid.x; s1.feat_gy = (int)feat_grid.y;
            const unsigned n1 = s1.n_cov + s1.n_fin + s1.n_eps + feat_grid.x * feat_grid.y * feat_grid.z;
            if ((rc = launch(prologue ? (const void*)stage1_kernel<true> : (const void*)stage1_kernel<false>, dim3(n1), &s1, lds_s1)))
                return rc;
            Stage2Args s2;
            s2.skip = skip2;
            s2.cov = ca; s2.gemm = ga;
            s2.cov.hy = hyp; s2.cov.commit = prologue ? 1 : 0;
            s2.cov_roles = (int)cov_b_grid.x; s2.n_cov = (int)(cov_b_grid.x * cov_b_grid.y * cov_b_grid.z);
            s2.gemm_gx = (int)gemm_grid.x; s2.gemm_gy = (int)gemm_grid.y;
            const int n_gemm = (int)(gemm_grid.x * gemm_grid.y * gemm_grid.z);
            s2.gemm_per_xcd = (s2.n_cov % 8 == 0 && n_gemm % 8 == 0) ? n_gemm / 8 : 0;
            if ((rc = launch(fn_s2, dim3(s2.n_cov + gemm_grid.x * gemm_grid.y * gemm_grid.z), &s2, lds_s2))) return rc;
            Stage3Args s3;
            s3.skip = skip3;
            s3.path = pa;
            // the counter has ticked in stage 2: it already names the next step
            s3.rng = make_rng_args(d, nz, seed, problem_base, step_i + 1u, ctr, 0u);
            s3.n_path = NC * pa.nsplit * L * P;
            s3.basis_gx = (int)basis_gx; s3.w_gx = (int)w_gx;
            s3.n_basis = (gen && more) ? (int)basis_gx * P : 0;
            const unsigned n3 = s3.n_path + s3.n_basis + ((gen && more) ? w_gx * P : 0u);
            if ((rc = launch(fn_s3, dim3(n3), &s3, lds_pf))) return rc;
        } else if (mid) {
            // cov_a | noise draws;  features;  cov_b | tiled GEMM
            MidAArgs ma;
            ma.cov = ca;
            ma.rng = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
            ma.n_cov = L * P; ma.basis_gx = (int)basis_gx; ma.n_basis = gen ? (int)basis_gx * P : 0;
            const uint32_t n_thr = (ma.rng.nW >> 2) + 2 * ma.rng.nE;
            ma.n_gx = (int)((n_thr + kBlock - 1) / kBlock);
            const unsigned nA = ma.n_cov + ma.n_basis + (gen ? (unsigned)ma.n_gx * P : 0u);
            if ((rc = launch((const void*)mid_cov_a_rng_kernel, dim3(nA), &ma, lds_cov_a))) return rc;
            if (!fused_small) hipLaunchKernelGGL(features_kernel, feat_grid, dim3(kBlock), 0, st, fe);
            MidCArgs mc;
            mc.cov = ca; mc.gemm = tga;
            mc.cov_roles = (int)cov_b_grid.x; mc.n_cov = (int)(cov_b_grid.x * cov_b_grid.y * cov_b_grid.z);
            mc.gemm_gx = (J + kTJ - 1) / kTJ; mc.gemm_gy = (S + kTS - 1) / kTS;
            const unsigned nC = mc.n_cov + (unsigned)mc.gemm_gx * mc.gemm_gy * P * L * ga.nsel;
            if (mid_gemm) {
                if ((rc = launch(fn_midC, dim3(nC), &mc, lds_midC))) return rc;
            } else if (fused_small) {
                if ((rc = launch(fn_cov_b, cov_b_grid, &ca, lds_cov_b))) return rc;
                launch_fused_small(nullptr, nullptr);
            } else {
                if ((rc = launch(fn_cov_b, cov_b_grid, &ca, lds_cov_b))) return rc;
                const size_t lds_tg = (size_t)2 * (kTS + kTJ) * kTLd * sizeof(float);
                if ((rc = set_dyn_lds((const void*)prior_gemm_tiled_kernel<1>, lds_tg))) return rc;
                hipLaunchKernelGGL(prior_gemm_tiled_kernel<1>, dim3((J + kTJ - 1) / kTJ, (S + kTS - 1) / kTS, P * L * ga.nsel),
                                   dim3(kBlock), lds_tg, st, tga);
            }
            if ((rc = launch(fn_pf, dim3(NC * pa.nsplit, L, P), &pa, lds_pf))) return rc;
        } else {
            mark();
            hipLaunchKernelGGL(cov_a_kernel, dim3(L, P), dim3(kCovThreads), lds_cov_a, st, ca);
            if ((rc = launch(fn_cov_b, cov_b_grid, &ca, lds_cov_b))) return rc;
            if (what & VGPMP_COV_ONLY) return (int)hipGetLastError();     // Kuu, Cholesky, q_sqrt, A, per-latent KL: done
            mark();
            // large batches drawing their own noise: W and the features are formed inside the GEMM (prior_fused_batch_kernel);
            // trainable inducing locations read W back (inducing.hip), so they keep it in memory
            const bool fbatch = gen && tiled_gemm && !fused_small && !ind && !(what & VGPMP_NO_FUSE_PRIOR);
            if (gen) {
                RngArgs r = make_rng_args(d, nz, seed, problem_base, step_i, ctr, 0u);
                if (fbatch) r.nW = 0;                    // omega, beta, eps, eps2 only
                hipLaunchKernelGGL(rng_basis_kernel, dim3((r.L * r.B + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
                const uint32_t nthr = (r.nW >> 2) + 2 * r.nE;
                hipLaunchKernelGGL(rng_normals_kernel, dim3((nthr + kBlock - 1) / kBlock, P), dim3(kBlock), 0, st, r);
            }
            mark();
            if (!fused_small && !fbatch) hipLaunchKernelGGL(features_kernel, feat_grid, dim3(kBlock), 0, st, fe);
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
                const size_t lds_fb = ((size_t)kTS * kFBLd + (size_t)2 * kTJ * kFBLd + (size_t)kTJ * dm + (size_t)2 * kFBK * (dm + 4)) * sizeof(float);
                const dim3 fb_grid((J + kTJ - 1) / kTJ, (S + kTS - 1) / kTS, P * L);
#define VG_FB(DELL_, DM_) hipExtLaunchKernelGGL((prior_fused_batch_kernel<DELL_, DM_>), fb_grid, dim3(kBlock), lds_fb, st, g0, g1, 0, fb)
                if (want_dell) { if (dm == 8) VG_FB(true, 8); else VG_FB(true, 16); }
                else { if (dm == 8) VG_FB(false, 8); else VG_FB(false, 16); }
#undef VG_FB
                if (fe.tick) hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, st, fe.tick);      // the feature kernel's tick
            } else if (fused_small) {      // features formed inside the GEMM (few samples: Phi / dPhi traffic is the cost)
                launch_fused_small(g0, g1);
            } else if (tiled_gemm) {
                const int mt = 1;      // 128-sample tiles (mt = 2) measured slower: 90 vs 98 TF/s at 64 problems
                const size_t lds_tg = (size_t)2 * (kTS * mt + kTJ) * kTLd * sizeof(float);
                const void* fn_tg = mt == 2 ? (const void*)prior_gemm_tiled_kernel<2> : (const void*)prior_gemm_tiled_kernel<1>;
                if ((rc = set_dyn_lds(fn_tg, lds_tg))) return rc;
                const dim3 tg_grid((J + kTJ - 1) / kTJ, (S + kTS * mt - 1) / (kTS * mt), P * L * ga.nsel);
                if (mt == 2) hipExtLaunchKernelGGL(prior_gemm_tiled_kernel<2>, tg_grid, dim3(kBlock), lds_tg, st, g0, g1, 0, tga);
                else hipExtLaunchKernelGGL(prior_gemm_tiled_kernel<1>, tg_grid, dim3(kBlock), lds_tg, st, g0, g1, 0, tga);
            } else if (glds)
                hipExtLaunchKernelGGL(prior_gemm_lds_kernel, gemm_grid, dim3(kBlock), kGemmLds, st, g0, g1, 0, ga);
            else
                hipExtLaunchKernelGGL(prior_gemm_kernel<0>, gemm_grid, dim3(kBlock), 0, st, g0, g1, 0, ga);
            mark();
            if ((rc = launch(fn_pf, dim3(NC * pa.nsplit, L, P), &pa, lds_pf))) return rc;
            mark();
        }
        // ---- likelihood forward + reverse (fk_sdf.hip)
        int nblk = 0;
        rc = vg_launch_loglik_paths(rb, sdf, out->f, P, S, L, N, (float)(-lik_scale), ws->G, out->logp, ws->lik_partial,
                                    &nblk, st, ev ? ev[VG_NUM_STAGES + 1] : nullptr, ev ? ev[VG_NUM_STAGES + 2] : nullptr,
                                    lk ? lsc.alpha_eff : nullptr, lk ? lsc.sigma_eff : nullptr,
                                    lk ? lsc.sig_partial : nullptr, (what & VGPMP_LIK_LDS_STATE) ? 2 : (what & VGPMP_LIK_LANES) ? 1 : 0);
        if (rc) return rc;
        fa.nblk = nblk;
        mark();
        if (!backward) {
            hipLaunchKernelGGL(elbo_pieces_kernel, dim3(P), dim3(kBlock), 0, st, L, nblk, ws->lik_partial, ws->kl_l, lik_scale,
                               pb->kl_scale, out->lik, out->kl, fa.alpha_fin);
            return (int)hipGetLastError();
        }
        // ---- reverse of the path assembly (+ hyper-parameter update), then (here or in the next stage 1) the rest
        if ((rc = launch(fn_pb, dim3(split_bwd ? 2 * NC : NC, L, P), &pa, lds_pb))) return rc;
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
            hipLaunchKernelGGL(lik_update_kernel, dim3(P), dim3(VGPMP_MAX_SPHERES), 0, st, lu);
        }
        mark();
        if (mid) {
            MidGArgs mg;
            hy.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            fa.lr_t = hy.lr_t;
            mg.hy = hy; mg.fin = fa; mg.n_hyper = P;
            if ((rc = launch((const void*)mid_hyper_final_kernel, dim3(P + L * P), &mg, lds_fin))) return rc;
        } else if (!(fused && more)) {      // otherwise both ride in stage 1 of the next step
            hy.lr_t = do_adam ? adam_lr_t(lr, adam_t + i > 0 ? adam_t + i : 1) : 0.0;
            fa.lr_t = hy.lr_t;
            hipLaunchKernelGGL(hyper_kernel, dim3(P), dim3(64), 0, st, hy);
            if ((rc = launch_final())) return rc;
        }
        mark();
    }
    return (int)hipGetLastError();
}
