// Random Fourier features and the prior GEMMs on the f32 MFMA pipe.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once
#include <type_traits>

namespace {

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
    double re, rv;
    if constexpr (PRO) {
        const HyperState o = hyper_update_wave(a.hy, pl);      // whole waves: before any lane leaves
        re = o.raw_ell; rv = o.raw_var;
    }
    if (b >= B) return;
    if constexpr (!PRO) {
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
constexpr int kNT = 3;     // 16-column tiles per wave

struct GemmArgs {
    int S, L, J, B, SK, nsel;
    const float *W, *Phi, *dPhi;
    float *F0, *H;
    size_t slab;
    int dbg;                 // measurement builds: 1 no stores, 2 no loads, 3 no MFMA
    int w_exact;             // W was drawn by the library's generator: float16 values, no low half (the f16 form skips its MFMA)
    const double *ell, *var; // [P,L] of this step (stage A's): the f16 form takes the features' constant factors out of its operands
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
__global__ __launch_bounds__(kBlock, 2) void prior_gemm_kernel(GemmArgs a) { prior_gemm_body<KS>(a, blockIdx.x, blockIdx.y, blockIdx.z); }

// The same tile with its operands staged through LDS by DMA in passes of 128 K (the form stage 2 uses): whole
// 512-byte rows per request instead of the 16 rows x 64 bytes a fragment-shaped load touches, all requests of a
// pass in flight together, fragments by ds_read_b128.  LDS rows are padded to 132 floats (33 units of 16 bytes;
// the pad unit repeats the row's last one): the 16-byte fragment reads of a 16-row group then fall on distinct
// bank slots.  LDS: (64 + 48) x 132 x 4 = 59 KB per workgroup.
constexpr int kGK = 128, kGLd = kGK + 4, kGRows = 64 + 16 * kNT;
constexpr size_t kGemmLds = (size_t)kGRows * kGLd * sizeof(float);

// F16: the products on the f16 matrix pipe (operands read from the same float32 LDS tiles, split into f16 halves in registers,
// hi lo + lo hi + hi hi in float32 accumulators: gp_prior_split.h) -- for a few problems of MANY samples (the sample-sharded
// job on one rank: 1024), where this role is the step: 24 float32 MFMAs of 32 cycles per 32 k and three column tiles become 9 f16
// MFMAs and 64 vector instructions.  Fewer samples keep the float32 form (the tests' bitwise reference; the role is not the
// longest of its launch there).
template <bool F16 = false>
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
    float sc_in = 1.0f, sc_out = 1.0f;                       // (F16) the features' constant factor, out of the operands / back in
    if constexpr (F16) {
        const size_t pl = (size_t)p * L + l;
        const float ell = (float)a.ell[pl], cf = __builtin_amdgcn_sqrtf(2.0f * (float)a.var[pl] / (float)B);
        sc_out = sel == 0 ? cf : cf / (ell * ell);
        sc_in = 1.0f / sc_out;
    }
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
        if constexpr (F16) {
            // a lane's fragment: 8 consecutive k of its row (two 16-byte reads), as f16 halves.  The stored features carry their
            // constant factors (Phi = c cos, dPhi = c / ell^2 sin (x . omega): a lengthscale of 0.01 puts the latter beyond f16's
            // 65504): `sc` takes them out -- cos and sin (x . omega) are f16-safe -- and the accumulators get them back at the end
            auto frag = [&](const float* row, int ks2, float sc, vg_h8& hi, vg_h8& lo) {
                const vg_f32x4 x0 = vg_scale4(*reinterpret_cast<const vg_f32x4*>(row + 32 * ks2 + 8 * g), sc);
                const vg_f32x4 x1 = vg_scale4(*reinterpret_cast<const vg_f32x4*>(row + 32 * ks2 + 8 * g + 4), sc);
                vg_h4 h0, l0, h1, l1;
                vg_split4(x0, h0, l0);
                vg_split4(x1, h1, l1);
                hi = (vg_h8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                lo = (vg_h8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            };
#pragma unroll
            for (int ks2 = 0; ks2 < kGK / 32; ++ks2) {
                vg_h8 ah, al;
                frag(As + (wave * 16 + r) * kGLd, ks2, 1.0f, ah, al);
                // (weights drawn by the library's generator are float16 values -- vgpmp_device.h, "The W stream" --: `al` is zero then)
                const bool w_lo = a.w_exact == 0;
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    vg_h8 bh, bl;
                    frag(Bs + (16 * t + r) * kGLd, ks2, sc_in, bh, bl);
                    if (w_lo) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
                }
            }
            continue;
        }
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
            if (s < S) vg_stream(Out + (((size_t)p * S + s) * L + l) * J + jc, F16 ? acc[t][q] * sc_out : acc[t][q]);
        }
    }
    VG_T(bx == 0 && by == 0 && bz == 0, 242);
    VG_T(bx == 2 && by == 1 && l == L - 1 && sk == SK - 1 && sel == nsel - 1, 245);
}

__global__ __launch_bounds__(kBlock, 2) void prior_gemm_lds_kernel(GemmArgs a) {
    extern __shared__ float gemm_lds[];
    prior_gemm_lds_body<false>(a, gemm_lds, blockIdx.x, blockIdx.y, blockIdx.z);
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
__global__ __launch_bounds__(kBlock, 2) void prior_gemm_tiled_kernel(TiledGemmArgs ta) {
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
// MT = 2: 128 sample rows per workgroup (two 16-row tiles per wave) -- the features of a K step, which do not depend on the
// sample, are formed once for 128 rows instead of once per 64, and every B fragment read from LDS feeds two MFMAs; 36
// accumulator tiles per wave (144 registers): one workgroup per CU.
template <bool DELL, int DM, int MT = 1>      // d/d ell wanted; joint-space extent padded to DM (8 or 16); 16-row tiles per wave
__device__ __forceinline__ void prior_fused_batch_body(const FusedBatchArgs& a, float* fb_lds, int bx, int by, int bz) {
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = bz % L, p = bz / L;
    constexpr int kRows = kTS * MT;
    const int s0 = by * kRows, j0 = bx * kTJ;
    const size_t pl = (size_t)p * L + l;
    float* As = fb_lds;                                  // [64 MT][kFBLd]      W tile
    float* Bs = As + kRows * kFBLd;                        // [2][144][kFBLd]     Phi, dPhi tiles
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
    uint32_t wbase[MT];                                  // W counter of (row, column 0): eight columns each; rows wrow, wrow + 64
#pragma unroll
    for (int m = 0; m < MT; ++m) wbase[m] = (a.wOff + ((uint32_t)min(s0 + wrow + kTS * m, S - 1) * L + l) * (uint32_t)B) >> 3;
    const int kcol = tid & 15, jg = tid >> 4;
    const int nom = kFBK * D;
    const bool is_om = tid < nom, is_bt = tid >= nom && tid < nom + kFBK;
    const int orow = is_om ? tid / D : tid - nom, ocol = is_om ? tid - orow * D : DM;      // phase sits behind the row
    const float* osrc = is_om ? a.omega + pl * B * D + tid : a.beta + pl * B + (tid - nom);
    const int ostep = is_om ? kFBK * D : kFBK;
    float onext = (is_om || is_bt) ? osrc[0] : 0.f;
    vg_f32x4 accF[MT][kTJ / 16], accH[MT][kTJ / 16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) { accF[m][t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f}; accH[m][t] = accF[m][t]; }
    const int r = lane & 15, g = lane >> 4;
    __syncthreads();
    if (is_om || is_bt) oms[orow * kOLd + ocol] = onext;
    __syncthreads();
    int ob = 0;
    for (int k0 = 0; k0 < B; k0 += kFBK) {
        // ---- generate the K step's operands (the next step's frequencies are requested first, stored last)
        {
            if ((is_om || is_bt) && k0 + kFBK < B) onext = osrc[(size_t)(k0 / kFBK + 1) * ostep];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // (a W counter holds eight weights: the threads of quads 2 h and 2 h + 1 draw the same one and keep a half each --
                //  this float32 form is the measurement / reference form of the large-batch draw, gp_prior_split.h the product)
                float z[8];
                vg_w8_f32(wbase[m] + (uint32_t)((k0 >> 3) + (wq >> 1)), key, kWTable, z);
                const float4 w4 = (wq & 1) ? make_float4(z[4], z[5], z[6], z[7]) : make_float4(z[0], z[1], z[2], z[3]);
                *reinterpret_cast<float4*>(As + (wrow + kTS * m) * kFBLd + 4 * wq) = w4;
            }
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
        // ---- 2 x 9 tiles of 16 x 16 per row tile, four k-interleaved MFMAs each
        {
            float4 a4[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const float4*>(As + (kTS * m + wave * 16 + r) * kFBLd + 4 * g);
#pragma unroll
            for (int t = 0; t < kTJ / 16; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(Bs + (t * 16 + r) * kFBLd + 4 * g);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].x, b4.x, accF[m][t], 0, 0, 0);
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].y, b4.y, accF[m][t], 0, 0, 0);
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].z, b4.z, accF[m][t], 0, 0, 0);
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].w, b4.w, accF[m][t], 0, 0, 0);
                }
                if (DELL) {
                    const float4 d4 = *reinterpret_cast<const float4*>(Bs + (kTJ + t * 16 + r) * kFBLd + 4 * g);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].x, d4.x, accH[m][t], 0, 0, 0);
                        accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].y, d4.y, accH[m][t], 0, 0, 0);
                        accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].z, d4.z, accH[m][t], 0, 0, 0);
                        accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].w, d4.w, accH[m][t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        ob ^= 1;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) {
            const int jc = j0 + 16 * t + r;
            if (jc >= J) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + kTS * m + wave * 16 + g * 4 + q;
                if (s < S) {
                    vg_stream(a.F0 + (((size_t)p * S + s) * L + l) * J + jc, accF[m][t][q]);
                    if (DELL) vg_stream(a.H + (((size_t)p * S + s) * L + l) * J + jc, accH[m][t][q]);
                }
            }
        }
}
template <bool DELL, int DM, int MT = 1>
__global__ __launch_bounds__(kBlock, 2) void prior_fused_batch_kernel(FusedBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    prior_fused_batch_body<DELL, DM, MT>(a, fb_lds, blockIdx.x, blockIdx.y, blockIdx.z);
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
constexpr int kFNT = 2;     // column tiles per workgroup     // config 3 (55 problems, S = 7, J = 96): 234 / 198 / 194 / 210 us per step with 5 / 3 / 2 / 1 -- more, lighter waves
template <int MT, int DM, bool DELL>    // 16-row sample tiles; joint-space extent DM = D for 6 and 7 joints, else padded to 8 or 16; d/d ell wanted
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2)))
// (at least two waves per SIMD = at most 256 registers: the compiler then keeps the MFMA accumulators in ordinary VGPRs -- with 512
//  allowed it puts them in AGPRs and pays a v_accvgpr_read per projection element plus moves per loop rotation: 134 -> 98 vector
//  instructions per 32 bases, config 3 164.6 -> 161.7 us per step)
void prior_fused_small_kernel(FusedPriorArgs a) {
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
    const float inv_ell = 1.0f / ell, c = __builtin_amdgcn_sqrtf(2.0f * var / (float)B), c_ell2 = c * inv_ell * inv_ell;
    const float rev_ell = inv_ell * 0.15915494309189535f;
    const int r = lane & 15, g = lane >> 4;
    const int kchunk = B / 4, kbeg = sk * kchunk;
    vg_f32x4 accF[MT][kFNT], accH[MT][kFNT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kFNT; ++t) { accF[m][t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f}; accH[m][t] = accF[m][t]; }
    const float* wrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wrow[m] = a.W + (((size_t)p * S + min(16 * m + r, S - 1)) * L + l) * B;
    // The projections x . omega of a pass (16 bases x kFNT tiles of 16 points) are matrix products themselves:
    //   proj[base 4 g + q][point r] = sum_d omega[base][d] pts[point][d]   =   accumulator element q of lane (r, g)
    // of 16 x 16 x 4 MFMAs with A = omega[base k0 + r][4 i + g] and B = pts[16 t + r][4 i + g] -- exactly the (point, four bases)
    // layout the feature fragments below need.  kDQ MFMAs per tile replace 4 D multiply-adds per lane, and a lane loads kDQ
    // frequencies per pass instead of 4 D.  (Columns beyond D: the points' zero padding meets a clamped, finite frequency.)
    constexpr int kDQ = (DM + 3) / 4;
    float pb[kFNT][kDQ];
#pragma unroll
    for (int t = 0; t < kFNT; ++t)
#pragma unroll
        for (int i = 0; i < kDQ; ++i) pb[t][i] = 4 * i + g < DM ? pts[16 * t + r][min(4 * i + g, DM - 1)] : 0.f;
    int ooff[kDQ];
#pragma unroll
    for (int i = 0; i < kDQ; ++i) ooff[i] = min(4 * i + g, D - 1);
    const float* orow = a.omega + ((size_t)pl * B + kbeg + r) * D;              // this lane's basis of the pass
    const float* brow = a.beta + (size_t)pl * B + kbeg + 4 * g;                 // phases of the four bases it forms features of
    // operands of a pass, requested one pass ahead (kDQ frequencies, 4 phases, the W fragments: a dozen registers)
    float oa[kDQ], oa_n[kDQ];
    vg_f32x4 bt, bt_n, a4[MT], a4_n[MT];
    auto fetch = [&](int k, float (&o)[kDQ], vg_f32x4& bb, vg_f32x4 (&aa)[MT]) {
        const int kk = min(k, kchunk - 16);              // (the look-ahead of the last pass re-reads it)
#pragma unroll
        for (int i = 0; i < kDQ; ++i) o[i] = orow[(size_t)kk * D + ooff[i]];
        bb = *reinterpret_cast<const vg_f32x4*>(brow + kk);
#pragma unroll
        for (int m = 0; m < MT; ++m) aa[m] = *reinterpret_cast<const vg_f32x4*>(wrow[m] + kbeg + kk + 4 * g);
    };
    // one pass: 16 bases x kFNT column tiles.  One wave per SIMD issues in order: the features of column tile t + 1 are formed
    // between the products of tile t (independent work next to each other in the instruction stream), not after them
    auto pass = [&](const float (&o)[kDQ], const vg_f32x4& bb, const vg_f32x4 (&aa)[MT]) {
        float ph[2][4], dh[2][4];
        vg_f32x4 btr = vg_scale4(bb, 0.15915494309189535f);        // phases in revolutions
        auto feats = [&](int t, float (&pc)[4], float (&dc)[4]) {
            vg_f32x4 proj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < kDQ; ++i) proj = __builtin_amdgcn_mfma_f32_16x16x4f32(o[i], pb[t][i], proj, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // phase in revolutions by ONE fused multiply-add; the factors c and c / ell^2 go to the accumulators at the end
                const float rev = __builtin_amdgcn_fractf(fmaf(proj[q], rev_ell, btr[q]));
                pc[q] = __builtin_amdgcn_cosf(rev);
                dc[q] = __builtin_amdgcn_sinf(rev) * proj[q];
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
                    accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m][q], ph[cb][q], accF[m][t], 0, 0, 0);
                    if (DELL) accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m][q], dh[cb][q], accH[m][t], 0, 0, 0);
                }
        }
    };
    fetch(0, oa, bt, a4);
    int k = 0;
    for (; k + 32 <= kchunk; k += 32) {
        fetch(k + 16, oa_n, bt_n, a4_n);
        pass(oa, bt, a4);
        fetch(k + 32, oa, bt, a4);
        pass(oa_n, bt_n, a4_n);
    }
    if (k < kchunk) pass(oa, bt, a4);      // (a slice of 16 bases: B = 64)
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
                vg_stream(F0 + o, c * accF[m][t][q]);
                if (DELL) vg_stream(H + o, c_ell2 * accH[m][t][q]);
            }
        }
}

}  // namespace
