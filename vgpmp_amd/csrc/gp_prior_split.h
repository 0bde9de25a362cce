// The prior draws of a large batch on the f16 matrix pipe: prior_fused_split_kernel.
// Private part of gp_path.hip (one translation unit).
//
// Same job as prior_fused_batch_kernel (gp_prior.h): F0 = W Phi^T and H = W (dPhi/dell)^T with W from the Philox
// generator and the random features from sin / cos, all formed inside the GEMM's K loop (models/vgpmp.py:281-282: the
// `temporary_paths` draw of the reference, restated in oracle.elbo_forward / rff_features).  What changes is the
// arithmetic of the products.  On gfx950 v_mfma_f32_16x16x4_f32 runs at the FP32 vector rate and shares that datapath
// (generation and products serialise: profiles/r02/final/sq_prior_fused_config5.txt), while v_mfma_f32_16x16x32_f16
// is 16x faster per flop and leaves the vector ALU free for all but 8 of its 16 cycles (tools/valu_probe.hip).  So every
// float32 operand x is split into two halves, x = hi + lo with hi = f16(x), lo = f16(x - hi) (both round-to-nearest:
// |x - hi - lo| <= 2^-22 |x|, or 2^-25 absolute once lo is subnormal), and a product a b is formed as
// a_hi b_hi + a_hi b_lo + a_lo b_hi in float32 accumulators -- three f16 MFMAs, relative error <= ~3 2^-22 per
// product, below what the hardware sin / cos already leave in Phi.  To keep the operands in f16's range the constant
// factors stay out of the tiles: the B operands are cos(arg) and sin(arg) (x w) (|.| <~ a few hundred), the
// accumulators are scaled by c = sqrt(2 var / B) and c / ell^2 when they are stored.
//
// Workgroup: 512 threads = 8 waves, a tile of 64 MT samples x 144 columns of BOTH products, K in steps of 32; two
// workgroups per CU (<= 128 VGPRs, 66 KB of LDS at 14 joints) = four waves per SIMD, which is what the vector ALU needs
// to issue at its full rate (tools/valu_probe.hip: v_fma_f32 9.5 cycles per instruction per wave at 1 or 2 waves per
// SIMD, 11.8 at 4 -> 2.4x the throughput), and the two workgroups drift apart so one's MFMA phase runs beside the other's
// generation phase.  Per K step a thread draws MT Philox counters (4 normals of a W row each) and forms ~4.5 PAIRS of
// features (one frequency at two adjacent points: the projections x . omega of both by v_pk_fma_f32 with the frequency
// broadcast through op_sel, so a thread holds DM frequency components, not 2 DM; the f16 halves by v_cvt_pk_f16_f32);
// then each wave runs 54 MFMAs on its 16-row tile(s).
// LDS tiles are [row][32 k] f16 with 64-byte rows and no padding: the 16-byte chunk c of row r sits at chunk
// c ^ h[(r >> 2) & 3], h = (0, 2, 3, 1), which makes the ds_read_b128 fragment reads of a 16-row tile conflict-free
// for the lane groups the hardware serves together ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS).
#pragma once

namespace {

typedef _Float16 vg_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 vg_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 vg_h8 __attribute__((ext_vector_type(8)));
typedef float vg_f2 __attribute__((ext_vector_type(2)));

#ifndef VG_HS_WBAR
#define VG_HS_WBAR 1         // scheduling barrier between the Philox counters of a thread
#endif
#ifndef VG_HS_PROJ2
#define VG_HS_PROJ2 0        // projection as two interleaved partial chains
#endif
#ifndef VG_HS_PRIO
#define VG_HS_PRIO 0         // raised priority for the product phase
#endif
#ifndef VG_HS_SKIP
#define VG_HS_SKIP 0         // measurement: 1 = no products, 2 = no W draws, 4 = no features (results are garbage)
#endif
constexpr int kHK = 32;                  // K step = one v_mfma_f32_16x16x32_f16
constexpr int kHThreads = 512;
constexpr int kHRowBytes = 2 * kHK;      // 64-byte tile rows

__device__ __forceinline__ int vg_swz(int row, int chunk) { return chunk ^ ((0x1320 >> (((row >> 2) & 3) << 2)) & 3); }

// Philox-4x32-10 with one v_mad_u64_u32 per 32 x 32 -> 64 product (both halves from one instruction; the compiler emits
// v_mul_lo_u32 + v_mul_hi_u32 for the C form).  Same values as vg_philox.
__device__ __forceinline__ uint4 vg_philox_mad(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c.x), "s"(0xD2511F53u) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c.z), "s"(0xCD9E8D57u) : "vcc");
        c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
// vg_normal4 on it: the same expressions, hence the same four normals
__device__ __forceinline__ float4 vg_normal4_mad(uint32_t i, uint32_t stream, uint2 key) {
    uint4 r = vg_philox_mad(make_uint4(i, stream, 0u, 0u), key);
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.z)));
    const float u1 = vg_u01(r.y), u3 = vg_u01(r.w);
    return make_float4(r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1),
                       r1 * __builtin_amdgcn_cosf(u3), r1 * __builtin_amdgcn_sinf(u3));
}

__device__ __forceinline__ float vg_uniform(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// x = hi + lo, both halves rounded to nearest (v_cvt_pk_f16_f32)
__device__ __forceinline__ void vg_split2(vg_f2 x, vg_h2& hi, vg_h2& lo) {
    hi = __builtin_convertvector(x, vg_h2);
    const vg_f2 back = __builtin_convertvector(hi, vg_f2);
    lo = __builtin_convertvector(x - back, vg_h2);
}

// acc += x * (w[0], w[0])  and  acc += x * (w[1], w[1]): v_pk_fma_f32 reading one half of the pair `w` for both lanes
// (op_sel / op_sel_hi).  Written as (w, w) vectors the compiler builds each broadcast pair with a v_mov_b64 first.
__device__ __forceinline__ vg_f2 vg_pk_fma_lo(vg_f2 x, vg_f2 w, vg_f2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
}
__device__ __forceinline__ vg_f2 vg_pk_fma_hi(vg_f2 x, vg_f2 w, vg_f2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
}

inline size_t vg_fused_split_lds(int MT, int DM) {
    return (size_t)2 * kTS * MT * kHRowBytes + (size_t)4 * kTJ * kHRowBytes + (size_t)kTJ * DM * sizeof(float) +
           (size_t)2 * kHK * (DM + 4) * sizeof(float);
}

template <bool DELL, int DM, int MT>      // d/d ell wanted; joint-space extent padded to DM (8 or 16); 64 MT sample rows
__global__ __launch_bounds__(kHThreads, 4) void prior_fused_split_kernel(FusedBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_lds[];
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.z % L, p = blockIdx.z / L;
    constexpr int kRows = kTS * MT;
    const int s0 = blockIdx.y * kRows, j0 = blockIdx.x * kTJ;
    const size_t pl = (size_t)p * L + l;
    unsigned char* Ah = hs_lds;                                   // [kRows][64 B]   W, high halves
    unsigned char* Al = Ah + kRows * kHRowBytes;                  //                 W, low halves
    unsigned char* Bt = Al + kRows * kHRowBytes;                  // [4][144][64 B]  cos hi, cos lo, (sin x.w) hi, (sin x.w) lo
    float* pts = reinterpret_cast<float*>(Bt + 4 * kTJ * kHRowBytes);       // [72][DM][2]  the tile's points in pairs (x[2 q][d], x[2 q + 1][d]), zero padded
    constexpr int kOLd = DM + 4;                                  // a frequency's row: DM components, the phase / 2 pi, pad to 16 bytes
    float* oms = pts + kTJ * DM;                                  // [2][32][kOLd]  the K step's frequencies, double buffered
    for (int e = tid; e < kTJ * DM; e += kHThreads) {
        const int jj = e / DM, d = e - jj * DM, j = min(j0 + jj, J - 1);
        const double* pt = j < N ? a.X + (size_t)j * D : a.Zy + (size_t)p * a.zy_stride + (size_t)(j - N) * D;
        pts[((jj >> 1) * DM + d) * 2 + (jj & 1)] = d < D ? (float)pt[d] : 0.f;
    }
    for (int e = tid; e < 2 * kHK * kOLd; e += kHThreads) oms[e] = 0.f;
    const float ell = softplus_f((float)a.raw_ell[pl]);
    const float var = (float)kVarFloor + softplus_f((float)a.raw_var[pl]);
    // (wave-uniform values computed by vector instructions: moved to scalar registers, the vector ones are all spoken for)
    const float inv_ell = vg_uniform(1.0f / ell), c = vg_uniform(__builtin_amdgcn_sqrtf(2.0f * var / (float)B));
    const float kInv2Pi = 0.15915494309189535f;
    const float rs = vg_uniform(inv_ell * kInv2Pi);
    const vg_f2 rev_scale = (vg_f2){rs, rs};
    const uint2 key = vg_key(a.seed, a.problem_base + p, a.ctr ? *a.ctr : a.step);
    // ---- generation roles
    // W: MT = 2: thread (row = tid / 4, part = tid % 4) draws the 8 normals of k = 8 part .. 8 part + 7 (two counters);
    //    MT = 1: thread (row = tid / 8, part = tid % 8) the 4 normals of k = 4 part .. 4 part + 3 (one counter)
    constexpr int kWShift = MT == 2 ? 2 : 3;
    const int wrow = tid >> kWShift, wpart = tid & ((1 << kWShift) - 1);
    const uint32_t wbase = ((a.wOff + ((uint32_t)min(s0 + wrow, S - 1) * L + l) * (uint32_t)B) >> 2) + (uint32_t)wpart * MT;
    // features: thread (fk = tid % 32, jq = tid / 32) forms frequency fk for the point pairs jq + 16 i (points 2 q, 2 q + 1)
    const int fk = tid & 31, jq = tid >> 5;
    // frequencies: element e of the step's 32 contiguous rows of omega (32 D floats), then the 32 phases
    const int n_om = kHK * D, n_ob = n_om + kHK;
    constexpr int kONext = (kHK * 16 + kHK + kHThreads - 1) / kHThreads;      // loads per thread (2 up to 16 joints)
    float onext[kONext];
    auto om_fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < kONext; ++q) {
            const int e = tid + q * kHThreads;
            const float* src = e < n_om ? a.omega + (pl * B + k0) * D + e : a.beta + pl * B + k0 + min(e - n_om, kHK - 1);
            onext[q] = e < n_ob ? *src : 0.f;
        }
    };
    auto om_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < kONext; ++q) {
            const int e = tid + q * kHThreads;
            if (e < n_ob) {
                const int row = e < n_om ? e / D : e - n_om, d = e < n_om ? e - row * D : DM;
                oms[(buf * kHK + row) * kOLd + d] = e < n_om ? onext[q] : onext[q] * kInv2Pi;
            }
        }
    };
    // ---- product roles: MT = 2: wave w owns sample rows 16 w .. 16 w + 15 of both products;
    //                     MT = 1: row tile w % 4, product w / 4 (F0 for waves 0-3, H for waves 4-7)
    constexpr int NU = MT;                                        // (row tile, product) units per wave
    const int rt = MT == 2 ? wave : (wave & 3);
    const int mat0 = MT == 2 ? 0 : (wave >> 2);
    vg_f32x4 acc[NU][kTJ / 16];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) acc[u][t] = (vg_f32x4){0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, g = lane >> 4;
    __syncthreads();
    om_fetch(0);
    om_store(0);
    __syncthreads();
    int ob = 0;
    for (int k0 = 0; k0 < B; k0 += kHK) {
        // ================= generate the K step's operands (the next step's frequencies are requested first, stored last)
        const bool more = k0 + kHK < B;
        if (more) om_fetch(k0 + kHK);
        // W: counter m of the thread covers k = 4 kq .. 4 kq + 3, kq = MT wpart + m: 8 bytes of each half tile
#pragma unroll
        for (int m = 0; m < ((VG_HS_SKIP & 2) ? 0 : MT); ++m) {
            const int kq = wpart * MT + m;
            const float4 w4 = vg_normal4_mad(wbase + (uint32_t)(k0 >> 2) + (uint32_t)m, VG_STREAM_W, key);
            vg_h2 h0, l0, h1, l1;
            vg_split2((vg_f2){w4.x, w4.y}, h0, l0);
            vg_split2((vg_f2){w4.z, w4.w}, h1, l1);
            const int off = wrow * kHRowBytes + vg_swz(wrow, kq >> 1) * 16 + (kq & 1) * 8;
            *reinterpret_cast<vg_h4*>(Ah + off) = (vg_h4){h0[0], h0[1], h1[0], h1[1]};
            *reinterpret_cast<vg_h4*>(Al + off) = (vg_h4){l0[0], l0[1], l1[0], l1[1]};
#if VG_HS_WBAR
            __builtin_amdgcn_sched_barrier(0);      // one counter at a time: two in flight need more registers than there are
#endif
        }
        {   // features
            vg_f2 om[DM / 2];                                     // (omega[fk][d], omega[fk][d + 1])
            const float* orow = oms + (ob * kHK + fk) * kOLd;
#pragma unroll
            for (int d = 0; d < DM; d += 4) {
                const vg_f32x4 o4 = *reinterpret_cast<const vg_f32x4*>(orow + d);
                om[d / 2] = (vg_f2){o4[0], o4[1]}; om[d / 2 + 1] = (vg_f2){o4[2], o4[3]};
            }
            const vg_f2 bt = (vg_f2){orow[DM], orow[DM]};
            const int ni = (VG_HS_SKIP & 4) ? 0 : wave < 4 ? 5 : 4;                      // point pairs jq + 16 i < 72 (wave-uniform)
            for (int i = 0; i < ni; ++i) {
                const int q = jq + 16 * i;
                vg_f2 proj = (vg_f2){0.f, 0.f};
#if VG_HS_PROJ2
                vg_f2 projb = (vg_f2){0.f, 0.f};
#pragma unroll
                for (int d = 0; d < DM; d += 2) {                 // (zero padding: the products beyond D add exact zeros)
                    const vg_f32x4 p4 = *reinterpret_cast<const vg_f32x4*>(pts + (q * DM + d) * 2);
                    proj = vg_pk_fma_lo(__builtin_shufflevector(p4, p4, 0, 1), om[d / 2], proj);
                    projb = vg_pk_fma_hi(__builtin_shufflevector(p4, p4, 2, 3), om[d / 2], projb);
                }
                proj += projb;
#else
#pragma unroll
                for (int d = 0; d < DM; d += 2) {                 // (zero padding: the products beyond D add exact zeros)
                    const vg_f32x4 p4 = *reinterpret_cast<const vg_f32x4*>(pts + (q * DM + d) * 2);
                    proj = vg_pk_fma_lo(__builtin_shufflevector(p4, p4, 0, 1), om[d / 2], proj);
                    proj = vg_pk_fma_hi(__builtin_shufflevector(p4, p4, 2, 3), om[d / 2], proj);
                }
#endif
                const vg_f2 rv = __builtin_elementwise_fma(proj, rev_scale, bt);
                const float r0 = __builtin_amdgcn_fractf(rv[0]), r1 = __builtin_amdgcn_fractf(rv[1]);
                const vg_f2 cs = (vg_f2){__builtin_amdgcn_cosf(r0), __builtin_amdgcn_cosf(r1)};
                vg_h2 hi, lo;
                vg_split2(cs, hi, lo);
                // rows 2 q and 2 q + 1 share their swizzle (same group of four rows): one address, the second row 64 bytes on
                unsigned char* dst = Bt + (2 * q) * kHRowBytes + vg_swz(2 * q, fk >> 3) * 16 + (fk & 7) * 2;
                *reinterpret_cast<_Float16*>(dst) = hi[0];
                *reinterpret_cast<_Float16*>(dst + kHRowBytes) = hi[1];
                *reinterpret_cast<_Float16*>(dst + kTJ * kHRowBytes) = lo[0];
                *reinterpret_cast<_Float16*>(dst + kTJ * kHRowBytes + kHRowBytes) = lo[1];
                if (DELL) {
                    const vg_f2 sn = (vg_f2){__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1)};
                    vg_split2(sn * proj, hi, lo);
                    *reinterpret_cast<_Float16*>(dst + 2 * kTJ * kHRowBytes) = hi[0];
                    *reinterpret_cast<_Float16*>(dst + 2 * kTJ * kHRowBytes + kHRowBytes) = hi[1];
                    *reinterpret_cast<_Float16*>(dst + 3 * kTJ * kHRowBytes) = lo[0];
                    *reinterpret_cast<_Float16*>(dst + 3 * kTJ * kHRowBytes + kHRowBytes) = lo[1];
                }
            }
        }
        if (more) om_store(ob ^ 1);
        __syncthreads();
        // ================= products: per 16 x 16 tile  hi hi + hi lo + lo hi, float32 accumulators
#if VG_HS_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        {
            const int arow = 16 * rt + r;
            const int aoff = arow * kHRowBytes + vg_swz(arow, g) * 16;
            const vg_h8 ah = *reinterpret_cast<const vg_h8*>(Ah + aoff);
            const vg_h8 al = *reinterpret_cast<const vg_h8*>(Al + aoff);
#pragma unroll
            for (int t = 0; t < ((VG_HS_SKIP & 1) ? 0 : kTJ / 16); ++t) {
                const int brow = 16 * t + r;
                const int boff = brow * kHRowBytes + vg_swz(brow, g) * 16;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int mat = MT == 2 ? u : mat0;
                    if (!DELL && mat == 1) continue;
                    const vg_h8 bh = *reinterpret_cast<const vg_h8*>(Bt + (2 * mat) * kTJ * kHRowBytes + boff);
                    const vg_h8 bl = *reinterpret_cast<const vg_h8*>(Bt + (2 * mat + 1) * kTJ * kHRowBytes + boff);
                    acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[u][t], 0, 0, 0);
                    acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[u][t], 0, 0, 0);
                    acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[u][t], 0, 0, 0);
                }
            }
        }
#if VG_HS_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        ob ^= 1;
    }
    // ---- D layout: col = lane & 15, row = (lane >> 4) * 4 + reg; the constant factors left out of the tiles go in here
    const float scale_f = c, scale_h = c * inv_ell * inv_ell;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int mat = MT == 2 ? u : mat0;
        if (!DELL && mat == 1) continue;
        float* dst = mat == 0 ? a.F0 : a.H;
        const float sc = mat == 0 ? scale_f : scale_h;
#pragma unroll
        for (int t = 0; t < kTJ / 16; ++t) {
            const int jc = j0 + 16 * t + r;
            if (jc >= J) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + 16 * rt + g * 4 + q;
                if (s < S) vg_stream(dst + (((size_t)p * S + s) * L + l) * J + jc, acc[u][t][q] * sc);
            }
        }
    }
}

}  // namespace
