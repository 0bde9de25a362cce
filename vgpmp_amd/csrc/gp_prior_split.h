// The prior draws of a large batch on the f16 matrix pipe: prior_fused_split_kernel.
// Private part of gp_path.hip (one translation unit).
//
// Same job as prior_fused_batch_kernel (gp_prior.h): F0 = W Phi^T and H = W (dPhi/dell)^T with W from the Philox
// generator and the random features from sin / cos, all formed inside the GEMM's K loop (models/vgpmp.py:281-282: the
// `temporary_paths` draw of the reference, restated in oracle.elbo_forward / rff_features).  What changes is the
// arithmetic of the products.  On gfx950 v_mfma_f32_16x16x4_f32 runs at the FP32 vector rate and shares that datapath
// (generation and products serialise: profiles/r02/final/sq_prior_fused_config5.txt), while v_mfma_f32_16x16x32_f16
// is 16x faster per flop and holds the vector issue for only 8 of its 16 cycles (tools/valu_probe.hip).  So every
// float32 operand x is split into two halves, x = hi + lo with hi = f16(x), lo = f16(x - hi) (both round-to-nearest:
// |x - hi - lo| <= 2^-22 |x|, or 2^-25 absolute once lo is subnormal), and a product a b is formed as
// a_hi b_hi + a_hi b_lo + a_lo b_hi in float32 accumulators -- relative error <= ~3 2^-22 per product, below what the
// hardware sin / cos already leave in Phi (tests/test_gpu_surface.py::test_f16_split_prior_kernel_...: the distance from
// a float64 evaluation is not larger than the float32-MFMA kernels').  To keep the operands in f16's range the constant
// factors stay out of the tiles: the B operands are cos(arg) and sin(arg) (x . omega) (|.| <~ a few hundred), the
// accumulators are scaled by c = sqrt(2 var / B) and c / ell^2 when they are stored.
//
// Workgroup: 512 threads = 8 waves, a tile of 64 MT samples x 144 columns of BOTH products, K in steps of 32; two
// workgroups per CU (<= 128 VGPRs, 74 KB of LDS) = four waves per SIMD -- the vector ALU needs that many to issue at its
// full rate -- and the two workgroups drift apart, so one's product phase runs beside the other's generation phase.
// Per K step:
//   * W: a thread draws ONE Philox counter = eight weights of a W row (vg_w8: one v_mad_u64_u32 per 32 x 32 -> 64 product, one
//     v_bitop3_b32 per three-way xor, then eight reads of the 16 KB bin-mean table kept in LDS).  The weights of the W stream ARE
//     float16 (vgpmp_device.h, "The W stream"): W has no low half, a product is  w b_hi + w b_lo  -- TWO MFMAs -- and with 128-row
//     tiles the lane that draws a counter is the lane whose A fragment it is: W never touches LDS.  (Round 4 drew float32 normals
//     by Box-Muller, split them and carried three MFMAs per product: the vector and the matrix pipe exclude each other on this
//     part -- profiles/r05/prior_phases.txt --, so the kernel's time IS its instruction count; this form has a third fewer.)
//   * features: the projections x . omega of the tile's 144 points on the step's 32 frequencies are themselves f16-split
//     products: A = [omega_hi | omega_lo] (joint coordinates 0..15 twice along K = 32), B1 = [x_hi | x_hi], B2 = [x_lo | 0],
//     so TWO MFMAs give omega_hi x_hi + omega_lo x_hi + omega_hi x_lo for a 16 x 16 block.  A lane then holds four adjacent
//     frequencies of one point: phase, v_fract, v_cos / v_sin, the f16 halves (v_cvt_pk_f16_f32 + v_fma_mix_f32) and ONE
//     8-byte LDS store per half tile.  (Formed by v_pk_fma_f32 chains this phase took half the kernel.)
//   * products: each wave runs 36 MFMAs on its 16-row tile(s), the fragments of the next column tile requested before
//     the MFMAs of the current one.
// LDS tiles are [row][32 k] f16 with 64-byte rows and no padding: the 16-byte chunk c of row r sits at chunk
// c ^ h[(r >> 2) & 3], h = (0, 2, 3, 1), which makes the ds_read_b128 fragment reads of a 16-row tile conflict-free
// for the lane groups the hardware serves together ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS).
#pragma once

namespace {

constexpr int kHK = 32;                  // K step = one v_mfma_f32_16x16x32_f16
constexpr int kHThreads = 512;
constexpr int kHRowBytes = 2 * kHK;      // 64-byte tile rows
constexpr int kHD = 16;                  // joint coordinates per half of a projection operand row (dof <= 16)

__device__ __forceinline__ int vg_swz(int row, int chunk) { return chunk ^ ((0x1320 >> (((row >> 2) & 3) << 2)) & 3); }
// byte offset of element k (0..31) of a tile row
__device__ __forceinline__ int vg_tile_off(int row, int k) { return row * kHRowBytes + vg_swz(row, k >> 3) * 16 + (k & 7) * 2; }

__device__ __forceinline__ float vg_uniform(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

inline size_t vg_fused_split_lds(int MT) {
    // W tile (MT = 1 only: with MT = 2 the fragments stay in the registers of the wave that multiplies with them), the four feature
    // tiles, the two point tiles, the double-buffered frequency tile and phases, the table of the W stream
    return (MT == 2 ? 0 : (size_t)kTS * MT * kHRowBytes) + (size_t)4 * kTJ * kHRowBytes + (size_t)2 * kTJ * kHRowBytes +
           (size_t)2 * kHK * kHRowBytes + (size_t)2 * kHK * sizeof(float) + (size_t)kWTableSize * sizeof(unsigned short);
}

// The path assembly as this kernel's epilogue (FWD; large batches, Mz = 32): what a workgroup ends with in its accumulators -- F0 of
// sixteen samples per wave at every point of the latent -- is what paths_fwd_regs (gp_paths.h) reads back from memory to form
//     u = m + eps C^T,   r = u - f0(Z) - sqrt(jitter) eps',   f = f0(X) + r A^T.
// The same float32 MFMAs on the same operands in the same order, F0's tile as the initial accumulator: f and r bit for bit what that
// kernel writes.  C = Lk pad(Q) (+ jitter on its first two diagonal entries) and m are formed HERE from stage A's Lk and the variables
// (stage B's q_sqrt role, which forms them for the reverse pass, runs behind this kernel's tiles in the same launch): the same routine
// on the same operands, the same bits.
struct FusedFwdArgs {
    const double *Lk64, *q_sqrt, *q_mu, *y_u;      // [P,L,Mz,Mz], [P,L,M,M], [P,L,M], [P,2,L]
    double jitter;
    const float *AT, *epsT, *eps2T;               // [P,L,Mz,N], [P,L,S,Mz] x 2
    float sqrt_jitter;
    float *f, *R;                                  // [P,S,L,N], [P,S,L,Mz]
};
// (bx, by, bz: column tile, row tile and latent pair of this workgroup -- the kernel below, or a role of prior_split_cov_b_kernel, gp_path.hip)
template <bool DELL, int MT, bool FWD = false>      // d/d ell wanted; 64 MT sample rows per workgroup; the path assembly as epilogue
__device__ __forceinline__ void prior_fused_split_body(const FusedBatchArgs& a, unsigned char* hs_lds, int bx, int by, int bz,
                                                       const FusedFwdArgs* fw = nullptr) {
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = bz % L, p = bz / L;
    constexpr int kRows = kTS * MT;
    constexpr int kARows = MT == 2 ? 0 : kRows;
    const int s0 = by * kRows, j0 = bx * kTJ;
    const size_t pl = (size_t)p * L + l;
    unsigned char* Ah = hs_lds;                                   // [kRows][64 B]   W (float16 as drawn)   -- MT = 1 only
    unsigned char* Bt = Ah + kARows * kHRowBytes;                 // [4][144][64 B]  cos hi, cos lo, (sin x.w) hi, (sin x.w) lo
    unsigned char* Xa = Bt + 4 * kTJ * kHRowBytes;                // [144][64 B]     the tile's points: [x_hi | x_hi]
    unsigned char* Xb = Xa + kTJ * kHRowBytes;                    //                                    [x_lo | 0]
    unsigned char* Om = Xb + kTJ * kHRowBytes;                    // [2][32][64 B]   the K step's frequencies [omega_hi | omega_lo], double buffered
    float* btp = reinterpret_cast<float*>(Om + 2 * kHK * kHRowBytes);      // [2][32]  their phases / 2 pi
    unsigned short* wtab = reinterpret_cast<unsigned short*>(btp + 2 * kHK);   // [8192]    magnitudes of the W stream (gp_wtable.h)
    for (int e = tid; e < kWTableSize / 8; e += kHThreads)
        reinterpret_cast<uint4*>(wtab)[e] = reinterpret_cast<const uint4*>(kWTable)[e];
    for (int e = tid; e < kTJ * kHD; e += kHThreads) {
        const int jj = e >> 4, d = e & (kHD - 1), j = min(j0 + jj, J - 1);
        const double* pt = j < N ? a.X + (size_t)j * D : a.Zy + (size_t)p * a.zy_stride + (size_t)(j - N) * D;
        const float x = d < D ? (float)pt[min(d, D - 1)] : 0.f;
        vg_h2 hi, lo;
        vg_split2(x, 0.f, hi, lo);
        *reinterpret_cast<_Float16*>(Xa + vg_tile_off(jj, d)) = hi[0];
        *reinterpret_cast<_Float16*>(Xa + vg_tile_off(jj, kHD + d)) = hi[0];
        *reinterpret_cast<_Float16*>(Xb + vg_tile_off(jj, d)) = lo[0];
        *reinterpret_cast<_Float16*>(Xb + vg_tile_off(jj, kHD + d)) = (_Float16)0.f;
    }
    for (int e = tid; e < 2 * kHK * kHRowBytes / 4; e += kHThreads) reinterpret_cast<uint32_t*>(Om)[e] = 0u;      // coordinates beyond D stay zero
    const float ell = softplus_f((float)a.raw_ell[pl]);
    const float var = (float)kVarFloor + softplus_f((float)a.raw_var[pl]);
    // (wave-uniform values computed by vector instructions: moved to scalar registers, the vector ones are all spoken for)
    const float inv_ell = vg_uniform(1.0f / ell), c = vg_uniform(__builtin_amdgcn_sqrtf(2.0f * var / (float)B));
    const float kInv2Pi = 0.15915494309189535f;
    const float rs = vg_uniform(inv_ell * kInv2Pi);
    const uint2 key = vg_key(a.seed, a.problem_base + p, a.ctr ? *a.ctr : a.step);
    // ---- generation roles
    // W: ONE counter of the W stream = the eight float16 weights k = 8 part .. 8 part + 7 of a row (vg_w8: forty Philox instructions
    //    and eight table reads; the weights ARE float16, so W has no low half: two MFMAs per product, not three).
    //    MT = 2: wave w multiplies with rows 16 w .. 16 w + 15 and lane (r, g)'s A fragment IS the counter (row 16 w + r, part g):
    //            every lane draws its own fragment into registers -- no W tile, no LDS round trip;
    //    MT = 1: two waves share a row tile (F0 / H): thread (row = tid / 4, part = tid % 4) of the first four waves draws into the tile
    const int wrow = MT == 2 ? 16 * wave + (lane & 15) : (tid >> 2) & (kRows - 1), wpart = MT == 2 ? lane >> 4 : tid & 3;
    const bool draws_w = MT == 2 || wave < 4;
    const uint32_t wbase = ((a.wOff + ((uint32_t)min(s0 + wrow, S - 1) * L + l) * (uint32_t)B) >> 3) + (uint32_t)wpart;
    // frequencies: element e of the step's 32 contiguous rows of omega (32 D floats), then the 32 phases
    const int n_om = kHK * D, n_ob = n_om + kHK;
    constexpr int kONext = (kHK * kHD + kHK + kHThreads - 1) / kHThreads;      // loads per thread (2 up to 16 joints)
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
            if (e < n_om) {
                const int row = e / D, d = e - row * D;
                vg_h2 hi, lo;
                vg_split2(onext[q], 0.f, hi, lo);
                unsigned char* tile = Om + buf * kHK * kHRowBytes;
                *reinterpret_cast<_Float16*>(tile + vg_tile_off(row, d)) = hi[0];
                *reinterpret_cast<_Float16*>(tile + vg_tile_off(row, kHD + d)) = lo[0];
            } else if (e < n_ob) {
                btp[buf * kHK + (e - n_om)] = onext[q] * kInv2Pi;
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
    vg_h8 a_frag = {};                                            // MT = 2: this step's A fragment
    __syncthreads();
    om_fetch(0);
    om_store(0);
    if (kHK < B) { om_fetch(kHK); om_store(1); }
    __syncthreads();
    int ob = 0;
#ifdef VGPMP_BISECT
    // phase stamps of workgroup (0, 0, 0), K steps 8..15, every wave: id = 640 + 48 (k - 8) + 8 phase + wave (tools/prior_trace.py)
#define VG_PT(phase) do { const int ki_ = (k0 >> 5) - 8; if (lane == 0 && bx == 0 && by == 0 && bz == 0 && ki_ >= 0 && ki_ < 8) \
        vg_tr_buf[640 + 48 * ki_ + 8 * (phase) + wave] = wall_clock64(); } while (0)
#else
#define VG_PT(phase) do { } while (0)
#endif
    for (int k0 = 0; k0 < B; k0 += kHK) {
        VG_PT(0);
        // ================= generate the K step's operands.  The frequency tile is double buffered two steps deep: the tile of step
        // k + 1 is complete, step k + 2's values sit in two registers since the end of the previous generation phase (their
        // requests flew under its products) and go into the buffer that step k just finished with -- so nothing requested from
        // memory is live across the features, where registers are shortest
        if (k0 > 0 && k0 + kHK < B) om_store(ob ^ 1);
        VG_PT(1);
        // ---- features: unit u = (point tile t, frequency half h) -> a 16 x 16 block of projections by two MFMAs;
        //      lane (r, g) then holds point 16 t + r against the frequencies 16 h + 4 g .. + 3
        // (MT = 1: waves 4-7, which draw no W, take the first twelve units, three each; waves 0-3 the last six)
        constexpr int kUnits = 2 * (kTJ / 16);
        const int u_first = MT == 2 ? wave : (wave >= 4 ? wave - 4 : 12 + wave);
        const int u_last = MT == 2 ? kUnits : (wave >= 4 ? 12 : kUnits);
        const int u_step = MT == 2 ? kHThreads / 64 : 4;
        for (int u = u_first; u < u_last; u += u_step) {
            const int t = u >> 1, h = u & 1;
            const int frow = 16 * h + r, prow = 16 * t + r;
            const vg_h8 fa = *reinterpret_cast<const vg_h8*>(Om + (ob * kHK + frow) * kHRowBytes + vg_swz(frow, g) * 16);
            const int poff = prow * kHRowBytes + vg_swz(prow, g) * 16;
            const vg_h8 xb = *reinterpret_cast<const vg_h8*>(Xb + poff);
            const vg_h8 xa = *reinterpret_cast<const vg_h8*>(Xa + poff);
            const vg_f32x4 bt = *reinterpret_cast<const vg_f32x4*>(btp + ob * kHK + 16 * h + 4 * g);
            vg_f32x4 proj = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, xb, (vg_f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            proj = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, xa, proj, 0, 0, 0);
            vg_f32x4 cs, sp;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rev = __builtin_amdgcn_fractf(fmaf(proj[i], rs, bt[i]));
                cs[i] = __builtin_amdgcn_cosf(rev);
                if (DELL) sp[i] = __builtin_amdgcn_sinf(rev) * proj[i];
            }
            // four adjacent k of row `prow`: chunk 2 h + g / 2, its half g % 2
            const int off = prow * kHRowBytes + vg_swz(prow, 2 * h + (g >> 1)) * 16 + (g & 1) * 8;
            vg_h4 hi, lo;
            vg_split4(cs, hi, lo);
            *reinterpret_cast<vg_h4*>(Bt + off) = hi;
            *reinterpret_cast<vg_h4*>(Bt + kTJ * kHRowBytes + off) = lo;
            if (DELL) {
                vg_split4(sp, hi, lo);
                *reinterpret_cast<vg_h4*>(Bt + 2 * kTJ * kHRowBytes + off) = hi;
                *reinterpret_cast<vg_h4*>(Bt + 3 * kTJ * kHRowBytes + off) = lo;
            }
        }
        // ---- W: the thread's counter covers k = 8 wpart .. 8 wpart + 7: one 16-byte chunk of the row (drawn last: the fragment is
        //      not live across the features)
        if (draws_w) {      // (uniform per wave)
            const vg_h8 w8 = vg_w8(wbase + (uint32_t)(k0 >> 3), key, wtab);
            if (MT == 2) a_frag = w8;
            else *reinterpret_cast<vg_h8*>(Ah + wrow * kHRowBytes + vg_swz(wrow, wpart) * 16) = w8;
        }
        if (k0 + 2 * kHK < B) om_fetch(k0 + 2 * kHK);
        VG_PT(2);
        __syncthreads();
        VG_PT(3);
        // ================= products: per 16 x 16 tile  w hi + w lo (the weight is a float16: nothing else), float32 accumulators
        {
            vg_h8 ah;
            if (MT == 2) ah = a_frag;
            else {
                const int arow = 16 * rt + r;
                ah = *reinterpret_cast<const vg_h8*>(Ah + arow * kHRowBytes + vg_swz(arow, g) * 16);
            }
            const int boff0 = r * kHRowBytes + vg_swz(r, g) * 16;      // (16 t + r has the swizzle of r)
            constexpr int NT = kTJ / 16;
            // items (column tile t, product u) in sequence, the fragments of item i + 1 requested before the MFMAs of item i: a ring of
            // two 8-register slots (a slot per column tile holding both products' fragments, 32 registers, left the allocator
            // so little room at 128 that it moved every read next to its MFMA: one exposed LDS round trip per product)
            constexpr int NI = (MT == 2 && DELL) ? 2 * NT : NT;
            vg_h8 bh[2], bl[2];
            auto load_i = [&](int i, int slot) {
                const int t = (MT == 2 && DELL) ? i >> 1 : i, mat = MT == 2 ? (DELL ? i & 1 : 0) : mat0;
                bh[slot] = *reinterpret_cast<const vg_h8*>(Bt + (2 * mat) * kTJ * kHRowBytes + t * 16 * kHRowBytes + boff0);
                bl[slot] = *reinterpret_cast<const vg_h8*>(Bt + (2 * mat + 1) * kTJ * kHRowBytes + t * 16 * kHRowBytes + boff0);
            };
            if (DELL || MT == 2 || mat0 == 0) {      // (without d/d ell the H waves of the 64-row form have nothing to multiply)
                load_i(0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int t = (MT == 2 && DELL) ? i >> 1 : i, u = (MT == 2 && DELL) ? i & 1 : 0;
                    if (i + 1 < NI) load_i(i + 1, (i + 1) & 1);
                    // (the scheduler, short of registers, otherwise sinks every read next to the MFMA that uses it)
                    __builtin_amdgcn_sched_barrier(0);
                    acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[i & 1], acc[u][t], 0, 0, 0);
                    acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[i & 1], acc[u][t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        VG_PT(4);
        __syncthreads();
        VG_PT(5);
        ob ^= 1;
    }
#undef VG_PT
    // ---- D layout: col = lane & 15, row = (lane >> 4) * 4 + reg; the constant factors left out of the tiles go in here
    const float scale_f = c, scale_h = c * inv_ell * inv_ell;
    if constexpr (FWD) {
        // (the K loop ended behind a barrier: its LDS is free)
        constexpr int Mz = 32, ldc = 34;
        const int M = Mz - 2;
        double* LkS = reinterpret_cast<double*>(hs_lds);               // [32][ldc]  Lk
        double* QpS = LkS + Mz * ldc;                                  // [32][ldc]  pad(Q): Q's lower triangle at [2:, 2:]
        float* Cs = reinterpret_cast<float*>(QpS + Mz * ldc);          // [32][33]   C (float32, as stage B writes it)
        float* ms = Cs + Mz * 33;                                      // [32]       m
        float* ATs = ms + Mz;                                          // [32][N]    A^T
        // (behind A^T, per wave: [16][32] F0 at the inducing points and [16][32] r of its sixteen samples)
        const double* Lkg = fw->Lk64 + pl * Mz * Mz;
        const double* Qg = fw->q_sqrt + pl * M * M;
        for (int e = tid; e < Mz * Mz; e += kHThreads) {
            const int i = e >> 5, jx = e & 31;
            LkS[i * ldc + jx] = Lkg[e];
            const int qi = i - 2, qj = jx - 2;
            QpS[i * ldc + jx] = (qi >= 0 && qj >= 0 && qj <= qi) ? Qg[min(qi, M - 1) * M + min(qj, M - 1)] : 0.0;
        }
        if (tid < Mz) {
            const double y0 = fw->y_u[((size_t)p * 2 + 0) * L + l], y1 = fw->y_u[((size_t)p * 2 + 1) * L + l];
            const double qm = fw->q_mu[pl * M + min(max(tid - 2, 0), M - 1)];
            ms[tid] = (float)(tid == 0 ? y0 : (tid == 1 ? y1 : qm));
        }
        {
            const float* ATg = fw->AT + pl * N * Mz;
            for (int e = tid; e < Mz * N; e += kHThreads) ATs[e] = ATg[e];
        }
        __syncthreads();
        if (tid < kCovThreads) {      // four waves, a 16 x 16 tile of C each: stage B's product (cov_b_body role 3), the same bits
            const double jit = fw->jitter;
            matmul_f64(MatView{LkS, ldc, 1}, MatView{QpS, ldc, 1}, Mz, tid, kCovThreads, [&](int rr, int cc, double v) {
                Cs[rr * 33 + cc] = (float)(v + (rr == cc && rr < 2 ? jit : 0.0));
            });
        }
        __syncthreads();
    }
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
                const float v = acc[u][t][q] * sc;
                if (s < S) vg_stream(dst + (((size_t)p * S + s) * L + l) * J + jc, v);
                if (FWD && mat == 0) acc[u][t][q] = v;      // (the tile as paths_fwd_regs would read it back)
            }
        }
    }
    if constexpr (FWD) {
        if (MT == 2 || mat0 == 0) {      // the waves that hold F0: sixteen samples each
            constexpr int Mz = 32;
            float* Cs = reinterpret_cast<float*>(reinterpret_cast<double*>(hs_lds) + 2 * Mz * 34);
            float* ms = Cs + Mz * 33;
            float* ATs = ms + Mz;
            float* f0z = ATs + ((Mz * N + 3) & ~3) + wave * 2 * 16 * Mz;
            float* rs = f0z + 16 * Mz;
            const int i = r, kk = g;      // (the names of paths_fwd_regs)
            const int srow0 = s0 + 16 * rt;
            // F0 at the inducing points, rows 4 kk + q of this wave's sixteen samples: out of the column tiles that hold points N ..
#pragma unroll
            for (int t = 0; t < kTJ / 16; ++t) {
                const int zc = 16 * t + i - N;
                if (zc >= 0 && zc < Mz) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) f0z[(4 * kk + q) * Mz + zc] = acc[0][t][q];
                }
            }
            // u = m + eps C^T, r = u - f0(Z) - sqrt(jitter) eps'
            const float* er = fw->epsT + (pl * S + min(srow0 + i, S - 1)) * Mz;
            float av[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) av[k8] = er[4 * k8 + kk];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mi = 16 * h + i;
                float bC[8];
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) bC[k8] = Cs[mi * 33 + 4 * k8 + kk];
                const float m0 = ms[mi];
                vg_f32x4_t au = {m0, m0, m0, m0};
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) au = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k8], bC[k8], au, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = 4 * kk + q, s = srow0 + sl;
                    const float e2 = fw->eps2T[(pl * S + min(s, S - 1)) * Mz + mi];
                    const float rr = vg_path_r(au[q], f0z[sl * Mz + mi], fw->sqrt_jitter, e2);
                    rs[sl * Mz + mi] = rr;
                    if (s < S) vg_stream(fw->R + (((size_t)p * S + s) * L + l) * Mz + mi, rr);
                }
            }
            // f = f0(X) + r A^T
            float rv[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) rv[k8] = rs[i * Mz + 4 * k8 + kk];
#pragma unroll
            for (int t = 0; t < kTJ / 16; ++t) {
                if (16 * t < N) {
                    const int n = min(16 * t + i, N - 1);
                    vg_f32x4_t af = {acc[0][t][0], acc[0][t][1], acc[0][t][2], acc[0][t][3]};
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) af = __builtin_amdgcn_mfma_f32_16x16x4f32(rv[k8], ATs[(4 * k8 + kk) * N + n], af, 0, 0, 0);
                    if (16 * t + i < N) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int s = srow0 + 4 * kk + q;
                            if (s < S) vg_stream(fw->f + (((size_t)p * S + s) * L + l) * N + n, af[q]);
                        }
                    }
                }
            }
        }
    }
}

template <bool DELL, int MT>
__global__ __launch_bounds__(kHThreads, 4) void prior_fused_split_kernel(FusedBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_lds[];
    prior_fused_split_body<DELL, MT>(a, hs_lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Few samples (S <= 32), more than 64 latents: prior_fused_small_kernel's job (gp_prior.h) with the products on the f16 matrix
// pipe.  That kernel is three quarters float32 MFMA -- 16-row tiles that 7 samples fill to 44 % -- on the pipe its vector work
// shares (lesson 42); here a K step of 32 bases costs twelve v_mfma_f32_16x16x32_f16 (hi lo + lo hi + hi hi for F0 and H of two
// column tiles: 8 of their 16 cycles hold the vector issue) instead of sixty-four float32 MFMAs of 32 cycles.  Everything stays in
// registers -- no LDS tiles, no barriers in the loop:
//   * a lane's B fragment is 8 CONSECUTIVE bases of one point: two float32 projection MFMAs per tile with the frequencies' rows
//     permuted (row 4 g + q of set s = base 8 g + 4 s + q) leave exactly those in lane (point, g);
//   * features: phase by one fused multiply-add, v_cos / v_sin, split into f16 halves (vg_split4) -- the constant factors c and
//     c / ell^2 go to the accumulators at the end, so the operands stay in f16's range as in prior_fused_split_kernel;
//   * a lane's A fragment is 8 consecutive weights of one sample row: two 16-byte loads of W, split.
// Four waves = the four K-slices whose slabs the path kernels sum; operands requested one K step ahead.
// WX: the weights in a.W were drawn by the library's own generator -- float16 values (vgpmp_device.h, "The W stream") -- so W has no
// low half: no split, two MFMAs per product.  Weights handed in by the caller (parity tests) are arbitrary float32: WX = false.
// (bx, by: latent pair and column-tile pair of this workgroup -- the kernel below, or a role of mid_cov_b_prior16_kernel, gp_path.hip)
template <int MT, int DM, bool DELL, bool WX>    // 16-row sample tiles; joint extent DM = D for 6 / 7 joints, else padded to 8 or 16; d/d ell wanted
__device__ __forceinline__ void prior_small16_body(const FusedPriorArgs& a, int bx, int by) {
    __shared__ float pts[kFNT * 16][DM];
    const int S = a.S, L = a.L, J = a.J, N = a.N, D = a.D, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, sk = tid >> 6;      // 4 waves = 4 K-slices
    const int pl = bx, l = pl % L, p = pl / L, j0 = by * (kFNT * 16);
    if (a.tick && bx == 0 && by == 0 && tid == 0) *a.tick += 1u;
    // (measurement build: phases of workgroup (0, 0) -- id 1300 + phase; starts / ends of a sample of workgroups; tools/step_trace.py)
    VG_T(bx == 0 && by == 0, 1300);
    VG_T((bx & 63) == 0 && bx < 512 && by < 4, 1320 + 4 * (bx >> 6) + by);
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
    // projections (float32 MFMAs, as in prior_fused_small_kernel): B operands = this lane's point coordinates, loop invariant
    constexpr int kDQ = (DM + 3) / 4;
    float pb[kFNT][kDQ];
#pragma unroll
    for (int t = 0; t < kFNT; ++t)
#pragma unroll
        for (int i = 0; i < kDQ; ++i) pb[t][i] = 4 * i + g < DM ? pts[16 * t + r][min(4 * i + g, DM - 1)] : 0.f;
    int ooff[kDQ];
#pragma unroll
    for (int i = 0; i < kDQ; ++i) ooff[i] = min(4 * i + g, D - 1);
    // A operand row r of projection set s holds base 8 (r >> 2) + 4 s + (r & 3) of the K step: accumulator element q of lane
    // (point, g) is then base 8 g + 4 s + q
    const float* orow = a.omega + ((size_t)pl * B + kbeg + 8 * (r >> 2) + (r & 3)) * D;
    const float* brow = a.beta + (size_t)pl * B + kbeg + 8 * g;                 // phases of this lane's 8 bases
    const float* wrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wrow[m] = a.W + (((size_t)p * S + min(16 * m + r, S - 1)) * L + l) * B + kbeg + 8 * g;
    struct Ops { float o[2][kDQ]; vg_f32x4 b0, b1, w0[MT], w1[MT]; };
    auto fetch = [&](int k, Ops& x) {
        const int kk = min(k, kchunk - kHK);             // (the look-ahead of the last step re-reads it)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < kDQ; ++i) x.o[s2][i] = orow[(size_t)(kk + 4 * s2) * D + ooff[i]];
        x.b0 = *reinterpret_cast<const vg_f32x4*>(brow + kk);
        x.b1 = *reinterpret_cast<const vg_f32x4*>(brow + kk + 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            x.w0[m] = *reinterpret_cast<const vg_f32x4*>(wrow[m] + kk);
            x.w1[m] = *reinterpret_cast<const vg_f32x4*>(wrow[m] + kk + 4);
        }
    };
    auto step = [&](const Ops& x) {
        // A fragments: 8 weights of the lane's sample row, split
        vg_h8 ah[MT], al[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (WX) {
                const vg_h4 h0 = __builtin_convertvector(x.w0[m], vg_h4), h1 = __builtin_convertvector(x.w1[m], vg_h4);      // exact
                ah[m] = (vg_h8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                al[m] = ah[m];                                    // (unused)
            } else {
                vg_h4 h0, l0, h1, l1;
                vg_split4(x.w0[m], h0, l0);
                vg_split4(x.w1[m], h1, l1);
                ah[m] = (vg_h8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                al[m] = (vg_h8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
        }
        const vg_f32x4 bt0 = vg_scale4(x.b0, 0.15915494309189535f), bt1 = vg_scale4(x.b1, 0.15915494309189535f);      // phases in revolutions
#pragma unroll
        for (int t = 0; t < kFNT; ++t) {
            vg_f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
            for (int i = 0; i < kDQ; ++i) {
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.o[0][i], pb[t][i], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.o[1][i], pb[t][i], p1, 0, 0, 0);
            }
            vg_f32x4 c0, c1, s0, s1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float r0 = __builtin_amdgcn_fractf(fmaf(p0[q], rev_ell, bt0[q]));
                const float r1 = __builtin_amdgcn_fractf(fmaf(p1[q], rev_ell, bt1[q]));
                c0[q] = __builtin_amdgcn_cosf(r0); c1[q] = __builtin_amdgcn_cosf(r1);
                if (DELL) { s0[q] = __builtin_amdgcn_sinf(r0) * p0[q]; s1[q] = __builtin_amdgcn_sinf(r1) * p1[q]; }
            }
            vg_h4 h0, l0, h1, l1;
            vg_split4(c0, h0, l0);
            vg_split4(c1, h1, l1);
            const vg_h8 bh = (vg_h8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            const vg_h8 bl = (vg_h8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (!WX) accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, accF[m][t], 0, 0, 0);
                accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, accF[m][t], 0, 0, 0);
                accF[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, accF[m][t], 0, 0, 0);
            }
            if (DELL) {
                vg_split4(s0, h0, l0);
                vg_split4(s1, h1, l1);
                const vg_h8 dh = (vg_h8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const vg_h8 dl = (vg_h8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (!WX) accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], dh, accH[m][t], 0, 0, 0);
                    accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], dl, accH[m][t], 0, 0, 0);
                    accH[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], dh, accH[m][t], 0, 0, 0);
                }
            }
        }
    };
    Ops xa, xb;
    VG_T(bx == 0 && by == 0, 1301);
    fetch(0, xa);
    int k = 0;
    for (; k + 2 * kHK <= kchunk; k += 2 * kHK) {
        fetch(k + kHK, xb);
        step(xa);
        fetch(k + 2 * kHK, xa);
        step(xb);
    }
    if (k < kchunk) step(xa);      // (an odd number of K steps)
    VG_T(bx == 0 && by == 0, 1302);
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg; the constant factors left out of the operands go in here
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
    VG_T(bx == 0 && by == 0, 1303);
    VG_T((bx & 63) == 0 && bx < 512 && by < 4, 1360 + 4 * (bx >> 6) + by);
}
template <int MT, int DM, bool DELL, bool WX>
__global__ __launch_bounds__(kBlock, 2) void prior_fused_small16_kernel(FusedPriorArgs a) {
    prior_small16_body<MT, DM, DELL, WX>(a, blockIdx.x, blockIdx.y);
}

}  // namespace
