// Float64 covariance path: Kuu, factorisation, q_sqrt, KL, A = Kfu (Kuu + jI)^-1 and tangents.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

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
    // stage B, role 3 also forms U = m + C eps of every sample (ws.U) for the likelihood that assembles its own paths
    // (Mz = 32): the MFMA sequence of paths_fwd_split_body on the float32 C it has just built
    int form_u, S;
    int rows_tpw;            // row tiles (kRowTile time points) per workgroup of the rows role
    int rows_wave;           // batches (with ki_in_a): the rows of A on the workgroup of stage A that formed the inverse, one wave per 16 time points, in registers (cov_rows_tail); stage B has no rows role then
    int ki_in_a;             // batches: (Kuu + jI)^-1 = Lk^-T Lk^-1 once per latent, by stage A (ws.Kinv) -- which then forms the rows of A as well (rows_wave);
                             // otherwise every row-tile workgroup of stage B forms its own (cov_rows_body)
    const float* eps;        // [P,L,S,Mz]  (ws.epsT: the generator's second copy, a latent's rows contiguous)
    HyperArgs hy;
    vg_workspace ws;
};

// stores of stage B that the NEXT launch reads (never this one): past the caches -- dirty lines left in L2 lengthen the
// hand-over to that launch
#define VG_HO(p, v) vg_stream((p), (v))
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
        for (int i = 0; i < 32; ++i) a[i] = La[min(i, Mz - 1) * ld + min(c & 31, Mz - 1)];      // loads first (pinned) ...
#pragma unroll
        for (int i = 0; i < 32; ++i) vg_pin(a[i]);
#pragma unroll
        for (int i = 0; i < 32; ++i) {                                                            // ... selects afterwards
            const double id = (i == (c & 31)) ? 1.0 : 0.0;
            a[i] = c < 32 ? ((i < Mz && c < Mz) ? a[i] : id) : id;
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
        // registers -> the [Mz][2 Mz + 1] image the tail expects: left half U = D L~^T, right half L~^-1.  One
        // unconditional store per row: lanes without a column write the image's pad column (2 Mz, never read) --
        // per-lane conditions here compile to a divergent branch per row
        double* dst = Img + (c < 32 ? (c < Mz ? c : 2 * Mz) : (c - 32 < Mz ? Mz + (c - 32) : 2 * Mz));
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < Mz) dst[i * la] = a[i];
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

// The elimination in two 16-pivot panels (16 < Mz <= 32), still on ONE wave and without barriers, but with half the row
// updates: the one-wave form above spends its time issuing 32 * 31 / 2 row updates (two scalar broadcasts + one FMA
// each) on a single SIMD; here the update of the second block row by the first panel is ONE pair of 16 x 16 x 16
// products on the float64 matrix cores.
//   panel 1: rows 0..15 of [K11 | K12 | I] (48 lanes)  ->  U11 = D1 L~11^T,  W = L~11^-1 K12,  L~11^-1
//   cores:   S = K22 - (W^T D1^-1) W   and   X = -(W^T D1^-1) L~11^-1        (= -K21 K11^-1)
//   panel 2: rows 16..31 as [S | X | I]                ->  U22 = D2 L~22^T,  L~22^-1 X,  L~22^-1
// which are the blocks of U = D L~^T and of L~^-1 the tail expects.  (The accumulator of S starts at K22, so the
// subtraction runs in pivot order like the elimination's; results differ from the forms above in the last bits.)
template <typename Load>
__device__ __forceinline__ void eliminate_panel16(double (&a)[16], double (&rinv)[16], Load) {
    double piv = vg_bcast_f64(a[0], 0);
    double r = __builtin_amdgcn_rcp(piv);
    r = fma(fma(-piv, r, 1.0), r, r);
    r = fma(fma(-piv, r, 1.0), r, r);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double rn = 0.0;
        rinv[k] = r;
        const double w = -a[k] * r;
        if (k + 1 < 16) {
            a[k + 1] = fma(vg_bcast_f64(a[k + 1], k), w, a[k + 1]);
            const double pn = vg_bcast_f64(a[k + 1], k + 1);
            rn = __builtin_amdgcn_rcp(pn);
            rn = fma(fma(-pn, rn, 1.0), rn, rn);
            rn = fma(fma(-pn, rn, 1.0), rn, rn);
        }
#pragma unroll
        for (int i = k + 2; i < 16; ++i) a[i] = fma(vg_bcast_f64(a[i], k), w, a[i]);
        r = rn;
    }
}
// Lkg / Lig: the global copies of both factors, written straight from the assembly (the LDS images are not formed then).
// li_img: also leave Lk^-1 in LDS (the Li region, [32][ld], zeros outside the lower triangle) for a product that follows.
__device__ __forceinline__ void chol_inverse_panels(double* La, double* Li, double* Aug, double* rsd, int Mz, int ld,
                                                    int tid, int nt, double* Lkg, double* Lig, bool li_img = false) {
    const int la = 2 * Mz + 1;
    double* Img = Aug;
    if (tid < VG_WAVE) {
        const int c = tid, cc = c & 15;
        double* Wt = Li;                 // [16][16] W            (the Li image is written by the tail only)
        double* Ws = Li + 256;           // [16][16] D1^-1 W
        double* Lt = Li + 512;           // [16][16] L~11^-1
        double* Dm = Li + 768;           // [16][16] dummy tile (stores of lanes without a column)
        double* SX = Img + 16 * la;      // [16][32] S | X        (rows 16.. of the image: written after panel 2 has read this)
        double a[16], rinv[16];
        // ---- panel 1
        // all loads first, pinned (the compiler otherwise sinks each load into the branch of its select: 16 dependent LDS
        // round trips, measured 1.4 us), then the selects
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = La[i * ld + min(c, Mz - 1)];
#pragma unroll
        for (int i = 0; i < 16; ++i) vg_pin(a[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = c < 32 ? (c < Mz ? a[i] : 0.0) : (i == c - 32 ? 1.0 : 0.0);
        VG_T(blockIdx.x == 0, 104);
        eliminate_panel16(a, rinv, 0);
        VG_T(blockIdx.x == 0, 105);
        {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) mine = c == i ? a[i] : mine;
            if (c < 16) rsd[c] = mine;
        }
        // one unconditional store per row and destination (per-lane conditions compile to a divergent branch per row):
        // lanes without a column write the image's pad column (2 Mz, never read) / a dummy tile
        {
            double* dI = Img + (c < 32 ? (c < Mz ? c : 2 * Mz) : (c < 48 ? Mz + (c - 32) : 2 * Mz));
            double* dT = (c >= 16 && c < 32 ? Wt : c >= 32 && c < 48 ? Lt : Dm) + cc;
            double* dS = (c >= 16 && c < 32 ? Ws : Dm) + cc;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                dI[i * la] = a[i];
                dT[i * 16] = a[i];
                dS[i * 16] = a[i] * rinv[i];
            }
        }
        VG_T(blockIdx.x == 0, 106);
        // ---- second block row through the matrix cores: lane -> A[i = cc][k = g], B[k = g][j = cc], D[row = g + 4 q][col = cc]
        {
            const int g = c >> 4;
            vg_f64x4 accS, accX = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) accS[q] = La[min(16 + g + 4 * q, Mz - 1) * ld + min(16 + cc, Mz - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) { double t = accS[q]; vg_pin(t); accS[q] = t; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 + g + 4 * q, col = 16 + cc;
                accS[q] = (row < Mz && col < Mz) ? accS[q] : (row == col ? 1.0 : 0.0);
            }
            double av[4], bs[4], bx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = g + 4 * u;
                av[u] = -Ws[k * 16 + cc]; bs[u] = Wt[k * 16 + cc]; bx[u] = Lt[k * 16 + cc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                accS = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bs[u], accS, 0, 0, 0);
                accX = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bx[u], accX, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                SX[(g + 4 * q) * 32 + cc] = accS[q];
                SX[(g + 4 * q) * 32 + 16 + cc] = accX[q];
            }
        }
        VG_T(blockIdx.x == 0, 107);
        // ---- panel 2
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = SX[i * 32 + (c & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) vg_pin(a[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = c < 32 ? a[i] : (i == c - 32 ? 1.0 : 0.0);
        VG_T(blockIdx.x == 0, 108);
        eliminate_panel16(a, rinv, 0);
        VG_T(blockIdx.x == 0, 109);
        {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) mine = c == i ? a[i] : mine;
            if (c < 16 && 16 + c < Mz) rsd[16 + c] = mine;
        }
        {
            double* dI = Img + (c < 16 ? (16 + c < Mz ? 16 + c : 2 * Mz)
                                : c < 32 ? Mz + (c - 16)
                                : c < 48 ? (16 + (c - 32) < Mz ? Mz + 16 + (c - 32) : 2 * Mz) : 2 * Mz);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (16 + i < Mz) dI[(16 + i) * la] = a[i];
        }
    }
    VG_T(blockIdx.x == 0, 140);
    __syncthreads();
    VG_T(blockIdx.x == 0, 141);
    if (tid < Mz) rsd[tid] = rsqrt(rsd[tid]);
    __syncthreads();
    VG_T(blockIdx.x == 0, 1100);
    for (int e = tid; e < 32 * 32; e += nt) {
        const int r = e >> 5, j = e & 31;
        const bool in = j <= r && r < Mz;
        const double lk = Img[min(j, Mz - 1) * la + min(r, Mz - 1)] * rsd[min(j, Mz - 1)];
        const double li = Img[min(r, Mz - 1) * la + Mz + min(j, Mz - 1)] * rsd[min(r, Mz - 1)];
        if (r < Mz && j < Mz) {
            Lkg[r * Mz + j] = in ? lk : 0.0;
            Lig[r * Mz + j] = in ? li : 0.0;
        }
        if (li_img) Li[r * ld + j] = in ? li : 0.0;      // (the wave's scratch tiles in this region are dead behind the barriers above)
    }
    VG_T(blockIdx.x == 0, 1101);
}

// The rows of A = Kfu (Kuu + jI)^-1 and their tangents for ONE latent, on the workgroup that has just formed the inverse (stage A,
// batches: CovArgs.rows_wave): wave w takes the 16-point tiles w, w + 4, ... entirely in registers -- Kfu^T by the Matern formula,
// then three kinds of 64-bit MFMA product (cov_rows_body's arithmetic and order of summation: the same bits).  Until round 5 these
// tiles were workgroups of stage B: they need nothing of stage B, only the inverse -- which stood in this workgroup's LDS when
// it ended -- and as the last role of that launch their 10-13 us each (a memory round trip for the operands, then every wave of
// the CU in its exponentials, then every wave in its MFMAs: lock step) were the launch's tail: 18 of its 32 us at 385 latents
// (profiles/r05/ab_runs.txt).
//   KiS: the inverse as a zero-padded Mp x ld image in LDS;  KdS: room for dKuu/dell's beside it;  zs: the latent's Mz inducing inputs in LDS
__device__ __forceinline__ void cov_rows_tail(const CovArgs& a, const double* KiS, double* KdS, int ld, const double* zs, double ell, double var,
                                              int l, int p, int tid, int nt) {
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D;
    const int Mp = (Mz + 15) & ~15;
    const int wave = tid >> 6, lane = tid & 63, j = lane & 15, kk = lane >> 4;
    const size_t pl = (size_t)p * L + l;
    const int ntile = (N + 15) >> 4;
    // dKuu/dell back from memory (this workgroup wrote it while it built Kuu; the elimination needed the LDS since) as a zero-padded
    // Mp x ld image like the inverse's: requested here, landing under the exponentials of the first tile
    constexpr int kKdRegs = (32 * 33 + kCovThreads - 1) / kCovThreads;
    const double* Kdg = a.ws.Kd_ell + pl * Mz * Mz;
    const float ild = 1.0f / (float)ld;
    double kdr[kKdRegs];
    if (a.want_dell) {
#pragma unroll
        for (int k = 0; k < kKdRegs; ++k) {
            const int e = tid + k * nt, r = vg_div(e, ild), c = e - r * ld;
            kdr[k] = Kdg[min(r, Mz - 1) * Mz + min(c, Mz - 1)];
        }
    }
    double kf[8], dkf[8];
    // Kfu^T and dKfu^T / dell at (inducing point 4 s + kk, time point n): the arithmetic of cov_rows_body (and of stage A's Kuu: one
    // division per thread, none per element)
    const double inv_ell = 1.0 / ell, c3 = 5.0 / (3.0 * ell);
    auto kfu = [&](int n, double xn) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double k = 0.0, dk = 0.0;
            const double zm = zs[min(4 * s + kk, Mz - 1)];
            if (4 * s + kk < Mz && n < N) {
                double rr = fabs(xn - zm) * inv_ell;
                double ex = exp(-kSqrt5 * rr);
                k = var * (1.0 + kSqrt5 * rr + (5.0 / 3.0) * rr * rr) * ex;
                dk = var * ex * (rr * rr * c3) * (1.0 + kSqrt5 * rr);
            }
            kf[s] = k; dkf[s] = dk;
        }
    };
    VG_T(l == 0 && p == 0, 143);
    if (wave < ntile) kfu(16 * wave + j, a.X[(size_t)min(16 * wave + j, N - 1) * D + l]);
    VG_T(l == 0 && p == 0, 144);
    if (a.want_dell) {
#pragma unroll
        for (int k = 0; k < kKdRegs; ++k) {
            const int e = tid + k * nt, r = vg_div(e, ild), c = e - r * ld;
            if (e < Mp * ld) KdS[e] = r < Mz && c < Mz ? kdr[k] : 0.0;
        }
    }
    __syncthreads();      // the inverse (emitted by the product before this call) and dKuu/dell stand
    VG_T(l == 0 && p == 0, 145);
    if (wave >= ntile) return;
    // A operands of the three kinds of product: X[16 rt + j][4 s + kk] of the zero-padded images, read where they are used (held in
    // registers -- 64 of them -- this routine would not fit the 128 of the launches it ends)
    const double* kiP = KiS + j * ld + kk;
    const double* kdP = KdS + j * ld + kk;
    float4* A4 = reinterpret_cast<float4*>(a.ws.A4) + pl * N * Mz;
    float* AT = a.ws.AT + pl * N * Mz;
    // the time stamps of this wave's tiles now: a load behind the stores of a tile would wait for them (one counter)
    constexpr int kMaxPass = 4;      // (N <= 256: the condition of CovArgs.ki_in_a)
    double xns[kMaxPass];
#pragma unroll
    for (int ps = 0; ps < kMaxPass; ++ps) xns[ps] = a.X[(size_t)min(16 * (wave + ps * (nt >> 6)) + j, N - 1) * D + l];
    (void)xns;
#pragma unroll
    for (int ps = 0; ps < kMaxPass; ++ps) {
        const int wt = wave + ps * (nt >> 6);
        if (wt >= ntile) break;
        const int n = 16 * wt + j;
        VG_T(l == 0 && p == 0, ps == 0 ? 146 : 148);
        if (ps > 0) kfu(n, xns[ps]);
        const vg_f64x4 zero = {0.0, 0.0, 0.0, 0.0};
        vg_f64x4 at[2], y[2], aell[2], avar[2];
        double op[2][8];      // the A operands of ONE kind of product: requested together, the two row tiles' chains interleaved
        auto fetch = [&](const double* P) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s = 0; s < 8; ++s) op[rt][s] = P[16 * rt * ld + 4 * s];
        };
        // A^T = Ki Kfu^T: element q of at[rt] is A[n][m], m = 16 rt + 4 q + kk = 4 (4 rt + q) + kk -- the B operand of step 4 rt + q
        fetch(kiP);
        at[0] = zero; at[1] = zero;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) at[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rt][s], kf[s], at[rt], 0, 0, 0);
        asm volatile("" ::: "memory");      // (the operand reads of the next product stay behind this one's: registers)
        // y^T = dKfu^T - dKuu/dell A^T
        y[0] = zero; y[1] = zero;
        if (a.want_dell) {
            fetch(kdP);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) y[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rt][s], at[s >> 2][s & 3], y[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[rt][q] = dkf[4 * rt + q] - y[rt][q];
        asm volatile("" ::: "memory");
        // A_ell^T = Ki y^T,  A_var^T = jitter / var Ki A^T
        fetch(kiP);
        aell[0] = zero; aell[1] = zero; avar[0] = zero; avar[1] = zero;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (a.want_dell) aell[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rt][s], y[s >> 2][s & 3], aell[rt], 0, 0, 0);
                avar[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rt][s], at[s >> 2][s & 3], avar[rt], 0, 0, 0);
            }
        VG_T(l == 0 && p == 0, ps == 0 ? 147 : 149);
        if (n < N) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = 16 * rt + 4 * q + kk;
                    if (m < Mz) {
                        const float av0 = (float)at[rt][q];
                        vg_stream(A4 + (size_t)n * Mz + m, make_float4(av0, (float)aell[rt][q], (float)(a.jitter / var * avar[rt][q]), 0.f));
                        vg_stream(AT + (size_t)m * N + n, av0);
                    }
                }
        }
    }
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
    // the latent's two scalars, each a chain of float64 exp / log / sqrt / division: on wave 0, the lengthscale in its
    // lower half and the variance in its upper half (the same arithmetic on different data: one instruction stream)
    if (tid < VG_WAVE) {
        const bool isv = tid >= 32;
        double raw;
        if (a.prologue == 2) {
            // large batches (mid_stage1_kernel): the hyper-parameter update of the previous step, committed here -- no other role of
            // that launch reads the kernel hyper-parameters, and this workgroup is the only writer of its latent's
            const HyperArgs& h = a.hy;
            const bool own = h.ctr && h.do_adam;
            const double lr_own = own ? adam_step_size(h.lr, (double)*h.ctr) : 0.0;
            if (own && pl == 0 && tid == 0) h.lr_store[0] = lr_own;
            const HyperState o = hyper_update_wave(h, pl, own, lr_own);
            if (tid == 0) {
                h.g_ell[pl] = o.g_ell;
                if (h.do_adam) { h.p_ell[pl] = o.raw_ell; h.m_ell[pl] = o.m_ell; h.v_ell[pl] = o.v_ell; }
            }
            if (tid == 32) {
                h.g_var[pl] = o.g_var;
                if (h.do_adam) { h.p_var[pl] = o.raw_var; h.m_var[pl] = o.m_var; h.v_var[pl] = o.v_var; }
            }
            raw = isv ? o.raw_var : o.raw_ell;
        } else if (a.prologue) {
            const HyperState o = hyper_update_wave(a.hy, pl);
            double* nx = a.hy.next + 6 * pl;
            if (tid == 0) { a.hy.g_ell[pl] = o.g_ell; nx[0] = o.raw_ell; nx[2] = o.m_ell; nx[3] = o.v_ell; }
            if (tid == 32) { a.hy.g_var[pl] = o.g_var; nx[1] = o.raw_var; nx[4] = o.m_var; nx[5] = o.v_var; }
            raw = isv ? o.raw_var : o.raw_ell;
        } else {
            raw = isv ? a.raw_var[pl] : a.raw_ell[pl];
        }
        double sp, sg;
        softplus_sigmoid_d(raw, &sp, &sg);
        if (tid == 0) { scal[0] = sp; a.ws.sig_ell[pl] = sg; }
        if (tid == 32) { scal[1] = kVarFloor + sp; a.ws.sig_var[pl] = sg; }
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
    // ((Kuu + jI)^-1 = Lk^-T Lk^-1 is formed by its consumers, the row tiles of stage B, from Lk^-1: a product less on
    // this workgroup, which is the longest role of its launch)
    double* Lkg = a.ws.Lk64 + pl * Mz * Mz;
    double* Lig = a.ws.Li64 + pl * Mz * Mz;
    if (Mz > 16 && Mz <= 32 && a.elim_wave) {
        chol_inverse_panels(La, Li, Sc, rsd, Mz, ld, tid, nt, Lkg, Lig, a.ki_in_a != 0);
        VG_T(l == 0 && p == 0, 102);
        if (a.ki_in_a) {
            // (Kuu + jI)^-1 = Lk^-T Lk^-1 here, once per latent: the routine, operands and order of the row-tile workgroups of
            // stage B (which then stage the product instead of Lk^-1 and skip theirs) -- the same bits
            __syncthreads();
            double* Kig = a.ws.Kinv + pl * Mz * Mz;
            double* KiS = Sc;      // (the elimination's scratch is free) the inverse stays here for the rows of A
            matmul_f64(MatView{Li, 1, ld}, MatView{Li, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
                if (r < Mz && c < Mz) Kig[(size_t)r * Mz + c] = v;
                KiS[r * ld + c] = v;      // (zero beyond Mz: Lk^-1 is zero padded)
            });
            VG_T(l == 0 && p == 0, 142);
            if (a.rows_wave) cov_rows_tail(a, KiS, Sc + Mp * ld, ld, zs, ell, var, l, p, tid, nt);
        }
    } else {
        if (Mz <= 32 && a.elim_wave) chol_inverse_wave(La, Li, Sc, rsd, Mz, ld, tid, nt);
        else if (Mz <= 32 && nt == 256) chol_inverse_regs(La, Li, Sc, rsd, Mz, ld, tid, nt);
        else chol_inverse_block(La, Li, Sc, rsd, Mz, ld, tid, nt);
        VG_T(l == 0 && p == 0, 102);
        for (int e = tid; e < Mz * Mz; e += nt) {
            const int i = vg_div(e, iMz), j = e - i * Mz;
            Lkg[e] = La[i * ld + j];
            Lig[e] = Li[i * ld + j];
        }
    }
    VG_T(l == 0 && p == 0, 103);
}

__global__ __launch_bounds__(kCovThreads, 2) void cov_a_kernel(CovArgs a) {
    extern __shared__ double sm[];
    cov_a_body(a, sm, blockIdx.x, blockIdx.y);
}

// ---- stage B: heterogeneous launch, role = blockIdx.x ---------------------------------------------
//   0            KL and its gradient wrt q_mu / q_sqrt
//   1, 2         forward-mode tangent wrt lengthscale / variance:  dC = (Lk Phi(Lk^-1 dK Lk^-T)) pad(Q), dKL
//   3            q_sqrt = Lk pad(Q) + jitter, m, the float32 copy of Lk -- and, when the likelihood assembles the paths
//                itself, U = m + C eps of every sample.  (KL and q_sqrt were one role: the longest of the stage, 14 us
//                against the tangents' 13; apart they take 6 and 9.)
//   4 + t        row tile t of A = Kfu (Kuu + jI)^-1 and its tangents (cov_rows_body)
constexpr int kCovRoleC = 3, kCovFixedRoles = 4;
// Role of workgroup b of a role-major grid (b = group * roles + j), rotated by the group: workgroups go to the 8 XCDs round
// robin, so with a role count that shares a factor with 8 the plain b % roles pins every role to a fixed subset of the XCDs --
// the tangent roles, the longest, to two of them with 8 roles (config-5 share: 946 -> 985 us per step).
__device__ __forceinline__ int cov_role_rotated(int b, int roles) { return (b + b / roles) % roles; }
__device__ void cov_rows_body(const CovArgs& a, double* sm, int tile, int l, int p, int tid, int nt);

// The small chores of the launch stage B belongs to, on ONE workgroup per latent (the first row tile; with the rows formed by stage A
// -- CovArgs.rows_wave -- the q_sqrt role).
__device__ __forceinline__ void cov_b_chores(const CovArgs& a, int l, int p, int tid) {
    // The step counter ticks where no kernel that reads it runs alongside: noise drawn before this launch
    // saw the old value, the noise of the next step and the Adam count see the new one.
    if (a.tick && l == 0 && p == 0 && tid == 0) {
        const uint32_t t = *a.tick + 1u;          // = 1-based Adam count of this step's update
        *a.tick = t;
        a.lr_dev[0] = adam_step_size(a.lr, (double)t);
    }
    // one word per thread, one round trip (on thread 0 of role 0 these were two serial ones on the role that
    // ended the stage)
    const size_t pl = (size_t)p * a.L + l;
    if (a.commit && a.hy.do_adam && tid < 6) {      // staged hyper-parameters of the prologue -> their tensors
        const HyperArgs& h = a.hy;
        double* dst = tid == 0 ? h.p_ell : tid == 1 ? h.p_var : tid == 2 ? h.m_ell : tid == 3 ? h.v_ell : tid == 4 ? h.m_var : h.v_var;
        dst[pl] = h.next[6 * pl + tid];
    }
    if (a.keep_prev && tid >= 8 && tid < 11) {      // this step's var / slopes for the prologue of the next step
        const double* src = tid == 8 ? a.ws.var : tid == 9 ? a.ws.sig_ell : a.ws.sig_var;
        double* dst = tid == 8 ? a.ws.prev_var : tid == 9 ? a.ws.prev_sig_ell : a.ws.prev_sig_var;
        dst[pl] = src[pl];
    }
}

template <bool TANGENTS, bool ROWS = true>      // ROWS: the launch has row-tile workgroups (false: the rows of A were stage A's, CovArgs.rows_wave)
__device__ __forceinline__ void cov_b_body(const CovArgs& a, double* sm, int role, int l, int p, int tid, int nt, double* red) {
    if (role >= kCovFixedRoles) {
        if (ROWS) {
            if (role == kCovFixedRoles) cov_b_chores(a, l, p, tid);
            cov_rows_body(a, sm, role - kCovFixedRoles, l, p, tid, nt);
        }
        return;
    }
    const bool crole = role == kCovRoleC, form_u = crole && a.form_u != 0;
    if (!ROWS && crole) cov_b_chores(a, l, p, tid);
    if ((role == 1 || role == 2) && (!TANGENTS || (role == 1 && !a.want_dell))) return;
    VG_T(l == 0 && p == 0, 200 + 10 * (crole ? 5 : role));
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
    // role 3 with form_u (Mz == 32): float32 q_sqrt^T and m in the scratch of the tangent roles, every sample's eps behind
    // the float64 regions (the launch's dynamic LDS is sized for it)
    float* ctl = reinterpret_cast<float*>(X2);          // [Mz][Mz]
    float* ml = ctl + Mz * Mz;                          // [Mz]
    float* epl = reinterpret_cast<float*>(qm + Mp);     // [S][Mz]
    const double jit = a.jitter, var = a.ws.var[pl];
    const double y0 = a.y_u[((size_t)p * 2 + 0) * L + l], y1 = a.y_u[((size_t)p * 2 + 1) * L + l];
    const double* Kg = a.ws.Ks64 + pl * Mz * Mz;
    constexpr int kQRegs = (VGPMP_MAX_MZ - 2) * (VGPMP_MAX_MZ - 2) / kCovThreads + 1;
    double qreg[kQRegs];
    double qdiag = 1.0;               // (role 0) Q[tid][tid]
    double mine_qmu = 0.0;            // (role 3) q_mu of row tid
    static_assert(VGPMP_MAX_MZ <= kCovThreads, "one row of m per thread");
    {
        // every operand by DMA, all requests in flight together (zero padding written directly)
        const double* Qg = a.q_sqrt + pl * M * M;
        auto all = [](int, int) { return true; };
        const bool square = Mz == Mp;      // no zero padding needed: whole rows in 16-byte units
        if (square) {
            if (role != 0) vg_stage_f64_square(La, ld, a.ws.Lk64 + pl * Mz * Mz, Mz, tid, nt);
            if (!crole) vg_stage_f64_square(Li, ld, a.ws.Li64 + pl * Mz * Mz, Mz, tid, nt);
        } else if (!(Mz & 1)) {     // zero padded to Mp, still in 16-byte units
            if (role != 0) vg_stage_f64_even(La, Mp, ld, a.ws.Lk64 + pl * Mz * Mz, Mz, Mz, tid, nt);
            if (!crole) vg_stage_f64_even(Li, Mp, ld, a.ws.Li64 + pl * Mz * Mz, Mz, Mz, tid, nt);
        } else {
            if (role != 0) vg_stage_f64(La, Mp, ld, a.ws.Lk64 + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
            if (!crole) vg_stage_f64(Li, Mp, ld, a.ws.Li64 + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
        }
        if (crole && !(M & 1)) {
            // pad(Q) with Q at [2:, 2:] as ONE linear image of 16-byte units (ld even: a unit = two columns of a row): units
            // inside the block come from the rows of q_sqrt by DMA, the others are zeros written directly; the upper triangle
            // of the block is cleared behind the wait.
            const int upl = ld >> 1, total = Mp * upl, lane = tid & (VG_WAVE - 1);
            for (int c0 = (tid & ~(VG_WAVE - 1)); c0 < total; c0 += nt) {
                const int iu = c0 + lane;
                if (iu < total) {
                    const int row = iu / upl, r = row - 2, c = 2 * (iu - row * upl) - 2;
                    if (r >= 0 && r < M && c >= 0 && c < M)
                        __builtin_amdgcn_global_load_lds((vg_gmem*)(Qg + (size_t)r * M + c), (vg_lmem*)((char*)Qp + 16 * (size_t)c0), 16, 0, VG_DMA_AUX);
                    else
                        reinterpret_cast<double2*>(Qp)[iu] = make_double2(0.0, 0.0);
                }
            }
        } else if (crole) {
            vg_stage_f64(Qp, Mp, ld, Qg, M, M, 2, 2, tid, nt, [](int r, int c) { return c <= r; });
        } else {
            if (role != 0) {
                const double* Kdg = role == 1 ? a.ws.Kd_ell + pl * Mz * Mz : Kg;
                if (square) vg_stage_f64_square(Kd, ld, Kdg, Mz, tid, nt);
                else if (!(Mz & 1)) vg_stage_f64_even(Kd, Mp, ld, Kdg, Mz, Mz, tid, nt);
                else vg_stage_f64(Kd, Mp, ld, Kdg, Mz, Mz, 0, 0, tid, nt, all);
            }
            // this thread's share of Q in registers: the KL terms are element-wise; the tangents write theirs where Lk was
            // once Lk has been used
#pragma unroll
            for (int k = 0; k < kQRegs; ++k) qreg[k] = Qg[min(tid + k * nt, M * M - 1)];
            // (role 0: the diagonal once more, one entry per thread of the first wave -- its logarithm and reciprocal are then ONE pass of one
            //  wave, not a branch every wave takes in every pass of the element-wise loop: 3.1-3.7 us of the role's 8-10)
            if (role == 0) qdiag = Qg[(size_t)min(tid, M - 1) * (M + 1)];
        }
        if (!crole) {
            // k0 | k1: the first two columns of Kuu;  qm: q_mu at [2:]
            vg_stage_words(k0, 4 * Mp, tid, nt, [&](int w) -> const void* {
                const int d = w >> 1, col = d >= Mp, i = d - col * Mp;
                return i < Mz ? reinterpret_cast<const uint32_t*>(Kg + (size_t)i * Mz + col) + (w & 1) : nullptr;
            });
            vg_stage_words(qm, 2 * Mp, tid, nt, [&](int w) -> const void* {
                const int i = w >> 1;
                return (i >= 2 && i < Mz) ? reinterpret_cast<const uint32_t*>(a.q_mu + pl * M + (i - 2)) + (w & 1) : nullptr;
            });
        } else if (tid >= 2 && tid < Mz) mine_qmu = a.q_mu[pl * M + (tid - 2)];
        // the samples' eps for U (16-byte units)
        if (form_u) vg_stage_16(epl, a.eps + pl * a.S * Mz, a.S * Mz / 4, tid, nt);
    }
    vg_dma_wait();
    __syncthreads();
    if (crole) {
        // ---- role 3: m, the float32 Lk, q_sqrt = Lk pad(Q) + jitter I (first two diagonal entries), U
        if (tid < Mz) {
            const float mi = (float)(tid == 0 ? y0 : (tid == 1 ? y1 : mine_qmu));
            VG_HO(a.ws.m + pl * Mz + tid, mi);
            if (form_u) ml[tid] = mi;
        }
        if (!(M & 1)) {      // (16-byte staging of Q above: the strict upper triangle of the block is not Q's)
            for (int e = tid; e < M * M; e += nt) {
                const int r = vg_div(e, iM), c = e - r * M;
                if (c > r) Qp[(r + 2) * ld + (c + 2)] = 0.0;
            }
        }
        {                                    // float32 copy for the gradient assembly (written here, not in stage A:
            float* Lk32 = a.ws.Lk32 + pl * Mz * Mz;      // stage A of the next step may overlap that kernel)
            for (int e = tid; e < Mz * Mz; e += nt) {
                const int i = vg_div(e, iMz), j = e - i * Mz;
                VG_HO(Lk32 + e, (float)La[i * ld + j]);
            }
        }
        __syncthreads();
        VG_T(l == 0 && p == 0, 251);
        float* C32 = a.ws.C + pl * Mz * Mz;
        float* C32T = a.ws.CT + pl * Mz * Mz;
        matmul_f64(MatView{La, ld, 1}, MatView{Qp, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
            if (r < Mz && c < Mz) {
                const float cv = (float)(v + (r == c && r < 2 ? jit : 0.0));
                VG_HO(C32 + (size_t)r * Mz + c, cv);
                VG_HO(C32T + (size_t)c * Mz + r, cv);
                if (form_u) ctl[c * Mz + r] = cv;
            }
        });
        if (form_u) {
            // U = m + eps C^T of every sample: sixteen samples per pass (two chunks of paths_fwd_split_body at once: its rows do
            // not mix), two 16-column tiles, eight k-interleaved MFMAs each -- the same operands in the same order, so the
            // same bits; four waves share the (pass, tile) units
            __syncthreads();
            const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
            const int S = a.S, units = 2 * ((S + 15) >> 4);
            // a wave's units share their column tile (units wv, wv + 4, ...: same parity): its B fragments once, in registers;
            // the A fragments of a unit are requested together, then the eight MFMAs run back to back
            const int mi = 16 * (wv & 1) + i;
            float bq[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) bq[k8] = ctl[(4 * k8 + kk) * 32 + mi];
            const float m0 = ml[mi];
            for (int u = wv; u < units; u += (int)(nt >> 6)) {
                const int s0 = (u >> 1) << 4, sr = s0 + i;
                const float* ep = epl + min(sr, S - 1) * 32;
                float aq[8];
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) aq[k8] = ep[4 * k8 + kk];
                vg_f32x4_t acc = {m0, m0, m0, m0};
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sr < S ? aq[k8] : 0.f, bq[k8], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = s0 + 4 * kk + q;
                    if (s < S) VG_HO(a.ws.U + (((size_t)p * S + s) * L + l) * 32 + mi, acc[q]);
                }
            }
        }
        VG_T(l == 0 && p == 0, 252);
        return;
    }
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
        dl[i] = mi - (K0(i) * c0 + K1(i) * c1);
    }
    const double kd_scale = role == 2 ? 1.0 / var : 1.0;      // dK/dvar = K / var, applied to the products
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
    VG_T(l == 0 && p == 0 && role == 0, 1110);
    if (role == 0) {
        // ---- role 0: KL and its gradient wrt q_mu / q_sqrt (Q from this thread's registers: element-wise terms)
        double* gklQ = a.ws.gkl_Q + pl * M * M;
#pragma unroll
        for (int k = 0; k < kQRegs; ++k) {
            const int e = tid + k * nt;
            if (e < M * M) {
                const int r = vg_div(e, iM), c = e - r * M;
                double gq = 0.0;
                if (c <= r) {
                    const double q = qreg[k];
                    klacc += q * q;
                    gq = q;
                }
                if (c != r) VG_HO(gklQ + e, gq);      // (the diagonal: below)
            }
        }
        if (tid < M) {
            klacc -= log(qdiag * qdiag);
            VG_HO(gklQ + (size_t)tid * (M + 1), qdiag - 1.0 / qdiag);
        }
        VG_T(l == 0 && p == 0, 1111);
        const double kl = block_sum(klacc, red, tid, nt);
        VG_T(l == 0 && p == 0, 1112);
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
        if (r < Mz && c < Mz) VG_HO(CT + (size_t)c * Mz + r, (float)v);       // stored transposed
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
    acc = block_sum(acc, red, tid, nt);
    if (tid == 0) (role == 1 ? a.ws.gkl_ell : a.ws.gkl_var)[pl] = acc;
    VG_T(l == 0 && p == 0, 203 + 10 * role);
}

template <bool TANGENTS, bool ROWS = true>
__device__ __forceinline__ void cov_b_body(const CovArgs& a, double* sm, int role, int l, int p) {      // (the whole workgroup on one role)
    __shared__ double red[kCovThreads / VG_WAVE];
    cov_b_body<TANGENTS, ROWS>(a, sm, role, l, p, (int)threadIdx.x, (int)blockDim.x, red);
}

template <bool TANGENTS, bool ROWS>
__global__ __launch_bounds__(kCovThreads, 4) void cov_b_kernel(CovArgs a) {
    extern __shared__ double sm[];
    // Workgroups are dispatched in linear order (x fastest): ROLE-major here, the long roles first -- d/dvar, d/dell, KL, q_sqrt,
    // then the row tiles -- so that what is still running when the queue is empty is a 6 us rows workgroup, not a 16 us tangent
    // (role-minor order left every latent's tangents of the last round as the launch's tail).  Consecutive workgroups are
    // different latents of one role: the roles spread evenly over the XCDs whatever their count.
    const unsigned roles = gridDim.x, latents = gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + roles * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned ord = lin / latents, lat = lin - ord * latents;
    const int role = ord == 0 ? 2 : ord == 1 ? 1 : ord == 2 ? 0 : (int)ord;      // (3 = q_sqrt, 4.. = row tiles)
    // (measurement build: start / end of every 64th latent's workgroup of each role -- ids 1500 / 1400 + 8 order + latent / 64)
    VG_T((lat & 63) == 0 && lat < 512 && ord < 8, 1500 + 8 * ord + (lat >> 6));
    cov_b_body<TANGENTS, ROWS>(a, sm, role, (int)(lat % gridDim.y), (int)(lat / gridDim.y));
    VG_T((lat & 63) == 0 && lat < 512 && ord < 8, 1400 + 8 * ord + (lat >> 6));
}

// A = Kfu (Kuu + jI)^-1 and its tangents for a tile of kRowTile time points:
//   A_ell = (dKfu/dell - A dKuu/dell) Kinv,   A_var = (jitter / var) A Kinv
// Output float32: A4[n][m] = {A, A_ell, A_var, 0} (one 16-byte load per use in the reverse pass)
// and AT[m][n] for the forward path assembly.
__device__ void cov_rows_padded_body(const CovArgs& a, double* sm, int wg_tile, int l, int p, int tid, int nt);
__device__ void cov_rows_body(const CovArgs& a, double* sm, int wg_tile, int l, int p, int tid, int nt) {
    if (a.M + 2 != 32) { cov_rows_padded_body(a, sm, wg_tile, l, p, tid, nt); return; }
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D, ld = (Mz + 2) & ~1;
    // a workgroup walks `tpw` consecutive tiles (large batches: Lk^-1, dKuu/dell and (Kuu + jI)^-1 once for both of them -- one
    // tile per workgroup re-stages 16 KB and redoes the 32^3 product for every 8 time points, 13 times per latent at N = 100)
    const int tpw = max(a.rows_tpw, 1), tile = wg_tile * tpw;
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
    double* xs = zs + Mz;                  // [tpw][RT] times of this workgroup's tiles
    double* Lt = xs + tpw * kRowTile;      // [Mz][ld] Lk^-1 as it arrives
    const double ell = a.ws.ell[pl], var = a.ws.var[pl];
    const double inv_ell = 1.0 / ell, c3 = 5.0 / (3.0 * ell);      // (stage A's Kuu arithmetic: one division per thread, none per element)
    float4* A4 = reinterpret_cast<float4*>(a.ws.A4) + pl * N * Mz;
    float* AT = a.ws.AT + pl * N * Mz;
    const int n00 = tile * kRowTile;
    {
        auto all = [](int, int) { return true; };
        if ((Mz & 1) == 0) {
            vg_stage_f64_square(Lt, ld, a.ws.Li64 + pl * Mz * Mz, Mz, tid, nt);
            if (a.want_dell) vg_stage_f64_square(Kd, ld, a.ws.Kd_ell + pl * Mz * Mz, Mz, tid, nt);
            else for (int e = tid; e < Mz * ld; e += nt) Kd[e] = 0.0;
        } else {
            vg_stage_f64(Lt, Mz, ld, a.ws.Li64 + pl * Mz * Mz, Mz, Mz, 0, 0, tid, nt, all);
            vg_stage_f64(Kd, Mz, ld, a.ws.Kd_ell + pl * Mz * Mz, a.want_dell ? Mz : 0, Mz, 0, 0, tid, nt, all);
        }
        vg_stage_words(zs, 2 * (Mz + tpw * kRowTile), tid, nt, [&](int w) -> const void* {
            const int i = w >> 1;
            const double* src = i < Mz ? a.Zy + (size_t)p * a.zy_stride + (size_t)i * D + l
                                       : a.X + (size_t)min(n00 + i - Mz, N - 1) * D + l;
            return reinterpret_cast<const uint32_t*>(src) + (w & 1);
        });
    }
    vg_dma_wait();
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 232);
    // (Kuu + jI)^-1 = Lk^-T Lk^-1: on the float64 matrix cores when Mz is a multiple of 16 (no padding needed), else by
    // dot products over the non-zero part of the two columns; tile 0 keeps the copy the views / the inducing-location
    // reverse pass read
    {
        double* Kig = tile == 0 ? a.ws.Kinv + pl * Mz * Mz : nullptr;
        if ((Mz & 15) == 0) {
            matmul_f64(MatView{Lt, 1, ld}, MatView{Lt, ld, 1}, Mz, tid, nt, [&](int r, int c, double v) {
                Ki[r * ld + c] = v;
                if (Kig) Kig[(size_t)r * Mz + c] = v;
            });
        } else {
            for (int e = tid; e < Mz * Mz; e += nt) {
                const int r = vg_div(e, iMz), c = e - r * Mz, k0 = max(r, c);
                const double v = dot4(Lt + k0 * ld + r, ld, Lt + k0 * ld + c, ld, Mz - k0);
                Ki[r * ld + c] = v;
                if (Kig) Kig[e] = v;
            }
        }
    }
    if (Mz == 32) {
        // ---- Mz = 32: the four [rows x 32] . [32 x 32] products on the float64 matrix cores, sixteen time points per pass
        // (two tiles; one tile leaves half the rows empty).  The scalar form below reads both operands of every output from LDS
        // -- 256 eight-byte reads per thread and tile: at 64 x 14 latents x 13 tiles that was 6 GB through the chip's 79 TB/s
        // of LDS bandwidth, more than half of stage B's launch.  Lk^-1 is dead once (Kuu + jI)^-1 stands: its space takes the
        // kernel rows, the scalar form's four row blocks take A and y.
        double* kf2 = Lt;                      // [16][32]  Kfu rows
        double* df2 = Lt + 16 * 32;            // [16][32]  dKfu/dell rows
        double* ar2 = kf;                      // [16][32]  A rows
        double* yr2 = kf + 16 * 32;            // [16][32]  dKfu/dell - A dKuu/dell
        const int rows_w = tpw * kRowTile, wv = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
        __syncthreads();       // every wave has read Lk^-1
        for (int r0 = 0; r0 < rows_w; r0 += 16) {
            const int n0 = n00 + r0;
            if (n0 >= N) break;
            for (int e = tid; e < 16 * 32; e += nt) {
                const int r = e >> 5, m = e & 31, n = n0 + r;
                double k = 0.0, dk = 0.0;
                if (n < N && r0 + r < rows_w) {
                    double rr = fabs(xs[r0 + r] - zs[m]) * inv_ell;
                    double ex = exp(-kSqrt5 * rr);
                    k = var * (1.0 + kSqrt5 * rr + (5.0 / 3.0) * rr * rr) * ex;
                    dk = var * ex * (rr * rr * c3) * (1.0 + kSqrt5 * rr);
                }
                kf2[e] = k; df2[e] = dk;
            }
            __syncthreads();
            VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 233);
            if (wv < 2) {                      // A = Kfu (Kuu + jI)^-1, this wave's sixteen columns
                const vg_f64x4 acc = mfma_tile_f64(MatView{kf2, 32, 1}, MatView{Ki, ld, 1}, 32, lane, 0, 16 * wv);
#pragma unroll
                for (int q = 0; q < 4; ++q) ar2[(g + 4 * q) * 32 + 16 * wv + i] = acc[q];
            }
            __syncthreads();
            VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 234);
            float av[4] = {0.f, 0.f, 0.f, 0.f};
            const int j0 = 16 * (wv & 1);
            if (wv < 2) {                      // y = dKfu/dell - A dKuu/dell
                const vg_f64x4 acc = mfma_tile_f64(MatView{ar2, 32, 1}, MatView{Kd, ld, 1}, 32, lane, 0, j0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = (g + 4 * q) * 32 + j0 + i;
                    yr2[o] = df2[o] - acc[q];
                }
            } else {                           // A_var = (jitter / var) A (Kuu + jI)^-1: stays with the lane that stores it
                const vg_f64x4 acc = mfma_tile_f64(MatView{ar2, 32, 1}, MatView{Ki, ld, 1}, 32, lane, 0, j0);
#pragma unroll
                for (int q = 0; q < 4; ++q) av[q] = (float)(a.jitter / var * acc[q]);
            }
            __syncthreads();
            VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 235);
            if (wv >= 2) {                     // A_ell = y (Kuu + jI)^-1, and the three planes go out
                vg_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                if (a.want_dell) acc = mfma_tile_f64(MatView{yr2, 32, 1}, MatView{Ki, ld, 1}, 32, lane, 0, j0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = g + 4 * q, n = n0 + r, m = j0 + i;
                    if (n < N && r0 + r < rows_w) {
                        const float av0 = (float)ar2[r * 32 + m];
                        vg_stream(A4 + (size_t)n * 32 + m, make_float4(av0, (float)acc[q], av[q], 0.f));
                        vg_stream(AT + (size_t)m * N + n, av0);
                    }
                }
            }
            // (the next pass writes kf2 / df2 -- read before the last barrier -- and meets a barrier before it writes ar2)
        }
        VG_T(tile == 0 && l == 0 && p == 0, 231);
        return;
    }
    VG_T(tile == 0 && l == 0 && p == 0, 231);
}

// The rows role for any other Mz: the same four products on the float64 matrix cores with every operand zero padded to
// Mp = roundup(Mz, 16) columns (one to three column tiles).  The scalar form this replaces read both operands of every
// output from LDS -- at config 3 (Mz = 26: 385 latents x 9 tiles) it made stage B 52 us of a 183 us step.
__device__ void cov_rows_padded_body(const CovArgs& a, double* sm, int wg_tile, int l, int p, int tid, int nt) {
    const int M = a.M, Mz = M + 2, N = a.N, L = a.L, D = a.D;
    const int Mp = (Mz + 15) & ~15, ld = Mp + 2, ct = Mp >> 4;
    const int tpw = max(a.rows_tpw, 1), tile = wg_tile * tpw;
    VG_T(tile == 0 && l == 0 && p == 0, 230);
    const size_t pl = (size_t)p * L + l;
    double* Ki = sm;                       // [Mp][ld]  (Kuu + jI)^-1, zero padded
    double* Kd = Ki + Mp * ld;             // [Mp][ld]  dKuu/dell
    double* Lt = Kd + Mp * ld;             // [Mp][ld]  Lk^-1 as it arrives; from Mp = 32 its space takes the kernel rows afterwards
    double* ar2 = Lt + Mp * ld;            // [16][Mp]  A rows
    double* yr2 = ar2 + 16 * Mp;           // [16][Mp]  dKfu/dell - A dKuu/dell
    double* zs = yr2 + 16 * Mp;            // [Mp]
    double* xs = zs + Mp;                  // [tpw][RT]
    double* kf2 = Mp >= 32 ? Lt : xs + tpw * kRowTile;      // [16][Mp]  Kfu rows
    double* df2 = kf2 + 16 * Mp;                            // [16][Mp]  dKfu/dell rows
    const double ell = a.ws.ell[pl], var = a.ws.var[pl];
    const double inv_ell = 1.0 / ell, c3 = 5.0 / (3.0 * ell);      // (stage A's Kuu arithmetic: one division per thread, none per element)
    float4* A4 = reinterpret_cast<float4*>(a.ws.A4) + pl * N * Mz;
    float* AT = a.ws.AT + pl * N * Mz;
    const int n00 = tile * kRowTile;
    {
        auto all = [](int, int) { return true; };
        double* first = Lt;
        const double* first_g = a.ws.Li64 + pl * Mz * Mz;
        if (!(Mz & 1)) {
            vg_stage_f64_even(first, Mp, ld, first_g, Mz, Mz, tid, nt);
            vg_stage_f64_even(Kd, Mp, ld, a.ws.Kd_ell + pl * Mz * Mz, a.want_dell ? Mz : 0, Mz, tid, nt);
        } else {
            vg_stage_f64(first, Mp, ld, first_g, Mz, Mz, 0, 0, tid, nt, all);
            vg_stage_f64(Kd, Mp, ld, a.ws.Kd_ell + pl * Mz * Mz, a.want_dell ? Mz : 0, Mz, 0, 0, tid, nt, all);
        }
        vg_stage_words(zs, 2 * (Mp + tpw * kRowTile), tid, nt, [&](int w) -> const void* {
            const int i = w >> 1;
            if (i >= Mz && i < Mp) return nullptr;
            const double* src = i < Mz ? a.Zy + (size_t)p * a.zy_stride + (size_t)i * D + l
                                       : a.X + (size_t)min(n00 + i - Mp, N - 1) * D + l;
            return reinterpret_cast<const uint32_t*>(src) + (w & 1);
        });
    }
    vg_dma_wait();
    __syncthreads();
    VG_T(tile == 0 && l == 0 && p == 0, 232);
    {   // (Kuu + jI)^-1 = Lk^-T Lk^-1; tile 0 keeps the copy the views / the inducing-location reverse pass read
        double* Kig = tile == 0 ? a.ws.Kinv + pl * Mz * Mz : nullptr;
        matmul_f64(MatView{Lt, 1, ld}, MatView{Lt, ld, 1}, Mp, tid, nt, [&](int r, int c, double v) {
            Ki[r * ld + c] = v;
            if (Kig && r < Mz && c < Mz) Kig[(size_t)r * Mz + c] = v;
        });
    }
    const int rows_w = tpw * kRowTile, wv = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const float iMp = 1.0f / (float)Mp;
    __syncthreads();       // every wave has read Lk^-1; Ki stands
    for (int r0 = 0; r0 < rows_w; r0 += 16) {
        const int n0 = n00 + r0;
        if (n0 >= N) break;
        for (int e = tid; e < 16 * Mp; e += nt) {
            const int r = vg_div(e, iMp), m = e - r * Mp, n = n0 + r;
            double k = 0.0, dk = 0.0;
            if (m < Mz && n < N && r0 + r < rows_w) {
                double rr = fabs(xs[r0 + r] - zs[m]) * inv_ell;
                double ex = exp(-kSqrt5 * rr);
                k = var * (1.0 + kSqrt5 * rr + (5.0 / 3.0) * rr * rr) * ex;
                dk = var * ex * (rr * rr * c3) * (1.0 + kSqrt5 * rr);
            }
            kf2[e] = k; df2[e] = dk;
        }
        __syncthreads();
        VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 233);
        const int j0 = 16 * wv;
        if (wv < ct) {                     // A = Kfu (Kuu + jI)^-1, this wave's sixteen columns
            const vg_f64x4 acc = mfma_tile_f64(MatView{kf2, Mp, 1}, MatView{Ki, ld, 1}, Mp, lane, 0, j0);
#pragma unroll
            for (int q = 0; q < 4; ++q) ar2[(g + 4 * q) * Mp + j0 + i] = acc[q];
        }
        __syncthreads();
        VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 234);
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        if (wv < ct) {                     // y = dKfu/dell - A dKuu/dell;  A_var = (jitter / var) A (Kuu + jI)^-1 stays with its lane
            const vg_f64x4 accy = mfma_tile_f64(MatView{ar2, Mp, 1}, MatView{Kd, ld, 1}, Mp, lane, 0, j0);
            const vg_f64x4 accv = mfma_tile_f64(MatView{ar2, Mp, 1}, MatView{Ki, ld, 1}, Mp, lane, 0, j0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = (g + 4 * q) * Mp + j0 + i;
                yr2[o] = df2[o] - accy[q];
                av[q] = (float)(a.jitter / var * accv[q]);
            }
        }
        __syncthreads();
        VG_T(tile == 0 && l == 0 && p == 0 && r0 == 0, 235);
        if (wv < ct) {                     // A_ell = y (Kuu + jI)^-1, and the three planes go out
            vg_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            if (a.want_dell) acc = mfma_tile_f64(MatView{yr2, Mp, 1}, MatView{Ki, ld, 1}, Mp, lane, 0, j0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = g + 4 * q, n = n0 + r, m = j0 + i;
                if (m < Mz && n < N && r0 + r < rows_w) {
                    const float av0 = (float)ar2[r * Mp + m];
                    vg_stream(A4 + (size_t)n * Mz + m, make_float4(av0, (float)acc[q], av[q], 0.f));
                    vg_stream(AT + (size_t)m * N + n, av0);
                }
            }
        }
        __syncthreads();                   // (the next pass writes kf2 / df2 / ar2 / yr2)
    }
    VG_T(tile == 0 && l == 0 && p == 0, 231);
}

// The rows role of the batch schedules (16 < Mz <= 32, stage A has left (Kuu + jI)^-1 in ws.Kinv): ONE wave per 16 time points,
// everything in registers -- no LDS, no barriers; a workgroup is four independent waves (64 time points).  The products are formed
// TRANSPOSED, A^T = Ki Kfu^T etc. (Ki and dKuu/dell are symmetric to the bit): for v_mfma_f64_16x16x4_f64 the accumulator of one
// product -- D[row = (l >> 4) + 4 q][col = l & 15] -- IS the B operand -- B[k = l >> 4][j = l & 15], step q + 4 rt -- of the next,
// so the chain  A^T -> y^T = dKfu^T - dKuu A^T -> A_ell^T = Ki y^T,  A_var^T = j / var Ki A^T  needs no re-layout.  Lane (j, kk)
// evaluates the kernel at time point n0 + j against the 8 inducing points 4 s + kk: both the B operand of the first product and
// the accumulator layout of dKfu^T.  Same products in the same k order as cov_rows_body: the same bits.
// (The LDS form keeps four waves busy for ~5 us per 16 time points, most of it at its four barriers; this one wave for ~6.)

}  // namespace
