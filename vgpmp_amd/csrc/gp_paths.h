// Matheron path assembly and its reverse pass.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

// =================================================================================================
// Path assembly  (decoupled / Matheron update, vgpmp.py:281-282):
//   u = m + C eps;  r = u - F0(Z) - sqrt(jitter) eps2;  f = F0(X) + A r
// One workgroup per (chunk of VG_SC samples, latent, problem).
// =================================================================================================

struct PathArgs {
    int S, N, Mz, L, SK, NC;
    size_t slab, part_len;
    float sqrt_jitter;
    const float4* A4;
    const float *AT, *C, *CT, *CT_ell, *CT_var, *m, *F0, *H, *eps, *eps2;
    const float *epsT, *eps2T;    // [P,L,S,Mz] copies of eps / eps' (paths_fwd_regs, paths_bwd_regs); nullptr: the [P,S,Mz,L] tensors
    int nsplit;               // 2: paths_fwd on two workgroups per (chunk, latent), halves of the time axis
    float *R, *f;
    const float* G;
    float* part;
    int want_dell;
    int stop;
    // Workgroups go to the 8 XCDs round robin.  xcd_span > 0 (= workgroups / 8): XCD x takes the CONTIGUOUS range
    // [x span, (x + 1) span) of the (chunk, latent, problem) order, so the 2 * NC workgroups that stage the same
    // latent's A / C tangents sit behind one L2 (at most two latents per XCD instead of all of them)
    int xcd_span;
    uint32_t* tick;           // paths_fwd_sc8: the step counter ticks here (large-batch schedule; nullptr: elsewhere)
    int cpw;                  // paths_bwd_sc8: sample chunks per workgroup (the latent's A / C tangents are staged once for all of them)
    int NCp;                  // sets of partial sums per latent the reverse pass leaves: NC (one per chunk), or one per workgroup (paths_bwd_regs)
};
__device__ __forceinline__ int xcd_contiguous(int id, int span) { return span > 0 ? (id & 7) * span + (id >> 3) : id; }

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
                    const float r = vg_path_r(acc[q], f0s[sl * J + N + mi], a.sqrt_jitter, e2s[e]);
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
            const float r = vg_path_r(u, f0s[sl * J + N + mi], a.sqrt_jitter, e2s[e]);
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
                    const float r = vg_path_r(acc[q], f0z[e], a.sqrt_jitter, e2s[e]);
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
            const float r = vg_path_r(u, f0z[e], a.sqrt_jitter, e2s[e]);
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
__global__ __launch_bounds__(kBlock, 2) void paths_bwd_sc8(PathArgs a) {
    constexpr int SC = 8;
    extern __shared__ float smf[];
    __shared__ float red[3][kBlock / VG_WAVE];
    const int l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = a.Mz, L = a.L, J = N + Mz, NM = N * Mz;
    const float iMz = 1.0f / (float)Mz, iN = 1.0f / (float)N, iJ = 1.0f / (float)J;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };      // 16-byte aligned regions
    float* Ap = take(3 * NM);                        // [3][N][Mz] A, dA/dell, dA/dvar as planes
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
    const bool dell = a.want_dell != 0;
    // ---- what every chunk of this (problem, latent) shares: staged ONCE per workgroup (a.cpw chunks each).  With one
    //      chunk per workgroup the 16 chunks of a latent each pulled these 59 KB again: 689 MB per launch at the config-5
    //      share against ~190 MB of operands.  The float4 records {A, A_ell, A_var, -} go through registers into
    //      three planes: a quarter less LDS, and the MFMA operand reads below fall on consecutive banks
    {
        const float4* Ag = a.A4 + pl * NM;
        for (int e0 = 0; e0 < NM; e0 += 4 * nt) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = Ag[min(e0 + q * nt + tid, NM - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * nt + tid;
                if (e < NM) { Ap[e] = v[q].x; Ap[NM + e] = v[q].y; Ap[2 * NM + e] = v[q].z; }
            }
        }
        const float* Ce = a.CT_ell + pl * Mz * Mz;
        const float* Cv = a.CT_var + pl * Mz * Mz;
        vg_stage_rows(Ces, 2 * Mz, Mz, tid, nt, [&](int r) -> const float* {
            return r < Mz ? (dell ? Ce + (size_t)r * Mz : nullptr) : Cv + (size_t)(r - Mz) * Mz;
        });
    }
    const int ch_end = min((int)(blockIdx.x + 1) * a.cpw, a.NC);
    for (int ch = blockIdx.x * a.cpw; ch < ch_end; ++ch) {
    const int s_base = ch * SC;
    VG_T(ch == 0 && l == 0 && p == 0, 500);
    {
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
            const float* ap = Ap + wv * NM;                          // plane wv at [n][mi]
            for (int n = 0; n < N; n += 4) {
                const float a0 = i < SC ? gp[n + kk] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, ap[(n + kk) * 32 + 16 * h + i], acc[h], 0, 0, 0);
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
            const float gv = g[n];
            d = fmaf(gv, Ap[n * Mz + mi], d);
            de = fmaf(gv, Ap[NM + n * Mz + mi], de);
            dv = fmaf(gv, Ap[2 * NM + n * Mz + mi], dv);
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
    __syncthreads();      // the chunk's operands are overwritten by the next one's
    }
}

// The reverse pass of large batches (Mz = 32, N a multiple of 4 up to 4 KS, one slab of prior draws): what a latent's
// chunks share -- the three planes of A4 and the two tangents of C, B operands of every product here -- lives in
// REGISTERS (a lane's fragments of all K steps: 2 KS values for the waves that form G A, 32 for the wave that forms
// eps C^T), read from global memory once per workgroup; LDS then holds only the operands of the current PAIR of 8-sample
// chunks (40 KB: four workgroups per CU instead of two), the 16 x 16 MFMA tiles use all sixteen rows (two chunks at once;
// paths_bwd_sc8 leaves eight of them empty) and no product waits on an LDS read of a B operand.  Per chunk the
// element-wise part, the partial sums and their order are those of paths_bwd_sc8 -- the same numbers per CHUNK; since round 5
// the chunks of a workgroup are then added in float32 here (one set of sums per workgroup), where paths_bwd_sc8 leaves a set per
// chunk for the assembly's float64 sum: agreement to the rounding of that float32 sum, 2e-6 relative, not bit for bit
// (tests/test_gpu_surface.py::test_reverse_path_pass_...).  Measured on paths_bwd_sc8 at the config-5 share (257 us): the
// MFMA loops 82 us, per-chunk staging latency 73 us, staging the constants through LDS and the loop skeleton 46 us.
// VG_PBR_DIRECT (the product form since round 5): the prior draws F0 / H of a chunk's samples are NOT staged -- every element is used
// once, by the thread that owns it, so it comes straight from memory into a register (10 per thread and chunk); their rows were 62 %
// of a pair's LDS-DMA requests, and issuing those requests was 40 % of the kernel (profiles/r05/ab_runs.txt: 4.3 of 10.4 us per
// pair).  131 -> 113 us at the config-5 share, bit-identical.  0 keeps the staged form (measurement).
#ifndef VG_PBR_DIRECT
#define VG_PBR_DIRECT 1
#endif
#ifndef VG_PBR_BUFS
#define VG_PBR_BUFS 1
#endif
constexpr int kPbrBufs = VG_PBR_BUFS;   // sets of staged rows (2 = the next pair requested under the current pair's work; with F0 / H staged that was
                                 // 67 KB of LDS, two workgroups per CU instead of three -- 198 against 131 us: the resident workgroups ARE the overlap)
constexpr int kPbrWaves = 3;     // 168 registers: at 4 (128) the register-resident fragments spill (160 vs 132 us at the config-5 share)
template <int KS>
__global__ __launch_bounds__(kBlock, kPbrWaves) void paths_bwd_regs(PathArgs a) {
    constexpr int SC = 8, Mz = 32, R2 = 2 * SC;
    extern __shared__ float smf[];
    __shared__ float red[3][kBlock / VG_WAVE];
    // (the step counter's tick when the path assembly was the prior kernel's epilogue: nothing in this launch reads the counter)
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *a.tick += 1u;
    // (re-reading the workgroup ids so that the two workgroups of a latent -- each fetches its A4 planes and the tangents of C, 59 KB --
    //  share an XCD and its L2 was measured in round 6: 725.6 against 726.1 us per step at the config-5 share, nothing; not kept)
    const int bx = (int)blockIdx.x, l = (int)blockIdx.y, p = (int)blockIdx.z;
    const int tid = threadIdx.x, nt = kBlock;
#ifdef VGPMP_BISECT
    // (measurement build: start / end of the workgroups of latent 0 of every 16th problem -- ids 1900 / 1930 + 2 (p / 16) + x; tools/step_trace.py)
    struct WgStamp {
        int id;
        __device__ WgStamp(int base, int l, int p, int x) : id(-1) { if (l == 0 && (p & 15) == 0 && p < 128 && x < 2) { id = base + 2 * (p >> 4) + x; VG_T(true, id); } }
        __device__ ~WgStamp() { if (id >= 0) VG_T(true, id + 30); }
    } wg_stamp_(1900, l, p, bx);
#endif
    const int S = a.S, N = a.N, L = a.L, J = N + Mz;
    const float iMz = 1.0f / (float)Mz, iN = 1.0f / (float)N;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };      // 16-byte aligned regions
    // the rows a pair of chunks stages: kPbrBufs sets of them (two: the next pair is requested under the current pair's work)
    float *GsB[kPbrBufs], *f0B[kPbrBufs], *RsB[kPbrBufs], *EsB[kPbrBufs];
#pragma unroll
    for (int b = 0; b < kPbrBufs; ++b) {
        GsB[b] = take(R2 * N);                       // [16][N]      the pair's rows: chunk c at rows 8 c ..
        f0B[b] = take(VG_PBR_DIRECT ? 0 : 2 * R2 * J);      // [16][J] prior draws, then [16][J] their d/dell (staged form only)
        RsB[b] = take(R2 * Mz);                      // [16][Mz]
        EsB[b] = take(R2 * Mz);                      // [16][Mz]
    }
    float* dRs2 = take(R2 * Mz);                     // [16][Mz]
    float* dGA = take(5 * R2 * Mz);                  // [5][16][Mz] G A, G A_ell, G A_var, eps C_var^T, eps C_ell^T
    float* accL = take(Mz * Mz + Mz);                // the workgroup's running sums of dC and dm
    const bool dell = a.want_dell != 0;
    const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    // ---- B fragments of every K step, once per workgroup
    static_assert(2 * KS >= 32, "the wave that forms eps C^T keeps 32 fragments in the same registers");
    float bA[2 * KS];                                // waves 0-2: plane wv of A4 at rows 4 ks + kk, columns 16 h + i
    float* bCv = bA;                                 // wave 3 (same registers): (dC/dvar)^T and (dC/dell)^T at rows 4 k8 + kk
    float* bCe = bA + 16;
    if (wv < 3) {
        const float* Ag = reinterpret_cast<const float*>(a.A4 + pl * N * Mz) + wv;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int n = 4 * ks + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float v = Ag[(size_t)(min(n, N - 1) * Mz + 16 * h + i) * 4];
                bA[2 * ks + h] = n < N ? v : 0.f;
            }
        }
    } else {
        const float* Cv = a.CT_var + pl * Mz * Mz;
        const float* Ce = a.CT_ell + pl * Mz * Mz;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = (4 * k8 + kk) * Mz + 16 * h + i;
                bCv[2 * k8 + h] = Cv[o];
                const float e = Ce[dell ? o : 0];
                bCe[2 * k8 + h] = dell ? e : 0.f;
            }
    }
    const int cp_end = min((int)(bx + 1) * a.cpw, a.NC);
    // (measurement build: phases of workgroup (0, 0, 0) -- id 1200 + 10 pair + phase; tools/pbr_trace.py)
#define VG_PBT(pair, phase) VG_T(bx == 0 && l == 0 && p == 0, 1200 + 10 * (pair) + (phase))
    VG_PBT(0, 9);
    auto stage_pair = [&](int ch0, int b) {
        const int s_base = ch0 * SC;
        float *Gs2 = GsB[b], *Rs2 = RsB[b], *Es2 = EsB[b];
        [[maybe_unused]] float* f0s2 = f0B[b];      // (staged only without VG_PBR_DIRECT)
        {
            vg_stage_rows(Gs2, R2, N, tid, nt, [&](int r) -> const float* {
                const int s = s_base + r;
                return s < S ? a.G + (((size_t)p * S + s) * L + l) * N : nullptr;              // zero beyond S
            });
            vg_stage_rows(Rs2, R2, Mz, tid, nt, [&](int r) -> const float* {
                const int s = s_base + r;
                return s < S ? a.R + (((size_t)p * S + s) * L + l) * Mz : nullptr;
            });
            if (a.epsT)
                vg_stage_rows(Es2, R2, Mz, tid, nt, [&](int r) -> const float* {
                    const int s = s_base + r;
                    return s < S ? a.epsT + (pl * S + s) * Mz : nullptr;
                });
            else
                vg_stage_words(Es2, R2 * Mz, tid, nt, [&](int w) -> const void* {
                    const int sl = vg_div(w, iMz), mi = w - sl * Mz, s = s_base + sl;
                    return s < S ? a.eps + (((size_t)p * S + s) * Mz + mi) * L + l : nullptr;
                });
#if !VG_PBR_DIRECT
            vg_stage_rows(f0s2, 2 * R2, J, tid, nt, [&](int r) -> const float* {
                const int second = r >= R2, s = min(s_base + (second ? r - R2 : r), S - 1);
                if (second && !dell) return nullptr;
                return (second ? a.H : a.F0) + (((size_t)p * S + s) * L + l) * J;
            });
#endif
        }
    };
    // The partial sums of ALL this workgroup's chunks leave it as ONE set (round 5; a set per chunk before: 16 x 4.2 KB per latent at
    // 128 samples -- 61 MB per launch at the config-5 share, written here and read again by the gradient assembly, which they made the
    // longest role of the next step's first launch): per chunk the sums are those of paths_bwd_sc8, the chunks' sums are added in
    // chunk order in float32 registers; the assembly adds the a.NCp sets of a latent in float64 as before.
    // (the running sums of dm / dC in LDS -- a thread only ever touches its own elements: no barrier -- the 168 registers are spoken for)
    constexpr int kCE = Mz * Mz / kBlock;
    float acc_se = 0.f, acc_sv = 0.f, acc_sr = 0.f;
#pragma unroll
    for (int k = 0; k < kCE; ++k) accL[tid + k * kBlock] = 0.f;
    if (tid < Mz) accL[Mz * Mz + tid] = 0.f;
    if (kPbrBufs > 1) stage_pair(bx * a.cpw, 0);
    int cb = 0;
    for (int ch0 = bx * a.cpw; ch0 < cp_end; ch0 += 2) {
        [[maybe_unused]] const int pair_i = (ch0 - bx * a.cpw) >> 1;      // (measurement build's phase stamps)
        VG_PBT(pair_i, 0);
        if (kPbrBufs == 1) stage_pair(ch0, 0);
        VG_PBT(pair_i, 1);
        vg_dma_wait();
        __syncthreads();
        VG_PBT(pair_i, 2);
        float *Gs2 = GsB[cb], *f0s2 = f0B[cb], *Rs2 = RsB[cb], *Es2 = EsB[cb];
        float* hs2 = f0s2 + R2 * J;
        // the next pair's rows into the other set while this pair is worked on (everybody has left that set: the barrier above)
        if (kPbrBufs > 1) { if (ch0 + 2 < cp_end) stage_pair(ch0 + 2, cb ^ 1); cb ^= 1; }
        // ---- the five products on all sixteen rows
        if (wv < 3) {
            float av[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) av[ks] = Gs2[i * N + min(4 * ks + kk, N - 1)];
            vg_f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)          // (K steps beyond N multiply by the zeros of bA: exact no-ops, no branches)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bA[2 * ks + h], acc[h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) dGA[(wv * R2 + 4 * kk + q) * Mz + 16 * h + i] = acc[h][q];
        } else {
            float ev[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) ev[k8] = Es2[i * Mz + 4 * k8 + kk];
            vg_f32x4_t accv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acce[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    accv[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[k8], bCv[2 * k8 + h], accv[h], 0, 0, 0);
                    acce[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[k8], bCe[2 * k8 + h], acce[h], 0, 0, 0);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dGA[(3 * R2 + 4 * kk + q) * Mz + 16 * h + i] = accv[h][q];
                    dGA[(4 * R2 + 4 * kk + q) * Mz + 16 * h + i] = acce[h][q];
                }
        }
        __syncthreads();
        VG_PBT(pair_i, 3);
        // ---- per chunk: exactly the element-wise part and the partials of paths_bwd_sc8
        for (int c = 0; c < 2 && ch0 + c < a.NC; ++c) {
            const int ch = ch0 + c;
            const float* Gs = Gs2 + c * SC * N;
            [[maybe_unused]] const float* f0s = f0s2 + c * SC * J;
            [[maybe_unused]] const float* hs = hs2 + c * SC * J;
            const float* Rs = Rs2 + c * SC * Mz;
            const float* Es = Es2 + c * SC * Mz;
            float* dRs = dRs2 + c * SC * Mz;
            float se = 0.f, sv = 0.f, sr = 0.f;
#if VG_PBR_DIRECT
            // the prior draws of the chunk's samples straight from memory into registers (each element is used once, by the thread that
            // owns it: staging their rows through LDS was 62 % of a pair's DMA requests); requested here, used below
            constexpr int kXE = (SC * 128 + kBlock - 1) / kBlock;      // elements per thread of the X part (N <= 128)
            float fz = 0.f, hz = 0.f, fx[kXE], hx[kXE];
            {
                const int s_c = ch * SC;
                auto row = [&](const float* base, int sl) { return base + (((size_t)p * S + min(s_c + sl, S - 1)) * L + l) * J; };
                if (tid < SC * Mz) {
                    const int sl = vg_div(tid, iMz), mi = tid - sl * Mz;
                    fz = row(a.F0, sl)[N + mi];
                    hz = dell ? row(a.H, sl)[N + mi] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < kXE; ++k) {
                    const int e = min(tid + k * kBlock, SC * N - 1), sl = vg_div(e, iN), n = e - sl * N;
                    fx[k] = row(a.F0, sl)[n];
                    hx[k] = dell ? row(a.H, sl)[n] : 0.f;
                }
            }
#endif
            for (int e = tid; e < SC * Mz; e += nt) {
                [[maybe_unused]] const int sl = vg_div(e, iMz), mi = e - sl * Mz;
                const int o = c * SC * Mz + e;
                const float d = dGA[o], de = dGA[R2 * Mz + o], dv = dGA[2 * R2 * Mz + o];
                dRs[e] = d;
                const float uv = dGA[3 * R2 * Mz + o], ue = dGA[4 * R2 * Mz + o];
                const float rv = Rs[e];
                sv += rv * dv + d * uv;
#if VG_PBR_DIRECT
                se += rv * de + d * ue - d * hz;
                sr -= d * fz;
#else
                se += rv * de + d * ue - d * hs[sl * J + N + mi];
                sr -= d * f0s[sl * J + N + mi];
#endif
            }
#if VG_PBR_DIRECT
#pragma unroll
            for (int k = 0; k < kXE; ++k) {
                const int e = tid + k * kBlock;
                if (e < SC * N) {
                    const float gv = Gs[e];
                    sr = fmaf(gv, fx[k], sr);
                    se = fmaf(gv, hx[k], se);
                }
            }
#else
            for (int e = tid; e < SC * N; e += nt) {
                const int sl = vg_div(e, iN), n = e - sl * N;
                const float gv = Gs[e];             // zero for samples beyond S
                sr = fmaf(gv, f0s[sl * J + n], sr);
                se = fmaf(gv, hs[sl * J + n], se);
            }
#endif
            __syncthreads();
            if (tid < Mz) {
                float t = 0.f;
                for (int sl = 0; sl < SC; ++sl) t += dRs[sl * Mz + tid];
                accL[Mz * Mz + tid] += t;
            }
            // (one element per thread and pass: a (mi, four adjacent k) form with float4 reads of eps measured SLOWER, 145 vs 130 us
            //  at the config-5 share)
#pragma unroll
            for (int k = 0; k < kCE; ++k) {
                const int e = tid + k * kBlock, mi = e >> 5, kc = e & 31;
                float t = 0.f;
                for (int sl = 0; sl < SC; ++sl) t = fmaf(dRs[sl * Mz + mi], Es[sl * Mz + kc], t);
                accL[e] += t;
            }
            acc_se += se; acc_sv += sv; acc_sr += sr;
            __syncthreads();      // the chunk's rows are reused
            VG_PBT(pair_i, 4 + c);
        }
    }
    {
        float* out = a.part + (pl * a.NCp + bx) * a.part_len;
        if (tid < Mz) vg_stream(out + tid, accL[Mz * Mz + tid]);
        float* oC = out + Mz;
#pragma unroll
        for (int k = 0; k < kCE; ++k) vg_stream(oC + tid + k * kBlock, accL[tid + k * kBlock]);
        const float se = vg_wave_sum(acc_se), sv = vg_wave_sum(acc_sv), sr = vg_wave_sum(acc_sr);
        if ((tid & 63) == 0) { red[0][tid >> 6] = se; red[1][tid >> 6] = sv; red[2][tid >> 6] = sr; }
        __syncthreads();
        if (tid == 0) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
            for (int k = 0; k < (int)(nt >> 6); ++k) { t0 += red[0][k]; t1 += red[1][k]; t2 += red[2][k]; }
            float* os = oC + (size_t)Mz * Mz;
            os[0] = t0; os[1] = t1; os[2] = t2; os[3] = 0.f;
            os[4] = 0.f; os[5] = 0.f; os[6] = 0.f; os[7] = 0.f;      // second set: paths_bwd_split only
        }
    }
#undef VG_PBT
}

// The same reverse pass on TWO workgroups per (sample chunk, latent) for launches that leave half the chip idle:
// a workgroup's time here is the ~100 KB it stages at the ~25 KB/us one CU can pull, and everything downstream is
// linear in G, so the work splits by COLUMNS of the inducing axis: half h owns columns [h Mz/2, (h+1) Mz/2) of
// A / dR / dm / dC (disjoint outputs, no extra partials) and the time points [n0, n0 + nx) of the two prior-draw dot
// products (a second set of the three scalars, added by hyper_update).  Two threads per (sample, column) halve the
// N-long chains.  Needs Mz % 8 == 0, N % 4 == 0 (16-byte rows), SK > 1.
// (body: `wg_lin` = workgroup number in (chunk half, latent, problem) order, `gx` = 2 NC; the launch of its own below, and a
//  role of stage4_kernel in gp_path.hip)
template <int SK, int MZ = 0>      // MZ = 32: inducing extent fixed at compile time (see paths_fwd_split_body)
__device__ __forceinline__ void paths_bwd_split_body(const PathArgs& a, float* smf, int wg_lin, int gx) {
    constexpr int SC = 8;
    __shared__ float red[3][kBlock / VG_WAVE];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int S = a.S, N = a.N, Mz = MZ ? MZ : a.Mz, L = a.L, J = N + Mz;
    int wg = xcd_contiguous(wg_lin, a.xcd_span);
    const int bx = wg % gx;
    wg /= gx;
    const int ch = bx >> 1, half = bx & 1, l = wg % L, p = wg / L;
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

template <int SK, int MZ = 0>
__global__ __launch_bounds__(kBlock, 2) void paths_bwd_split(PathArgs a) {
    extern __shared__ float smf[];
    paths_bwd_split_body<SK, MZ>(a, smf, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)gridDim.x);
}

// The forward path assembly of large batches (Mz = 32, one slab of prior draws, N <= 64 NT): what a latent's chunks share --
// q_sqrt and A^T, the B operands of the two products -- lives in REGISTERS (a lane's fragments of all eight K steps: 8 for
// the two waves that form u = m + eps C^T, 8 NT for every wave's time tiles), read once per workgroup; LDS holds only the
// noise and the prior draws of the current PAIR of 8-sample chunks (15 KB instead of 36), the 16 x 16 tiles use all sixteen
// rows (paths_fwd_sc8 leaves eight empty).  A row of a tile depends on that row's operands alone: the same numbers as
// paths_fwd_sc8, bit for bit.  (paths_fwd_sc8 re-staged 17 KB of constants per 8 samples: 77 us at the config-5 share.)
template <int NT>
__global__ __launch_bounds__(kBlock, 4) void paths_fwd_regs(PathArgs a) {
    constexpr int SC = 8, Mz = 32, R2 = 2 * SC;
    extern __shared__ float smf[];
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *a.tick += 1u;
    const int l = blockIdx.y, p = blockIdx.z, tid = threadIdx.x, nt = kBlock;
#ifdef VGPMP_BISECT
    // (measurement build: start / end of the workgroups of latent 0 of every 16th problem -- ids 1960 / 1990 + 2 (p / 16) + x; tools/step_trace.py)
    struct WgStamp {
        int id;
        __device__ WgStamp(int base, int l, int p, int x) : id(-1) { if (l == 0 && (p & 15) == 0 && p < 128 && x < 2) { id = base + 2 * (p >> 4) + x; VG_T(true, id); } }
        __device__ ~WgStamp() { if (id >= 0) VG_T(true, id + 30); }
    } wg_stamp_(1960, l, p, (int)blockIdx.x);
#endif
    const int S = a.S, N = a.N, L = a.L, J = N + Mz;
    const float iMz = 1.0f / (float)Mz;
    const size_t pl = (size_t)p * L + l;
    float* cur = smf;
    auto take = [&](int n) { float* q = cur; cur += (n + 3) & ~3; return q; };      // 16-byte aligned regions
    float* es2 = take(2 * R2 * Mz);                  // [16][Mz] eps, then [16][Mz] eps'
    float* e2s2 = es2 + R2 * Mz;
    float* f0s2 = take(R2 * J);                      // [16][J]  prior draws
    float* rs2 = take(R2 * Mz);                      // [16][Mz]
    const int wv = tid >> 6, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    // ---- B fragments of every K step, once per workgroup
    float bAT[NT][8], bC[8];
    int ncol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ncol[t] = 16 * (wv + 4 * t) + i;
        const float* ATg = a.AT + pl * N * Mz + min(ncol[t], N - 1);
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) bAT[t][k8] = ATg[(size_t)(4 * k8 + kk) * N];
    }
    const int mi = 16 * (wv & 1) + i;
    const float m0 = a.m[pl * Mz + mi];
    {
        const float* Cg = a.C + pl * Mz * Mz + mi * Mz;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) bC[k8] = Cg[4 * k8 + kk];
    }
    const int cp_end = min((int)(blockIdx.x + 1) * a.cpw, a.NC);
    for (int ch0 = blockIdx.x * a.cpw; ch0 < cp_end; ch0 += 2) {
        const int s_base = ch0 * SC;
        if (a.epsT)
            vg_stage_rows(es2, 2 * R2, Mz, tid, nt, [&](int r) -> const float* {
                const int second = r >= R2, s = s_base + (second ? r - R2 : r);
                return s < S ? (second ? a.eps2T : a.epsT) + (pl * S + s) * Mz : nullptr;                      // zero beyond S
            });
        else
            vg_stage_words(es2, 2 * R2 * Mz, tid, nt, [&](int w) -> const void* {
                const int second = w >= R2 * Mz, e = second ? w - R2 * Mz : w;
                const int sl = vg_div(e, iMz), k = e - sl * Mz, s = s_base + sl;
                return s < S ? (second ? a.eps2 : a.eps) + (((size_t)p * S + s) * Mz + k) * L + l : nullptr;
            });
        vg_stage_rows(f0s2, R2, J, tid, nt, [&](int r) -> const float* {
            return a.F0 + (((size_t)p * S + min(s_base + r, S - 1)) * L + l) * J;
        });
        vg_dma_wait();
        __syncthreads();
        if (wv < 2) {
            // u = m + eps C^T on sixteen samples, this wave's sixteen inducing points; r = u - f0(Z) - sqrt(jitter) eps'
            float av[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) av[k8] = es2[i * Mz + 4 * k8 + kk];
            vg_f32x4_t acc = {m0, m0, m0, m0};
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k8], bC[k8], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sl = 4 * kk + q, e = sl * Mz + mi, s = s_base + sl;
                const float r = vg_path_r(acc[q], f0s2[sl * J + N + mi], a.sqrt_jitter, e2s2[e]);
                rs2[e] = r;
                if (s < S) vg_stream(a.R + (((size_t)p * S + s) * L + l) * Mz + mi, r);
            }
        }
        __syncthreads();
        {
            // f = f0(X) + r A^T: this wave's time tiles
            float rv[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) rv[k8] = rs2[i * Mz + 4 * k8 + kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (16 * (wv + 4 * t) < N) {             // (wave-uniform)
                    const int n = min(ncol[t], N - 1);
                    vg_f32x4_t acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = f0s2[(4 * kk + q) * J + n];
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rv[k8], bAT[t][k8], acc, 0, 0, 0);
                    if (ncol[t] < N) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int s = s_base + 4 * kk + q;
                            if (s < S) vg_stream(a.f + (((size_t)p * S + s) * L + l) * N + n, acc[q]);
                        }
                    }
                }
            }
        }
        __syncthreads();      // the pair's operands are overwritten by the next one's
    }
}

template <int SK, bool RAW>
__global__ __launch_bounds__(kBlock, 2) void paths_fwd_sc8(PathArgs a) {
    extern __shared__ float smf[];
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *a.tick += 1u;
    if (SK > 1 && a.nsplit == 2) {
        if constexpr (SK > 1) {
            if (a.Mz == 32) paths_fwd_split_body<SK, 32>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
            else paths_fwd_split_body<SK>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
        }
        return;
    }
    paths_fwd_body<SK, 8, RAW>(a, smf, blockIdx.x, blockIdx.y, blockIdx.z);
}

}  // namespace
