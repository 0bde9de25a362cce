// Gradient assembly, hyper-parameter and q_mu / q_sqrt Adam updates, ELBO pieces.
// Private part of gp_path.hip (one translation unit: the stage launches call these bodies by role).
#pragma once

namespace {

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
    int dma;                  // the chunk partials go to LDS by DMA, `dma` chunks per pass (a multiple of 8, or all NC of them; 0: summed from global memory)
    int split;                // final_kernel / the merged launches: kFinSplit workgroups per (latent, problem), by column strips
    int tshift;               // measurement builds: added to the stamp ids (the role inside stage 1 vs the stand-alone launch)
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
#pragma clang fp contract(off)      // hyper_update_wave() must round identically
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

// The same update by a WHOLE WAVE (all 64 lanes must call it, converged) for the callers that have one to spare -- the
// prologue forms in stage 1, where this sits on the critical chain of the step.  The scalar form above is ~1500
// instructions for one lane (48 loads, 3 x 31 additions, the Adam arithmetic twice); here lane (j, k) = (lane >> 4,
// lane & 15) loads partial j of chunk k -- every load of the update in ONE request per lane -- the three sums are
// butterflies over 16 lanes, which add the SAME operand pairs as the tree of sum_chunks() (IEEE addition commutes
// exactly), and the two independent scalar tails (lengthscale, variance: same arithmetic, different data) run side by
// side in the lower and upper half of the wave.  Bit-identical to hyper_update(); every lane returns the full state.
// value of a CONSTANT lane (v_readlane: an SGPR broadcast, no LDS round trip like ds_bpermute)
template <int LANE>
__device__ __forceinline__ double vg_lane_f64(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
    return __hiloint2double(hi, lo);
}
// DPP move of a double (both halves): quad permutes, half-row mirror, row rotate -- a few cycles each
template <int CTRL>
__device__ __forceinline__ double vg_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ HyperState hyper_update_wave(const HyperArgs& h, size_t pl, bool own_lr = false, double lr_own = 0.0) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & (VG_WAVE - 1), j = min(lane >> 4, 2), k = lane & 15;
    const bool isv = lane >= 32;                 // upper half: variance, lower half: lengthscale
    const float* part = h.part + pl * h.NC * h.part_len + (h.Mz + h.Mz * h.Mz);
    const double* mp = h.do_adam ? (isv ? h.m_var : h.m_ell) : (isv ? h.p_var : h.p_ell);
    const double* vp = h.do_adam ? (isv ? h.v_var : h.v_ell) : (isv ? h.p_var : h.p_ell);
    // every operand requested in one go
    const float* q0 = part + (size_t)min(k, h.NC - 1) * h.part_len;
    float va = q0[j], vb = q0[4 + j];
    double raw = (isv ? h.p_var : h.p_ell)[pl], m = mp[pl], v = vp[pl];
    const double gkl = (isv ? h.gkl_var : h.gkl_ell)[pl], sig = (isv ? h.sig_var : h.sig_ell)[pl], var = h.var[pl];
    const double lr_dev = h.lr_dev[0];
    const double lr_t = own_lr ? lr_own : ((h.do_adam && h.use_lr_dev) ? lr_dev : h.lr_t);
    if (!h.do_adam) m = v = 0.0;
    double s = 0.0;
    for (int c0 = 0; c0 < h.NC; c0 += 16) {
        if (c0 > 0) {
            const float* q = part + (size_t)min(c0 + k, h.NC - 1) * h.part_len;
            va = q[j]; vb = q[4 + j];
        }
        double d = c0 + k < h.NC ? (double)(va + vb) : 0.0;        // the two halves of paths_bwd_split, then the chunk tree
        d += vg_dpp_f64<0xB1>(d);                        // quad_perm [1, 0, 3, 2]: lane ^ 1
        d += vg_dpp_f64<0x4E>(d);                        // quad_perm [2, 3, 0, 1]: lane ^ 2
        d += vg_dpp_f64<0x141>(d);                       // row_half_mirror: the other quad of the eight (all four lanes of a quad agree)
        const double other = vg_dpp_f64<0x128>(d);       // row_ror 8: the other eight of the sixteen
        s += (lane & 8) ? other : d;                     // + chunks 0..7 of the pass, then + chunks 8..15, as sum_chunks() does
        s += (lane & 8) ? d : other;
    }
    const double s0 = vg_lane_f64<0>(s), s1 = vg_lane_f64<16>(s), s2 = vg_lane_f64<32>(s);
    const double g_ell = ((h.want_dell ? s0 : 0.0) + h.kl_scale * gkl) * sig;
    const double g_var = (s1 + s2 / (2.0 * var) + h.kl_scale * gkl) * sig;
    const double g = isv ? g_var : g_ell;
    const bool upd = h.do_adam && (h.trainable & (isv ? VGPMP_TRAIN_KERNEL_VARIANCE : VGPMP_TRAIN_LENGTHSCALES));
    double nraw = raw, nm = m, nv = v;
    adam_update(&nraw, &nm, &nv, g, lr_t);
    if (upd) { raw = nraw; m = nm; v = nv; }
    HyperState o;
    o.raw_ell = vg_lane_f64<0>(raw); o.m_ell = vg_lane_f64<0>(m); o.v_ell = vg_lane_f64<0>(v); o.g_ell = vg_lane_f64<0>(g);
    o.raw_var = vg_lane_f64<32>(raw); o.m_var = vg_lane_f64<32>(m); o.v_var = vg_lane_f64<32>(v); o.g_var = vg_lane_f64<32>(g);
    return o;
}

// the same update with the state already in registers
__device__ __forceinline__ void adam_apply(double* x, double* m, double* v, double x0, double m0, double v0, double g,
                                           double lr_t) {
#pragma clang fp contract(off)
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
// MZCAP: the largest Mz the instantiation serves -- its per-thread register arrays are sized for it (48: 9 + 10 elements per
// thread, 138-167 registers: three workgroups per CU; 32 -- every reference problem set -- : 4 + 5 elements, under 128: four)
template <int MZCAP = VGPMP_MAX_MZ>
__device__ __forceinline__ void final_body(const FinalArgs& b, double* sm, int l, int p) {
    constexpr int kFinRegs = ((MZCAP - 2) * (MZCAP - 1) + kBlock - 1) / kBlock;      // elements per thread
    VG_STOP(b, 7);
    const int tid = threadIdx.x, nt = blockDim.x;
    VG_T(l == 0 && p == 0, 110 + b.tshift);
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
    const int cpp = b.dma;           // chunks per DMA pass
    if (cpp) vg_stage_rows(raw, min(cpp, b.NC), np, tid, nt, [&](int c) -> const float* { return part + (size_t)c * b.part_len; });
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
    if (!cpp)
        for (int e = tid; e < np; e += nt) {
            const double s = sum_chunks(part, b.part_len, b.NC, e);
            if (e < Mz) dmv[e] = s;
            else dC[e - Mz] = s;
        }
    VG_T(l == 0 && p == 0, 115 + b.tshift);
    vg_dma_wait();
    __syncthreads();
    VG_T(l == 0 && p == 0, 114 + b.tshift);
    if (cpp) {
        // running sums of this thread's elements e = tid + j nt, passes of `cpp` chunks (sample-sharded runs on few ranks have
        // more chunks than LDS holds: 128 at S = 1024; summed from global memory that cost 16 dependent round trips)
        constexpr int kSumRegs = (MZCAP + MZCAP * MZCAP + kBlock - 1) / kBlock;
        double sacc[kSumRegs];
#pragma unroll
        for (int j = 0; j < kSumRegs; ++j) sacc[j] = 0.0;
        if (cpp == 4) {
            // passes of FOUR chunks (half the LDS of eight: a fourth workgroup per CU): sum_chunks() adds groups of eight as
            // ((d0 + d1) + (d2 + d3)) + ((d4 + d5) + (d6 + d7)) -- the first half sum is carried from the even pass to the odd one,
            // so the additions and their order are unchanged (a missing half is the +0.0 the eight-chunk form adds there)
            double half[kSumRegs];
            for (int cb = 0; cb < b.NC; cb += 4) {
                const int nc = min(4, b.NC - cb);
                const bool odd = (cb & 4) != 0;
                if (cb > 0) {
                    __syncthreads();                     // the previous pass has been read
                    vg_stage_rows(raw, nc, np, tid, nt, [&](int c) -> const float* { return part + (size_t)(cb + c) * b.part_len; });
                    vg_dma_wait();
                    __syncthreads();
                }
#pragma unroll
                for (int j = 0; j < kSumRegs; ++j) {
                    const int e = tid + j * nt;
                    if (e < np) {
                        double d[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] = k < nc ? (double)raw[(size_t)min(k, nc - 1) * np + e] : 0.0;
                        const double h = (d[0] + d[1]) + (d[2] + d[3]);
                        if (odd) sacc[j] += half[j] + h;
                        else if (cb + 4 >= b.NC) sacc[j] += h + ((0.0 + 0.0) + (0.0 + 0.0));
                        else half[j] = h;
                    }
                }
            }
        } else
        for (int cb = 0; cb < b.NC; cb += cpp) {
            const int nc = min(cpp, b.NC - cb);
            if (cb > 0) {
                __syncthreads();                     // the previous pass has been read
                vg_stage_rows(raw, nc, np, tid, nt, [&](int c) -> const float* { return part + (size_t)(cb + c) * b.part_len; });
                vg_dma_wait();
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < kSumRegs; ++j) {
                const int e = tid + j * nt;
                if (e < np) {
                    double s = sacc[j];
                    for (int c0 = 0; c0 < nc; c0 += 8) {       // the order of sum_chunks()
                        double d[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) d[k] = c0 + k < nc ? (double)raw[(size_t)min(c0 + k, nc - 1) * np + e] : 0.0;
                        s += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
                    }
                    sacc[j] = s;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kSumRegs; ++j) {
            const int e = tid + j * nt;
            if (e < np) {
                if (e < Mz) dmv[e] = sacc[j];
                else dC[e - Mz] = sacc[j];
            }
        }
        __syncthreads();
    }
    VG_STOP(b, 6);
    VG_T(l == 0 && p == 0, 111 + b.tshift);
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
#pragma unroll 4
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
    VG_T(l == 0 && p == 0, 112 + b.tshift);
    if (l == 0) elbo_pieces(b.lik_partial, b.nblk, b.kl_l, L, p, b.alpha_fin ? b.alpha_fin[p] : b.lik_scale, kls, b.out_lik, b.out_kl);
    VG_T(l == 0 && p == 0, 113 + b.tshift);
}

// The same assembly on kFinSplit workgroups per (latent, problem), for the few-problem schedule where one workgroup
// pulling all NC chunk partials of a latent (72 KB at Mz = 32) through one CU is the longest role of its launch.
// Column c of the q_sqrt gradient tril(Lk^T dC)[2:, 2:] needs column c + 2 of dC only, so workgroup q takes the
// W = Mz / kFinSplit columns [q W, (q + 1) W) of dC: it stages that strip of every chunk partial (NC Mz rows of W floats)
// and the factor, sums in the order of sum_chunks() and forms its columns -- bit-identical to final_body().  q_mu
// (which needs the Mz-vector dm) rides with q = 0, whose first two columns are the fixed end points; the ELBO pieces
// with the last strip of latent 0.  Needs Mz % (4 kFinSplit) == 0 (16-byte strips).
constexpr int kFinSplit = 4;
__device__ __forceinline__ void final_cols_body(const FinalArgs& b, double* sm, int q, int l, int p) {
    const int tid = threadIdx.x, nt = blockDim.x;
    VG_T(q == 0 && l == 0 && p == 0, 110 + b.tshift);
    const int M = b.M, Mz = M + 2, L = b.L, W = Mz / kFinSplit, NC = b.NC;
    const float iW = 1.0f / (float)W, iMz = 1.0f / (float)Mz;
    const size_t pl = (size_t)p * L + l;
    double* dC = sm;                 // [Mz][W] columns q W ... of the summed dC
    double* dmv = dC + Mz * W;       // [Mz]
    float* Lks = reinterpret_cast<float*>(dmv + Mz + (Mz & 1));   // [Mz][Mz] chol factor
    float* raw = Lks + ((Mz * Mz + 3) & ~3);                      // [NC][Mz][W] strips of the chunk partials
    float* rawm = raw + (size_t)NC * Mz * W;                      // [NC][Mz] dm partials (q = 0)
    const float* part = b.part + pl * NC * b.part_len;
    const double lr_t = b.do_adam ? (b.use_lr_dev ? b.lr_dev[0] : b.lr_t) : 0.0;
    vg_stage_rows(raw, NC * Mz, W, tid, nt, [&](int r) -> const float* {
        const int c = vg_div(r, iMz), i = r - c * Mz;
        return part + (size_t)c * b.part_len + Mz + (size_t)i * Mz + q * W;
    });
    vg_stage_rows(Lks, 1, Mz * Mz, tid, nt, [&](int) -> const float* { return b.Lk32 + pl * Mz * Mz; });
    if (q == 0) vg_stage_rows(rawm, NC, Mz, tid, nt, [&](int c) -> const float* { return part + (size_t)c * b.part_len; });
    // this thread's elements: q = 0 has q_mu first; then (r, w) over M rows x W columns of the strip
    const int nmu = q == 0 ? M : 0, nel = nmu + M * W;
    double kg[2], xs[2], mo[2], vo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = min(tid + j * nt, nel - 1);
        const bool mu = k < nmu;
        const int idx = k - nmu, r = vg_div(max(idx, 0), iW), c = max(q * W + (idx - r * W) - 2, 0);
        const size_t o = mu ? pl * M + k : pl * M * M + (size_t)r * M + c;
        kg[j] = (mu ? b.gkl_qmu : b.gkl_Q)[o];
        if (b.do_adam) {
            xs[j] = (mu ? b.pq_mu : b.pq_sqrt)[o];
            mo[j] = (mu ? b.mq_mu : b.mq_sqrt)[o];
            vo[j] = (mu ? b.vq_mu : b.vq_sqrt)[o];
        }
    }
    VG_T(q == 0 && l == 0 && p == 0, 115 + b.tshift);
    vg_dma_wait();
    __syncthreads();
    VG_T(q == 0 && l == 0 && p == 0, 114 + b.tshift);
    for (int e = tid; e < Mz * W + (q == 0 ? Mz : 0); e += nt) {
        const bool isd = e >= Mz * W;
        const float* src = isd ? rawm + (e - Mz * W) : raw + e;
        const int stride = isd ? Mz : Mz * W;
        double s = 0.0;
        for (int c0 = 0; c0 < NC; c0 += 8) {             // the order of sum_chunks()
            double d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = c0 + k < NC ? (double)src[(size_t)min(c0 + k, NC - 1) * stride] : 0.0;
            s += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
        }
        if (isd) dmv[e - Mz * W] = s;
        else dC[e] = s;
    }
    __syncthreads();
    VG_T(q == 0 && l == 0 && p == 0, 111 + b.tshift);
    const double kls = b.kl_scale;
    double* gQ = b.g_qsqrt + pl * M * M;
    double* gm = b.g_qmu + pl * M;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = tid + j * nt;
        if (k >= nel) continue;
        if (k < nmu) {
            const double g = dmv[k + 2] + kls * kg[j];
            gm[k] = g;
            if (b.do_adam && (b.trainable & VGPMP_TRAIN_Q_MU))
                adam_apply(b.pq_mu + pl * M + k, b.mq_mu + pl * M + k, b.vq_mu + pl * M + k, xs[j], mo[j], vo[j], g, lr_t);
            continue;
        }
        const int idx = k - nmu, r = vg_div(idx, iW), w = idx - r * W, c = q * W + w - 2;
        if (c < 0) continue;                             // the two fixed end points: no variable
        const int e = r * M + c;
        double g = 0.0;
        if (c <= r) {
            double s0 = 0.0, s1 = 0.0;
            int i = r + 2;
            for (; i + 1 < Mz; i += 2) {
                s0 = fma((double)Lks[i * Mz + (r + 2)], dC[i * W + w], s0);
                s1 = fma((double)Lks[(i + 1) * Mz + (r + 2)], dC[(i + 1) * W + w], s1);
            }
            if (i < Mz) s0 = fma((double)Lks[i * Mz + (r + 2)], dC[i * W + w], s0);
            g = s0 + s1 + kls * kg[j];
        }
        gQ[e] = g;
        if (c <= r && b.do_adam && (b.trainable & VGPMP_TRAIN_Q_SQRT))
            adam_apply(b.pq_sqrt + pl * M * M + e, b.mq_sqrt + pl * M * M + e, b.vq_sqrt + pl * M * M + e, xs[j], mo[j], vo[j],
                       g, lr_t);
    }
    VG_T(q == 0 && l == 0 && p == 0, 112 + b.tshift);
    if (l == 0 && q == kFinSplit - 1)
        elbo_pieces(b.lik_partial, b.nblk, b.kl_l, L, p, b.alpha_fin ? b.alpha_fin[p] : b.lik_scale, kls, b.out_lik, b.out_kl);
    VG_T(q == 0 && l == 0 && p == 0, 113 + b.tshift);
}

__global__ __launch_bounds__(kBlock) void final_kernel(FinalArgs b) {      // grid (L, P), or (L kFinSplit, P) with b.split
    extern __shared__ double sm[];
    if (b.split) final_cols_body(b, sm, blockIdx.x % kFinSplit, blockIdx.x / kFinSplit, blockIdx.y);
    else final_body(b, sm, blockIdx.x, blockIdx.y);
}

// forward-only epilogue: ELBO pieces without the reverse pass
__global__ __launch_bounds__(kBlock) void elbo_pieces_kernel(int L, int nblk, const float* __restrict__ lik_partial,
                                                              const double* __restrict__ kl_l, double lik_scale, double kls,
                                                              double* __restrict__ out_lik, double* __restrict__ out_kl,
                                                              const double* __restrict__ alpha_fin) {
    elbo_pieces(lik_partial, nblk, kl_l, L, blockIdx.x, alpha_fin ? alpha_fin[blockIdx.x] : lik_scale, kls, out_lik, out_kl);
}

// stand-alone Adam over the packed variables (sample-sharded mode, after the gradient all-reduce): ONE launch for the
// (up to) four tensors -- the blocks of segment k follow those of segment k - 1 (four launches cost ~8 us each on the
// one-problem step of BASELINE config 4, more than the update itself)
struct AdamAllArgs {
    int nseg;
    unsigned first_block[5];      // segment k owns blocks [first_block[k], first_block[k + 1])
    size_t n[4];
    double *x[4], *m[4], *v[4];
    const double* g[4];
    int tril_M[4];
    double lr_t;
};
__global__ __launch_bounds__(kBlock) void adam_all_kernel(AdamAllArgs a) {
    int k = 0;
    while (k + 1 < a.nseg && blockIdx.x >= a.first_block[k + 1]) ++k;
    const size_t i = (size_t)(blockIdx.x - a.first_block[k]) * kBlock + threadIdx.x;
    if (i >= a.n[k]) return;
    const int M = a.tril_M[k];
    if (M > 0) {
        const int e = (int)(i % ((size_t)M * M));
        if (e % M > e / M) return;
    }
    adam_update(a.x[k] + i, a.m[k] + i, a.v[k] + i, a.g[k][i], a.lr_t);
}

}  // namespace
