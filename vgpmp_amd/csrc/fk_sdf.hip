// Forward kinematics -> sphere placement -> signed-distance lookup -> hinge likelihood, forward and
// reverse, for gfx950.  One lane owns one joint configuration (one (sample, time) pair): the DH chain
// lives in registers, every sphere costs ONE 16-byte gather from the {d, grad} voxel table, and the
// reverse pass is the geometric Jacobian accumulated per frame (force / moment sums), so no
// intermediate leaves the registers.
//
// Reference path: likelihoods/likelihood.py:57-176, utils/sampler.py:103-120,142-244,
// utils/sdf_utils.py:62-136.
#include "vgpmp_device.h"
#include "fk_chain.h"
#include <type_traits>
#include <hip/hip_ext.h>

// No implicit contraction in this file: every fused multiply-add is written as fmaf, so that template instantiations
// that differ only in WHERE a voxel record comes from (layout, free-space summary) round identically.
#pragma clang fp contract(off)

#ifdef VGPMP_BISECT
#include <stdlib.h>
static int lik_bisect_mode() { const char* e = getenv("VGPMP_STOP_LIK"); return e ? atoi(e) : 0; }
int vg_trace_take_lik(unsigned long long* host, int cap) { return vg_trace_take(host, cap); }
#endif


namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ vg_sdf_dev load_sdf(const vgpmp_sdf& s) { return vg_load_sdf(s); }

// ---- voxel index: float32 fast path, exact float64 fallback near cell boundaries -----------------
// The reference computes trunc(((p - offset) - origin) / delta) in float64.  A float64 division costs
// ~600 issue cycles per sphere on this part, so the quotient is first formed in float32 from a
// hi/lo split of (offset + origin): its error is < 1e-6 (|q| + 1), and whenever q is farther than that
// from an integer the truncated float32 value IS the reference index.  Otherwise (a few 1e-4 of the
// queries) the reference's float64 expression is evaluated, so indices stay bit-identical.
struct SdfFast {
    float chx, chy, chz, clx, cly, clz, inv_delta;
};

__device__ __forceinline__ SdfFast make_fast(const vg_sdf_dev& s, double offx, double offy, double offz) {
    SdfFast f;
    const double cx = offx + s.ox, cy = offy + s.oy, cz = offz + s.oz;
    f.chx = (float)cx; f.clx = (float)(cx - (double)f.chx);
    f.chy = (float)cy; f.cly = (float)(cy - (double)f.chy);
    f.chz = (float)cz; f.clz = (float)(cz - (double)f.chz);
    f.inv_delta = (float)(1.0 / s.delta);
    return f;
}

// The reference index clamp(trunc(RN(num / delta)), 0, n - 1), num = (p - offset) - origin in float64, when the float32
// quotient q is known to lie within its error bound of the integer m = rint(q) -- WITHOUT the float64 division (25
// dependent float64 instructions; one lane of a wave near a cell boundary sends all 64 through this path, ~13 % of the
// sphere iterations at 512 cells per axis).  RN(num / delta) >= m  <=>  num / delta >= m - u, u = half the distance from m
// to the float64 below it (no float64 lies strictly between m - 2u and m, and a quotient exactly on m - u would need
// num = delta (m - u), 100+ significant bits) <=>  num - m delta >= -u delta: ONE fma (exact difference, rounded once:
// sign and size right to 2^-53 relative) against a power-of-two multiple of delta.  m <= 0 gives 0 (the quotient is below 1),
// m >= n gives n - 1.  Pinned bit for bit on lattice points +- a few float32 ulps by tests/test_gpu_config5.py.
__device__ __forceinline__ int voxel_axis_near(double num, double delta, float q, int n) {
    const float mf = rintf(q);
    const int m = (int)mf;
    const uint32_t fb = __builtin_bit_cast(uint32_t, mf);
    // u delta = 2^(e - 53 - [m is a power of two]) delta, e = floor(log2 m): a float64 with that exponent, mantissa 0
    const int ue = (int)(fb >> 23) - 127 - 53 - ((fb & 0x7fffffu) == 0u ? 1 : 0);
    const double u = __hiloint2double((ue + 1023) << 20, 0);
    const double r = fma(-(double)mf, delta, num);
    const int idx = m - (r >= -(u * delta) ? 0 : 1);
    return m <= 0 ? 0 : (m >= n ? n - 1 : idx);
}

__device__ __forceinline__ int voxel_axis(float pos, float ch, float cl, float inv_delta, int n, double off,
                                          double origin, double delta) {
    const float q = ((pos - ch) - cl) * inv_delta;
    const int hi = n - 1;
    int idx = q < 0.f ? 0 : (q > (float)hi ? hi : (int)q);
    if (fabsf(q - rintf(q)) < 1e-6f * (fabsf(q) + 1.f))
        idx = voxel_axis_near(((double)pos - off) - origin, delta, q, n);      // the reference's index, exactly
    return idx;
}

// Per-configuration scratch in LDS ([slot][configuration], conflict free): the frame loop is a run-time
// loop (dof is a run-time value), so per-frame data cannot live in indexed
// registers.  LPC lanes (1 or 4, adjacent lanes of one wave) share one configuration and its scratch.
struct LikScratch {
    float* base;
    int stride;
    __device__ __forceinline__ float& at(int slot) const { return base[slot * stride]; }
};
// slots: [0, D) sin, [D, 2D) cos, then 6 per frame (F, M)
__device__ __forceinline__ int lik_scratch_slots(int D) { return 2 * D + 6 * (D + 1); }

template <int LPC>
__device__ __forceinline__ float quad_sum(float v) {
    if (LPC >= 2) v += __shfl_xor(v, 1, VG_WAVE);
    if (LPC >= 4) v += __shfl_xor(v, 2, VG_WAVE);
    if (LPC >= 8) v += __shfl_xor(v, 4, VG_WAVE);
    return v;
}

// DPP row shift: lane i reads lane i - R of its row of 16 (lanes without a source read 0)
template <int R>
__device__ __forceinline__ float row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + R, 0xF, 0xF, false));
}
template <int R>
__device__ __forceinline__ Frame frame_shr(const Frame& a) {
    Frame o;
    o.cx = vg_make3(row_shr<R>(a.cx.x), row_shr<R>(a.cx.y), row_shr<R>(a.cx.z));
    o.cy = vg_make3(row_shr<R>(a.cy.x), row_shr<R>(a.cy.y), row_shr<R>(a.cy.z));
    o.cz = vg_make3(row_shr<R>(a.cz.x), row_shr<R>(a.cz.y), row_shr<R>(a.cz.z));
    o.t = vg_make3(row_shr<R>(a.t.x), row_shr<R>(a.t.y), row_shr<R>(a.t.z));
    return o;
}
// rigid transforms: (A B)(p) = A(B(p))
__device__ __forceinline__ Frame frame_mul(const Frame& a, const Frame& b) {
    Frame o;
    o.cx = axpy(b.cx.x, a.cx, lin2(b.cx.y, a.cy, b.cx.z, a.cz));
    o.cy = axpy(b.cy.x, a.cx, lin2(b.cy.y, a.cy, b.cy.z, a.cz));
    o.cz = axpy(b.cz.x, a.cx, lin2(b.cz.y, a.cy, b.cz.z, a.cz));
    o.t = axpy(b.t.x, a.cx, axpy(b.t.y, a.cy, axpy(b.t.z, a.cz, a.t)));
    return o;
}
// the DH link transform of joint j by itself (dh_apply on the identity, products with 0 / 1 folded)
__device__ __forceinline__ Frame dh_link(const vgpmp_robot* __restrict__ rb, int j, float st, float ct) {
    const float4 jt = *reinterpret_cast<const float4*>(rb->joint_tab[j]);
    const float ca = jt.x, sa = jt.y, d = jt.z, a = jt.w;
    Frame o;
    if (rb->craig) {
        o.cx = vg_make3(ct, st * ca, st * sa);
        o.cy = vg_make3(-st, ct * ca, ct * sa);
        o.cz = vg_make3(0.f, -sa, ca);
        o.t = vg_make3(a, -d * sa, d * ca);
    } else {
        o.cx = vg_make3(ct, st, 0.f);
        o.cy = vg_make3(-st * ca, ct * ca, sa);
        o.cz = vg_make3(st * sa, -ct * sa, ca);
        o.t = vg_make3(a * ct, a * st, d);
    }
    return o;
}

// ---- the second sweep's operand fence ------------------------------------------------------------------
// Every iteration of a second sweep over the chain (the gradient's walk: loglik_config*, below) REQUESTS all it reads -- the joint's DH
// row by a scalar load, its sin / cos and the iteration's other LDS words -- and then passes them through ONE explicit
// `s_waitcnt lgkmcnt(0)` inside an asm statement that names them all: the compiler can neither move a load behind that point nor start
// the arithmetic before it, and nothing of the iteration is in flight while it computes.
// History (profiles/r06/flake.md): round 5's gradients came back WRONG for sixteen consecutive lanes in launches that ran beside another
// process's kernels.  In this kernel every form that pinned the loop's schedule cured it (0 of 35 sessions against 8 of 11), which is how
// the fence came to be.  The cause was found later (tools/sweep_probe.hip, tools/depack_pk.py, tools/pk_probe.hip, tools/trigger_probe.py):
// a packed-FP32 instruction (what the SLP vectoriser makes of this arithmetic) that takes source 1's high register for both results reads
// 0.0 there in lanes 48-63 while another wave of the compute unit runs a wide f16 matrix instruction -- this library's own prior draws in
// the other process; the fence had merely changed which products the vectoriser paired.  The library is therefore built WITHOUT the
// vectorisers (vgpmp_amd/build.py: no packed instruction in the binary, held at zero by a CPU test); the fence stays: it costs nothing and
// keeps the sweep's operands at rest.
__device__ __forceinline__ void vg_sweep_fence(float4& jt, float& st, float& ct) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jt.x), "+s"(jt.y), "+s"(jt.z), "+s"(jt.w), "+v"(st), "+v"(ct));
}
__device__ __forceinline__ void vg_sweep_fence(float4& jt, float& st, float& ct, float& a) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jt.x), "+s"(jt.y), "+s"(jt.z), "+s"(jt.w), "+v"(st), "+v"(ct), "+v"(a));
}
__device__ __forceinline__ void vg_sweep_fence(float4& jt, float& st, float& ct, float& a, float& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jt.x), "+s"(jt.y), "+s"(jt.z), "+s"(jt.w), "+v"(st), "+v"(ct), "+v"(a), "+v"(b));
}
__device__ __forceinline__ void vg_sweep_fence(float4& jt, float& st, float& ct, float (&m)[6]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jt.x), "+s"(jt.y), "+s"(jt.z), "+s"(jt.w), "+v"(st), "+v"(ct), "+v"(m[0]), "+v"(m[1]),
                 "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]));
}

__device__ __forceinline__ void lik_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- one sphere query, lean ---------------------------------------------------------------------------
// The three axis quotients in float32, clamp + truncate, and ONE test for "some quotient is within its error bound
// of an integer" that sends the query through the reference's float64 expression (all three axes): a few 1e-3 of
// the queries.  Indices stay bit-identical to utils/sdf_utils.py:62-66.
struct Vox3 { int ix, iy, iz; };
__device__ __forceinline__ Vox3 voxel3(vg_float3 p, const SdfFast& fs, const vg_sdf_dev& s, double offx, double offy,
                                       double offz) {
    const float qx = ((p.x - fs.chx) - fs.clx) * fs.inv_delta;
    const float qy = ((p.y - fs.chy) - fs.cly) * fs.inv_delta;
    const float qz = ((p.z - fs.chz) - fs.clz) * fs.inv_delta;
    Vox3 o;
    o.ix = (int)__builtin_amdgcn_fmed3f(qx, 0.f, (float)(s.nx - 1));
    o.iy = (int)__builtin_amdgcn_fmed3f(qy, 0.f, (float)(s.ny - 1));
    o.iz = (int)__builtin_amdgcn_fmed3f(qz, 0.f, (float)(s.nz - 1));
    // margin over the bound 1e-6 (|q| + 1) on the float32 quotient's error (see SdfFast)
    const float ex = fabsf(qx - rintf(qx)) - fmaf(1.1e-6f, fabsf(qx), 1.1e-6f);
    const float ey = fabsf(qy - rintf(qy)) - fmaf(1.1e-6f, fabsf(qy), 1.1e-6f);
    const float ez = fabsf(qz - rintf(qz)) - fmaf(1.1e-6f, fabsf(qz), 1.1e-6f);
    if (fminf(ex, fminf(ey, ez)) < 0.f) {
        // some axis sits within the float32 error of a cell boundary: the reference's float64 index, exactly, on the axes
        // concerned (voxel_axis_near); the others keep theirs
        if (ex < 0.f) o.ix = voxel_axis_near(((double)p.x - offx) - s.ox, s.delta, qx, s.nx);
        if (ey < 0.f) o.iy = voxel_axis_near(((double)p.y - offy) - s.oy, s.delta, qy, s.ny);
        if (ez < 0.f) o.iz = voxel_axis_near(((double)p.z - offz) - s.oz, s.delta, qz, s.nz);
    }
    return o;
}

// sin and cos of a bounded angle (|x| < ~400: joint limit + DH twist): three-term Cody-Waite reduction by pi/2 and the
// single-precision minimax polynomials of Cephes sinf / cosf on [-pi/4, pi/4] (~1 ulp); a third of the instructions
// of the library's sincosf, whose large-argument path is dead weight here.
__device__ __forceinline__ void vg_sincos(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                          fmaf(-0.5f, z, 1.0f));
    const int n = (int)k;
    const float a = (n & 1) ? pc : ps, b = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -a : a;
    *cs = ((n + 1) & 2) ? -b : b;
}

// log p(e | g) of one configuration on one lane.  The spheres are walked in table order in batches of U: for a
// batch the chain is advanced as far as its spheres need (the sphere index, hence the frame, is uniform over the
// wave: scalar control flow), all U positions and voxel addresses are formed and all U gathers are in flight together
// -- one memory round trip per U spheres instead of one per frame --, then the hinge and the per-frame force / moment
// sums.  With GRAD, d logp / d g_j is handed to `emit(j, value)`.  Returns log p.
// With SIG (trainable sigma_obs) the per-sphere variances come from `sig` and every sphere's c^2 / sigma, weighted by
// sig_w (0 on dead lanes), is summed over the wave and handed to emit_sig(q, total).
// With FAR the brick summary is read first and spheres in free space skip the table.
struct NoSig { __device__ __forceinline__ void operator()(int, float) const {} };
template <bool GRAD, int U, bool SIG = false, bool FAR = false, typename LoadRaw, typename ToAngle, typename Emit,
          typename EmitSig = NoSig>
__device__ __forceinline__ float loglik_config(const vgpmp_robot* __restrict__ rb, const vg_sdf_dev& sdf,
                                               const LikScratch sc, LoadRaw load_raw, ToAngle to_angle, Emit emit,
                                               const float* __restrict__ sig = nullptr, float sig_w = 0.f,
                                               EmitSig emit_sig = NoSig()) {
    static_assert(VGPMP_MAX_SPHERES % U == 0, "a batch of sphere constants never leaves the table");
    const int D = rb->dof, P = rb->num_spheres;
    // every joint's input requested before the first is used (one memory round trip, not one per joint)
    float raw[VGPMP_MAX_DOF];
#pragma unroll
    for (int j = 0; j < VGPMP_MAX_DOF; ++j) raw[j] = load_raw(min(j, D - 1));
    const float eps = rb->epsilon;
    const double offx = rb->scene_offset[0], offy = rb->scene_offset[1], offz = rb->scene_offset[2];
    const SdfFast fs = make_fast(sdf, offx, offy, offz);
#pragma unroll
    for (int j = 0; j < VGPMP_MAX_DOF; ++j) {
        if (j < D) {                                     // uniform
            float st, ct;
            vg_sincos(to_angle(j, raw[j]) + rb->joint_tab[j][4], &st, &ct);
            sc.at(j) = st; sc.at(D + j) = ct;
        }
    }
    Frame T = base_frame(rb);
    int cur = 0;                                         // frame T stands at (issue side)
    int pcur = 0;                                        // frame of the running sums (consumer side)
    vg_float3 F = vg_make3(0.f, 0.f, 0.f), Mo = vg_make3(0.f, 0.f, 0.f);
    vg_float3 Ft = vg_make3(0.f, 0.f, 0.f), Mt = vg_make3(0.f, 0.f, 0.f);
    float acc = 0.f;
    auto flush = [&]() {                                 // sums of frame pcur are complete
        const int o = 2 * D + 6 * pcur;
        sc.at(o) = F.x; sc.at(o + 1) = F.y; sc.at(o + 2) = F.z;
        sc.at(o + 3) = Mo.x; sc.at(o + 4) = Mo.y; sc.at(o + 5) = Mo.z;
        Ft = vg_make3(Ft.x + F.x, Ft.y + F.y, Ft.z + F.z);
        Mt = vg_make3(Mt.x + Mo.x, Mt.y + Mo.y, Mt.z + Mo.z);
        F = vg_make3(0.f, 0.f, 0.f); Mo = vg_make3(0.f, 0.f, 0.f);
        ++pcur;
    };
#pragma nounroll
    for (int q0 = 0; q0 < P; q0 += U) {
        float4 v[U];
        vg_float3 pos[U];
        uint32_t at[U];
        // the batch's constants: contiguous rows of the robot table, uniform addresses (wide scalar loads, one wait)
        float4 ca[U];
        float2 cb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ca[u] = *reinterpret_cast<const float4*>(rb->sphere_a[q0 + u]);      // {offset, frame}; rows >= P: frame = D
            cb[u] = *reinterpret_cast<const float2*>(rb->sphere_b[q0 + u]);      // {radius, 1 / sigma}
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q < P) {                                 // uniform
                const int fr = __builtin_bit_cast(int, ca[u].w);
                while (cur < fr) {
                    dh_apply(rb, cur, sc.at(cur), sc.at(D + cur), T);
                    ++cur;
                }
                pos[u] = axpy(ca[u].x, T.cx, axpy(ca[u].y, T.cy, axpy(ca[u].z, T.cz, T.t)));
                const Vox3 ix = voxel3(pos[u], fs, sdf, offx, offy, offz);
                at[u] = (uint32_t)vg_table_offset(sdf, ix.ix, ix.iy, ix.iz);
                if (FAR) v[u].x = sdf.brick_min[vg_brick_of(sdf, ix.ix, ix.iy, ix.iz)];
                else v[u] = sdf.table[at[u]];
            } else {
                v[u] = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);       // hinge exactly 0
                pos[u] = vg_make3(0.f, 0.f, 0.f);
                at[u] = 0u;
            }
        }
        if (FAR) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (q0 + u < P) {
                    // free space: the hinge is exactly 0 on every voxel of the brick -> no table access
                    const float bm = v[u].x;
                    v[u] = make_float4(bm, 0.f, 0.f, 0.f);
                    if (eps - (bm - cb[u].x) > 0.f) v[u] = sdf.table[at[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q < P) {                                 // uniform
                if (GRAD) {
                    const int fr = __builtin_bit_cast(int, ca[u].w);
                    while (pcur < fr) flush();
                }
                const float c = fmaxf(eps - (v[u].x - cb[u].x), 0.f);           // likelihood.py:131-143
                const float cs = SIG ? c / sig[q] : c * cb[u].y;
                acc = fmaf(cs, c, acc);                                          // likelihood.py:99
                if (SIG) emit_sig(q, vg_wave_sum(cs * c * sig_w));               // one total per sphere
                if (GRAD) {
                    const vg_float3 gp = vg_make3(cs * v[u].y, cs * v[u].z, cs * v[u].w);   // d logp / d pos
                    F = vg_make3(F.x + gp.x, F.y + gp.y, F.z + gp.z);
                    Mo = vg_cross_acc(Mo, pos[u], gp);
                }
            }
        }
    }
    if (GRAD) {
        while (pcur <= D) flush();
        // second sweep over the chain (sin/cos kept): joint i turns about z of frame i (Craig) or frame
        // i-1 (classic) and moves every sphere on frames >= i, i.e. the totals minus the prefix < i
        const bool craig = rb->craig != 0;
        T = base_frame(rb);
        vg_float3 Fs = Ft, Ms = Mt;
#pragma nounroll
        for (int i = 1; i <= D; ++i) {
            const int o = 2 * D + 6 * (i - 1);
            // the iteration's operands, all requested, then at rest (vg_sweep_fence)
            float4 jt = *reinterpret_cast<const float4*>(rb->joint_tab[i - 1]);
            float st = sc.at(i - 1), ct = sc.at(D + i - 1);
            float m[6] = {sc.at(o), sc.at(o + 1), sc.at(o + 2), sc.at(o + 3), sc.at(o + 4), sc.at(o + 5)};
            vg_sweep_fence(jt, st, ct, m);
            Fs = vg_make3(Fs.x - m[0], Fs.y - m[1], Fs.z - m[2]);
            Ms = vg_make3(Ms.x - m[3], Ms.y - m[4], Ms.z - m[5]);
            vg_float3 z = T.cz, org = T.t;
            dh_apply_row(jt, craig, st, ct, T);
            if (craig) { z = T.cz; org = T.t; }
            const vg_float3 oxF = vg_cross(org, Fs);
            emit(i - 1, vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z)));
        }
    }
    return -0.5f * acc;
}

// The same walk with the per-frame force / moment sums in REGISTERS, as six 16-wide vectors indexed by the (wave-uniform)
// frame number -- indirect register addressing -- and only sin / cos / d g / d f of the joints in LDS: 3 D instead of
// 9 D + 6 words per lane (132 at 14 joints, which held the one-lane form at one wave per SIMD).  Up to 15 joints.
typedef float vg_f32x16 __attribute__((ext_vector_type(16)));
template <int U, bool SIG, bool FAR, typename LoadRaw, typename ToAngle, typename Emit, typename EmitSig = NoSig>
__device__ __forceinline__ float loglik_config_regs(const vgpmp_robot* __restrict__ rb, const vg_sdf_dev& sdf,
                                                    const LikScratch sc, LoadRaw load_raw, ToAngle to_angle, Emit emit,
                                                    const float* __restrict__ sig = nullptr, float sig_w = 0.f,
                                                    EmitSig emit_sig = NoSig()) {
    static_assert(VGPMP_MAX_SPHERES % U == 0, "a batch of sphere constants never leaves the table");
    const int D = rb->dof, P = rb->num_spheres;          // D <= 15: frames 0 .. 15
    float raw[VGPMP_MAX_DOF];
#pragma unroll
    for (int j = 0; j < VGPMP_MAX_DOF; ++j) raw[j] = load_raw(min(j, D - 1));
    const float eps = rb->epsilon;
    const double offx = rb->scene_offset[0], offy = rb->scene_offset[1], offz = rb->scene_offset[2];
    const SdfFast fs = make_fast(sdf, offx, offy, offz);
    // sin / cos / d g / d f of every joint: LDS slots [0, D), [D, 2D), [2D, 3D) of this lane
#pragma unroll
    for (int j = 0; j < VGPMP_MAX_DOF; ++j) {
        if (j < D) {                                     // uniform
            float st, ct, d;
            vg_sincos(to_angle(j, raw[j], d) + rb->joint_tab[j][4], &st, &ct);
            sc.at(j) = st; sc.at(D + j) = ct; sc.at(2 * D + j) = d;
        }
    }
    vg_f32x16 fx = 0.f, fy = 0.f, fz = 0.f, mx = 0.f, my = 0.f, mz = 0.f;      // per-frame sums
    Frame T = base_frame(rb);
    int cur = 0;                                         // frame T stands at (issue side)
    int pcur = 0;                                        // frame of the running sums (consumer side)
    vg_float3 F = vg_make3(0.f, 0.f, 0.f), Mo = vg_make3(0.f, 0.f, 0.f);
    vg_float3 Ft = vg_make3(0.f, 0.f, 0.f), Mt = vg_make3(0.f, 0.f, 0.f);
    float acc = 0.f;
    auto flush = [&]() {                                 // sums of frame pcur are complete
        fx[pcur] = F.x; fy[pcur] = F.y; fz[pcur] = F.z; mx[pcur] = Mo.x; my[pcur] = Mo.y; mz[pcur] = Mo.z;
        Ft = vg_make3(Ft.x + F.x, Ft.y + F.y, Ft.z + F.z);
        Mt = vg_make3(Mt.x + Mo.x, Mt.y + Mo.y, Mt.z + Mo.z);
        F = vg_make3(0.f, 0.f, 0.f); Mo = vg_make3(0.f, 0.f, 0.f);
        ++pcur;
    };
#pragma nounroll
    for (int q0 = 0; q0 < P; q0 += U) {
        float4 v[U];
        vg_float3 pos[U];
        uint32_t at[U];
        float4 ca[U];
        float2 cb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ca[u] = *reinterpret_cast<const float4*>(rb->sphere_a[q0 + u]);      // {offset, frame}; rows >= P: frame = D
            cb[u] = *reinterpret_cast<const float2*>(rb->sphere_b[q0 + u]);      // {radius, 1 / sigma}
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q < P) {                                 // uniform
                const int fr = __builtin_bit_cast(int, ca[u].w);
                while (cur < fr) {
                    dh_apply(rb, cur, sc.at(cur), sc.at(D + cur), T);
                    ++cur;
                }
                pos[u] = axpy(ca[u].x, T.cx, axpy(ca[u].y, T.cy, axpy(ca[u].z, T.cz, T.t)));
                const Vox3 ix = voxel3(pos[u], fs, sdf, offx, offy, offz);
                at[u] = (uint32_t)vg_table_offset(sdf, ix.ix, ix.iy, ix.iz);
                if (FAR) v[u].x = sdf.brick_min[vg_brick_of(sdf, ix.ix, ix.iy, ix.iz)];
                else v[u] = sdf.table[at[u]];
            } else {
                v[u] = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);       // hinge exactly 0
                pos[u] = vg_make3(0.f, 0.f, 0.f);
                at[u] = 0u;
            }
        }
        if (FAR) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (q0 + u < P) {
                    // free space: the hinge is exactly 0 on every voxel of the brick -> no table access
                    const float bm = v[u].x;
                    v[u] = make_float4(bm, 0.f, 0.f, 0.f);
                    if (eps - (bm - cb[u].x) > 0.f) v[u] = sdf.table[at[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q < P) {                                 // uniform
                const int fr = __builtin_bit_cast(int, ca[u].w);
                while (pcur < fr) flush();
                const float c = fmaxf(eps - (v[u].x - cb[u].x), 0.f);           // likelihood.py:131-143
                const float cs = SIG ? c / sig[q] : c * cb[u].y;
                acc = fmaf(cs, c, acc);                                          // likelihood.py:99
                if (SIG) emit_sig(q, vg_wave_sum(cs * c * sig_w));               // one total per sphere
                const vg_float3 gp = vg_make3(cs * v[u].y, cs * v[u].z, cs * v[u].w);   // d logp / d pos
                F = vg_make3(F.x + gp.x, F.y + gp.y, F.z + gp.z);
                Mo = vg_cross_acc(Mo, pos[u], gp);
            }
        }
    }
    while (pcur <= D) flush();
    // second sweep over the chain: joint i turns about z of frame i (Craig) or frame i-1 (classic) and moves every
    // sphere on frames >= i, i.e. the totals minus the prefix < i
    const bool craig = rb->craig != 0;
    T = base_frame(rb);
    vg_float3 Fs = Ft, Ms = Mt;
#pragma nounroll
    for (int i = 1; i <= D; ++i) {
        // the iteration's operands, all requested, then at rest (vg_sweep_fence)
        float4 jt = *reinterpret_cast<const float4*>(rb->joint_tab[i - 1]);
        float st = sc.at(i - 1), ct = sc.at(D + i - 1), dgdf = sc.at(2 * D + i - 1);
        vg_sweep_fence(jt, st, ct, dgdf);
        Fs = vg_make3(Fs.x - fx[i - 1], Fs.y - fy[i - 1], Fs.z - fz[i - 1]);
        Ms = vg_make3(Ms.x - mx[i - 1], Ms.y - my[i - 1], Ms.z - mz[i - 1]);
        vg_float3 z = T.cz, org = T.t;
        dh_apply_row(jt, craig, st, ct, T);
        if (craig) { z = T.cz; org = T.t; }
        const vg_float3 oxF = vg_cross(org, Fs);
        emit(i - 1, vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z)) * dgdf);
    }
    return -0.5f * acc;
}

// The register form as a two-stage software pipeline over batches of U spheres.  Measured on the config-5 share (r04): the
// duration of the register form is its vector time PLUS its gather time, T = 177 us + (gathers issued) / 165 per ns, whatever
// the number of waves per SIMD or of gathers in flight -- waves of one launch run the same program in phase, so the memory
// system is idle while they compute and saturated while they wait.  Here a wave overlaps the two by itself: the gathers of
// batch b are issued, THEN the chain is advanced and the positions, voxel indices, addresses and free-space tests of batch
// b + 1 are formed (the largest share of the vector work) while they are in flight, then the hinge and the force / moment sums
// of batch b.  Two position / address buffers, one record buffer; U = 4 keeps it at two waves per SIMD.  Same arithmetic in
// the same order as loglik_config_regs: bit-identical results.  (Two batches of gathers in flight -- a second record buffer, the
// gathers of batch b + 1 issued before the hinge of batch b -- measured 276 against 266 us with the masks and 399 against 414 us
// with every sphere gathering: the pass sits at what the memory system serves in scattered 64-byte sectors, profiles/r04.)
// FAR: 0 every sphere reads the table; 1 brick summary (requested for batch b + 1 under the gathers of batch b: straight-line
// loads, so the compiler's wait for the gathers leaves them in flight); 2 free-space masks in LDS.
// SMALL (up to 7 joints: frames 0 .. 7; round 6): the per-frame sums as six 8-wide vectors and the joint loops to 8 -- 168 instead of 222-238
// registers (32-64 bytes of scratch), three waves per SIMD: the 7-joint arms at the speed of the retired prefix-scalar form (64 Franka
// problems 354 us per step; 378 with the 16-wide sums at two waves), the same arithmetic in the same order (bit-identical to the LDS form:
// tests/test_gpu_config5.py), 0 of 12 sessions of the process-mix reproducer.
typedef float vg_f32x8 __attribute__((ext_vector_type(8)));
template <int U, int FAR, bool SMALL = false, typename LoadRaw, typename ToAngle, typename Emit>
__device__ __forceinline__ float loglik_config_pipe(const vgpmp_robot* __restrict__ rb, const vg_sdf_dev& sdf,
                                                    const LikScratch sc, LoadRaw load_raw, ToAngle to_angle, Emit emit,
                                                    const uint32_t* lmask = nullptr) {
    static_assert(VGPMP_MAX_SPHERES % (2 * U) == 0, "two batches of sphere constants never leave the table");
    static_assert(FAR >= 0 && FAR <= 2, "none, summary or masks");
    const int D = rb->dof, P = rb->num_spheres;          // D <= 15: frames 0 .. 15 (SMALL: D <= 7)
    constexpr int kJ = SMALL ? 8 : VGPMP_MAX_DOF;
    using SumVec = typename std::conditional<SMALL, vg_f32x8, vg_f32x16>::type;
    float raw[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) raw[j] = load_raw(min(j, D - 1));
    const float eps = rb->epsilon;
    const double offx = rb->scene_offset[0], offy = rb->scene_offset[1], offz = rb->scene_offset[2];
    const SdfFast fs = make_fast(sdf, offx, offy, offz);
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
        if (j < D) {                                     // uniform
            float st, ct, d;
            vg_sincos(to_angle(j, raw[j], d) + rb->joint_tab[j][4], &st, &ct);
            sc.at(j) = st; sc.at(D + j) = ct; sc.at(2 * D + j) = d;
        }
    }
    SumVec fx = 0.f, fy = 0.f, fz = 0.f, mx = 0.f, my = 0.f, mz = 0.f;      // per-frame sums
    Frame T = base_frame(rb);
    int cur = 0;                                         // frame T stands at (issue side)
    int pcur = 0;                                        // frame of the running sums (consumer side)
    vg_float3 F = vg_make3(0.f, 0.f, 0.f), Mo = vg_make3(0.f, 0.f, 0.f);
    vg_float3 Ft = vg_make3(0.f, 0.f, 0.f), Mt = vg_make3(0.f, 0.f, 0.f);
    float acc = 0.f;
    auto flush = [&]() {                                 // sums of frame pcur are complete
        fx[pcur] = F.x; fy[pcur] = F.y; fz[pcur] = F.z; mx[pcur] = Mo.x; my[pcur] = Mo.y; mz[pcur] = Mo.z;
        Ft = vg_make3(Ft.x + F.x, Ft.y + F.y, Ft.z + F.z);
        Mt = vg_make3(Mt.x + Mo.x, Mt.y + Mo.y, Mt.z + Mo.z);
        F = vg_make3(0.f, 0.f, 0.f); Mo = vg_make3(0.f, 0.f, 0.f);
        ++pcur;
    };
    struct Batch {
        vg_float3 pos[U];
        uint32_t at[U];
        float sm[FAR ? U : 1];        // what the free-space test of issue() compares: FAR 1 the brick's smallest distance,
                                      // FAR 2 +inf where the sphere's block is marked free in its radius class's mask, else -inf
    };
    // stage 1: chain, positions, voxel addresses and the free-space test's operand of batch q0
    auto stage1 = [&](int q0, Batch& b) {
        float4 ca[U];
        float2 cb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ca[u] = *reinterpret_cast<const float4*>(rb->sphere_a[q0 + u]);      // {offset, frame}; rows >= P: frame = D
            cb[u] = *reinterpret_cast<const float2*>(rb->sphere_b[q0 + u]);      // {radius, 1 / sigma}
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u < P) {                            // uniform
                const int fr = __builtin_bit_cast(int, ca[u].w);
                while (cur < fr) {
                    dh_apply(rb, cur, sc.at(cur), sc.at(D + cur), T);
                    ++cur;
                }
                b.pos[u] = axpy(ca[u].x, T.cx, axpy(ca[u].y, T.cy, axpy(ca[u].z, T.cz, T.t)));
                const Vox3 ix = voxel3(b.pos[u], fs, sdf, offx, offy, offz);
                b.at[u] = (uint32_t)vg_table_offset(sdf, ix.ix, ix.iy, ix.iz);
                if (FAR == 1) b.sm[u] = sdf.brick_min[vg_brick_of(sdf, ix.ix, ix.iy, ix.iz)];
                if (FAR == 2) {
                    // the mask of the smallest clearance that covers this sphere (uniform): clearance >= eps + r in the hinge's
                    // own float32 form, so a set bit means cost exactly 0 on every voxel of the block (monotone rounding)
                    int mo = -1;
#pragma unroll
                    for (int k = VGPMP_MAX_MASKS - 1; k >= 0; --k)
                        if (k < sdf.mcount && !(eps - (sdf.mclr[k] - cb[u].x) > 0.f)) mo = k * sdf.mwords;
                    const uint32_t bit = (uint32_t)(((ix.ix >> sdf.mshift) * sdf.mby + (ix.iy >> sdf.mshift)) * sdf.mbz + (ix.iz >> sdf.mshift));
                    const uint32_t w = mo < 0 ? 0u : lmask[mo + (bit >> 5)] >> (bit & 31u);
                    b.sm[u] = (w & 1u) ? __builtin_inff() : -__builtin_inff();      // a distance that is / is not beyond every hinge
                }
            } else {
                b.pos[u] = vg_make3(0.f, 0.f, 0.f);
                b.at[u] = 0u;
                if (FAR) b.sm[u] = __builtin_inff();
            }
        }
    };
    // the gathers of a batch: spheres in free space keep a record whose hinge is exactly 0
    auto issue = [&](int q0, const Batch& b, float4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u < P) {                            // uniform
                if (FAR) {
                    const float r = rb->sphere_b[q0 + u][0];
                    v[u] = make_float4(FAR == 1 ? b.sm[u] : __builtin_inff(), 0.f, 0.f, 0.f);
                    if (eps - (b.sm[u] - r) > 0.f) v[u] = sdf.table[b.at[u]];
                } else {
                    v[u] = sdf.table[b.at[u]];
                }
            } else {
                v[u] = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);       // hinge exactly 0
            }
        }
    };
    // stage 3: hinge, log-density and the per-frame force / moment sums of batch q0
    auto stage3 = [&](int q0, const Batch& b, const float4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q < P) {                                 // uniform
                const float4 ca = *reinterpret_cast<const float4*>(rb->sphere_a[q]);
                const float2 cb = *reinterpret_cast<const float2*>(rb->sphere_b[q]);
                const int fr = __builtin_bit_cast(int, ca.w);
                while (pcur < fr) flush();
                const float c = fmaxf(eps - (v[u].x - cb.x), 0.f);               // likelihood.py:131-143
                const float cs = c * cb.y;
                acc = fmaf(cs, c, acc);                                          // likelihood.py:99
                const vg_float3 gp = vg_make3(cs * v[u].y, cs * v[u].z, cs * v[u].w);   // d logp / d pos
                F = vg_make3(F.x + gp.x, F.y + gp.y, F.z + gp.z);
                Mo = vg_cross_acc(Mo, b.pos[u], gp);
            }
        }
    };
    Batch A, B;
    float4 v[U];
    stage1(0, A);
#pragma nounroll
    for (int q0 = 0; q0 < P; q0 += 2 * U) {
        issue(q0, A, v);
        if (q0 + U < P) stage1(q0 + U, B);               // (under the gathers of A)
        stage3(q0, A, v);
        if (q0 + U < P) {
            issue(q0 + U, B, v);
            if (q0 + 2 * U < P) stage1(q0 + 2 * U, A);   // (under the gathers of B)
            stage3(q0 + U, B, v);
        }
    }
    while (pcur <= D) flush();
    // second sweep over the chain: joint i turns about z of frame i (Craig) or frame i-1 (classic) and moves every
    // sphere on frames >= i, i.e. the totals minus the prefix < i
    const bool craig = rb->craig != 0;
    T = base_frame(rb);
    vg_float3 Fs = Ft, Ms = Mt;
#pragma nounroll
    for (int i = 1; i <= D; ++i) {
        // the iteration's operands, all requested, then at rest (vg_sweep_fence)
        float4 jt = *reinterpret_cast<const float4*>(rb->joint_tab[i - 1]);
        float st = sc.at(i - 1), ct = sc.at(D + i - 1), dgdf = sc.at(2 * D + i - 1);
        vg_sweep_fence(jt, st, ct, dgdf);
        Fs = vg_make3(Fs.x - fx[i - 1], Fs.y - fy[i - 1], Fs.z - fz[i - 1]);
        Ms = vg_make3(Ms.x - mx[i - 1], Ms.y - my[i - 1], Ms.z - mz[i - 1]);
        vg_float3 z = T.cz, org = T.t;
        dh_apply_row(jt, craig, st, ct, T);
        if (craig) { z = T.cz; org = T.t; }
        const vg_float3 oxF = vg_cross(org, Fs);
        emit(i - 1, vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z)) * dgdf);
    }
    return -0.5f * acc;
}

// (Until round 6 batches of up to 8 joints ran a third one-lane form, loglik_config_prefix: the prefix term of every joint's gradient as a
//  scalar in LDS, formed by a second copy of the chain inside the forward loop -- three waves per SIMD, 4-7 % faster than the forms above
//  at 7 joints (a speed the 8-wide pipelined form, loglik_config_pipe<.., SMALL>, has since matched).  It was the ONE batch form whose results
//  differed beside another process's kernels (two same-seed planners parted ways in 31 of 38 reproducer sessions whatever was
//  fenced; every other form 0 of 36): the form richest in three-component arithmetic that the SLP vectoriser packs -- the instructions
//  that misbehave beside a wide f16 matrix instruction (see vg_sweep_fence above).  Retired before that was known; nothing to bring back.)

constexpr int kLikBlock = 128;
constexpr int kLikBatchBlock = 64;
constexpr int kLikBatchU = 8;           // sphere gathers in flight per lane in the batch form

// ---- stand-alone log_prob: g [n, dof] row major ------------------------------------------------
template <bool GRAD>
__global__ __launch_bounds__(kLikBlock) void log_prob_kernel(const vgpmp_robot* __restrict__ rb, vgpmp_sdf sdfh,
                                                              const float* __restrict__ gq, int64_t n,
                                                              float* __restrict__ logp, float* __restrict__ dlogp) {
    extern __shared__ float lik_lds[];
    const int64_t i = (int64_t)blockIdx.x * kLikBlock + threadIdx.x;
    if (i >= n) return;
    const vg_sdf_dev sdf = load_sdf(sdfh);
    const int D = rb->dof;
    const LikScratch sc{lik_lds + threadIdx.x, kLikBlock};
    const float* g = gq + i * D;
    float* dg = dlogp + i * D;
    logp[i] = loglik_config<GRAD, 8>(rb, sdf, sc, [&](int j) { return g[j]; }, [&](int, float x) { return x; },
                                     [&](int j, float v) { dg[j] = v; });
}

// ---- ELBO path: f [P,S,L,N] -> logp [P,S,N], G = dloss/df [P,S,L,N], block partial sums ----------
// LPC lanes per (sample, time) configuration; BLK / LPC configurations per workgroup.  Large batches run one-wave
// workgroups (kLikBatchBlock): 188 instead of 204 us per launch at 64 problems (finer tail).
template <int LPC, int BLK, bool SIG = false, bool FAR = false, bool REGS = false>
__global__ __launch_bounds__(BLK, REGS ? 2 : 1) void loglik_paths_kernel(const vgpmp_robot* __restrict__ rb, vgpmp_sdf sdfh,
                                                                  const float* __restrict__ f, int S, int L, int N,
                                                                  float scale, float* __restrict__ G,
                                                                  float* __restrict__ logp,
                                                                  float* __restrict__ lik_partial, int dbg,
                                                                  const float* __restrict__ alpha_eff = nullptr,
                                                                  const float* __restrict__ sigma_eff = nullptr,
                                                                  float* __restrict__ sig_partial = nullptr) {
    static_assert(LPC == 1, "one lane per configuration");
    static_assert(!SIG || BLK == VG_WAVE, "per-sphere sums of a workgroup are one wave's sums");
    extern __shared__ float lik_lds[];
    __shared__ float red[BLK / VG_WAVE];
#ifdef VGPMP_BISECT
    if (dbg == 1) return;
#endif
    constexpr int CPB = BLK / LPC;                 // configurations per workgroup
    const int pb = blockIdx.y;
    VG_T(blockIdx.x == 0 && pb == 0, 400);
    const int cl = threadIdx.x / LPC, sub = threadIdx.x % LPC;
    const int idx = blockIdx.x * CPB + cl;
    const bool live = idx < S * N;
    float lp = 0.f;
    {
        const vg_sdf_dev sdf = load_sdf(sdfh);
        const int ci = live ? idx : S * N - 1;           // dead groups recompute the last configuration, write nothing
        const int s = ci / N, n = ci - s * N;
        const size_t base = ((size_t)pb * S + s) * L * N + n;
        const LikScratch sc{lik_lds + cl, CPB};
        float* dgdf = lik_lds + (size_t)lik_scratch_slots(L) * CPB + cl;   // [L][CPB]
        const float scl = SIG ? -alpha_eff[pb] : scale;
        float* sp = SIG ? sig_partial + ((size_t)pb * gridDim.x + blockIdx.x) * VGPMP_MAX_SPHERES : nullptr;
        auto raw_f = [&](int j) { return f[base + (size_t)j * N]; };
        const float* sigp = SIG ? sigma_eff + (size_t)pb * VGPMP_MAX_SPHERES : nullptr;
        auto put_sig = [&](int q, float t) { if (threadIdx.x == 0) sp[q] = t; };
        if (REGS) {
            auto regs_form = [&](auto&&... a) { return loglik_config_regs<kLikBatchU, SIG, FAR>(a...); };
            lp = regs_form(
                rb, sdf, sc, raw_f,
                [&](int j, float x, float& d) {
                    const float sg = 1.0f / (1.0f + __expf(-x));                        // likelihood.py:49-52
                    const float span = rb->joint_tab[j][7];
                    d = span * sg * (1.0f - sg);
                    return fmaf(span, sg, rb->joint_tab[j][5]);
                },
                [&](int j, float v) { if (live) G[base + (size_t)j * N] = scl * v; }, sigp, live ? 1.f : 0.f, put_sig);
        } else {
            lp = loglik_config<true, kLikBatchU, SIG, FAR>(
                rb, sdf, sc, raw_f,
                [&](int j, float x) {
                    const float sg = 1.0f / (1.0f + __expf(-x));                        // likelihood.py:49-52
                    const float span = rb->joint_tab[j][7];
                    dgdf[j * CPB] = span * sg * (1.0f - sg);
                    return fmaf(span, sg, rb->joint_tab[j][5]);
                },
                [&](int j, float v) { if (live) G[base + (size_t)j * N] = scl * (v * dgdf[j * CPB]); }, sigp, live ? 1.f : 0.f,
                put_sig);
        }
        if (live && sub == 0) logp[((size_t)pb * S + s) * N + n] = lp;
    }
    float w = vg_wave_sum(live && sub == 0 ? lp : 0.f);
    if ((threadIdx.x & (VG_WAVE - 1)) == 0) red[threadIdx.x / VG_WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < BLK / VG_WAVE; ++k) t += red[k];
        lik_partial[(size_t)pb * gridDim.x + blockIdx.x] = t;
    }
    VG_T(blockIdx.x == 0 && pb == 0, 401);
    VG_T(blockIdx.x == gridDim.x - 1 && pb == 0, 405);
}

// ---- ELBO path, batch form of up to 15 joints: pipelined, free-space masks in LDS -----------------------------
// FARM = the pipelined form's FAR: 0 every sphere gathers, 1 brick summary, 2 free-space masks (the scene's default).
// Four waves per workgroup share ONE copy of the scene's free-space masks (include/vgpmp.h: a bit per block of voxels, a few KB
// to 32 KB) staged into LDS by DMA while the waves load their joint values; each wave then runs the register form on its own 64
// configurations exactly as a one-wave workgroup of loglik_paths_kernel would (same partial sums, one per wave), except that a
// sphere whose block is marked free costs an LDS read instead of a request to the memory system.  The 8 MiB brick summary of a
// 512^3 grid does not fit an XCD's L2 next to the table's lines: every query paid a scattered 4-byte load for it (36.9 M per launch
// at the config-5 share) before its 16-byte gather.  Results are bit-identical (skipped spheres cost exactly 0).
constexpr int kLikMaskBlock = 256;
constexpr int kLikPipeU = 4;      // spheres per batch of the pipelined form (two batches in registers; 8 spills 190 registers)
template <int FARM, bool SMALL = false>
__global__ __launch_bounds__(kLikMaskBlock, SMALL ? 3 : 2) void loglik_paths_mask_kernel(const vgpmp_robot* __restrict__ rb, vgpmp_sdf sdfh,
                                                                            const float* __restrict__ f, int S, int L, int N,
                                                                            float scale, float* __restrict__ G,
                                                                            float* __restrict__ logp,
                                                                            float* __restrict__ lik_partial, int nwaves) {
    extern __shared__ float lik_lds[];
    const vg_sdf_dev sdf = load_sdf(sdfh);
    const int mask_words = sdf.mcount * sdf.mwords;                 // a multiple of 4
    uint32_t* lmask = reinterpret_cast<uint32_t*>(lik_lds);
    float* scratch = lik_lds + mask_words;
    vg_stage_16(lmask, sdf.free_mask, mask_words >> 2, threadIdx.x, kLikMaskBlock);
    const int pb = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (VG_WAVE - 1);
    const int wv = blockIdx.x * (kLikMaskBlock / VG_WAVE) + wave;        // this wave's 64 configurations
    const int idx = wv * VG_WAVE + lane;
    const bool live = idx < S * N;
    const int ci = live ? idx : S * N - 1;           // dead lanes recompute the last configuration, write nothing
    const int s = ci / N, n = ci - s * N;
    const size_t base = ((size_t)pb * S + s) * L * N + n;
    const LikScratch sc{scratch + threadIdx.x, kLikMaskBlock};
    vg_dma_wait();
    __syncthreads();
    auto raw_f = [&](int j) { return f[base + (size_t)j * N]; };
    auto angle = [&](int j, float x, float& d) {
        const float sg = 1.0f / (1.0f + __expf(-x));                        // likelihood.py:49-52
        const float span = rb->joint_tab[j][7];
        d = span * sg * (1.0f - sg);
        return fmaf(span, sg, rb->joint_tab[j][5]);
    };
    auto put = [&](int j, float v) { if (live) G[base + (size_t)j * N] = scale * v; };
    const float lp = loglik_config_pipe<kLikPipeU, FARM, SMALL>(rb, sdf, sc, raw_f, angle, put, lmask);
    if (live) logp[((size_t)pb * S + s) * N + n] = lp;
    const float w = vg_wave_sum(live ? lp : 0.f);
    if (lane == 0 && wv < nwaves) lik_partial[(size_t)pb * nwaves + wv] = w;
}

// free-space masks from the brick summary: thread = block of 2^shift voxels per edge, bit = every brick of the block clears
__global__ __launch_bounds__(kBlock) void sdf_free_mask_kernel(vgpmp_sdf sdfh) {
    const vg_sdf_dev s = vg_load_sdf(sdfh);
    const int sh = s.mshift, e = 1 << (sh - 2);                         // bricks per block edge
    const int nbx = (s.nx + 3) >> 2;
    const int mbx = (s.nx + (1 << sh) - 1) >> sh;
    const size_t nbits = (size_t)mbx * s.mby * s.mbz;
    const size_t b = (size_t)blockIdx.x * kBlock + threadIdx.x;         // bit index; the grid covers mwords * 32 bits
    float m = __builtin_inff();
    if (b < nbits) {
        const int bz = (int)(b % s.mbz), by = (int)((b / s.mbz) % s.mby), bx = (int)(b / ((size_t)s.mbz * s.mby));
        for (int x = bx * e; x < min((bx + 1) * e, nbx); ++x)
            for (int y = by * e; y < min((by + 1) * e, s.nby); ++y)
                for (int z = bz * e; z < min((bz + 1) * e, s.nbz); ++z)
                    m = fminf(m, s.brick_min[((size_t)x * s.nby + y) * s.nbz + z]);
    }
    uint32_t* out = const_cast<uint32_t*>(s.free_mask);
    for (int k = 0; k < s.mcount; ++k) {
        const unsigned long long bal = __ballot(b < nbits && m >= s.mclr[k]);
        const int lane = threadIdx.x & (VG_WAVE - 1);
        // mwords is a multiple of 4 words, the grid of 8: the last workgroup's upper waves may lie beyond the mask
        const bool in_mask = (b >> 5) < (size_t)s.mwords;
        if (lane == 0 && in_mask) out[(size_t)k * s.mwords + (b >> 5)] = (uint32_t)bal;
        if (lane == 32 && in_mask) out[(size_t)k * s.mwords + (b >> 5)] = (uint32_t)(bal >> 32);
    }
}

// ---- ELBO path, few-problem form ---------------------------------------------------------------------
// LPC (4 or 8) adjacent lanes share one (sample, time) configuration.  What bounds this launch when few problems
// are in flight is the LENGTH of the dependent chain of one configuration, not throughput, so the chain is
// cut to: joint angles -> frames (all of them, kept in LDS) -> ONE round of voxel gathers (every sphere
// of a lane in flight together, 48 / LPC per lane) -> hinge -> per-frame force / moment sums -> one sweep
// back over the joints.  The robot table is copied to LDS first (its per-sphere rows are read per lane).
// Sums over a lane's spheres are kept per frame in lane-private LDS slots, written once each (spheres are
// sorted by frame) and combined across the four lanes by shuffles in a fixed order: deterministic.
static_assert(sizeof(vgpmp_robot) % 16 == 0, "the robot table is copied to LDS in 16-byte units");
constexpr int kWideSpheres = 48;   // spheres covered by ONE round of gathers (kWideSpheres / LPC per lane)

__device__ __forceinline__ int wide_group_slots(int D) { return 2 * D + 12 * (D + 1); }     // sin, cos, frames
__device__ __forceinline__ int wide_lane_slots(int D) { return 6 * (D + 1); }               // F, M per frame
static size_t wide_lds_bytes(int D, int lanes) {
    const int cpb = kLikBlock / lanes;
    return sizeof(vgpmp_robot) + ((size_t)(2 * D + 12 * (D + 1) + D) * cpb + (size_t)6 * (D + 1) * kLikBlock) * sizeof(float);
}

// sum of the SK split-K slabs of an LDS image [SK][n]: the fixed-order tree of gp_paths.h::sum_slabs_lds
template <int SK>
__device__ __forceinline__ float lik_sum_slabs(const float* raw, int e, int n) {
    float v[SK];
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = raw[k * n + e];
#pragma unroll
    for (int w = SK / 2; w > 0; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) v[k] += v[k + w];
    return v[0];
}
// words the path operands of the SK > 0 form need, overlaid on the per-lane force / moment slots (used only later)
static int wide_paths_words(int L, int SK) { (void)SK; return L * 32 + L * 16; }

// SK > 0 (eight lanes, Mz = 32, N a multiple of 4): the workgroup ASSEMBLES the paths of its sixteen configurations itself --
// one sample, sixteen consecutive time points (blockIdx.x = sample x ceil(N / 16) + tile) -- instead of reading f that a
// path-assembly launch wrote: r = U - f0(Z) - sqrt(jitter) eps' (U = m + C eps from stage B), f = f0(X) + r A^T by the MFMA
// sequence of paths_fwd_split_body on row 0 of the tile (same operands, same order: the same bits); tile 0 of a sample
// stores r for the reverse pass, every tile its f.  One launch and its hand-over fewer on the one-problem step.
// The chain as a prefix product over the 8 lanes of a group (up to 8 joints): lane j holds link j (sin / cos of its own joint)
// and ends with the product of links 0 .. j -- three rounds of (row shift, 3x4 product) instead of dof dependent products on
// every lane.  Shared with the tests' sphere-centre kernel below: the same code, the same roundings.
__device__ __forceinline__ Frame wide_scan_links(const vgpmp_robot* __restrict__ rb, int sub, int L, float st0, float ct0) {
    Frame Pm;
    if (sub < L) {
        Pm = dh_link(rb, sub, st0, ct0);
    } else {
        Pm.cx = vg_make3(1.f, 0.f, 0.f); Pm.cy = vg_make3(0.f, 1.f, 0.f); Pm.cz = vg_make3(0.f, 0.f, 1.f);
        Pm.t = vg_make3(0.f, 0.f, 0.f);
    }
    {
        const Frame Lf = frame_shr<1>(Pm);
        if (sub >= 1) Pm = frame_mul(Lf, Pm);
    }
    {
        const Frame Lf = frame_shr<2>(Pm);
        if (sub >= 2) Pm = frame_mul(Lf, Pm);
    }
    {
        const Frame Lf = frame_shr<4>(Pm);
        if (sub >= 4) Pm = frame_mul(Lf, Pm);
    }
    return Pm;
}

template <int LPC, bool SIG = false, int SK = 0>
__global__ __launch_bounds__(kLikBlock) void loglik_paths_wide_kernel(const vgpmp_robot* __restrict__ rb_g, vgpmp_sdf sdfh,
                                                                       const float* __restrict__ f, int S, int L, int N,
                                                                       float scale, float* __restrict__ G,
                                                                       float* __restrict__ logp,
                                                                       float* __restrict__ lik_partial,
                                                                       const float* __restrict__ alpha_eff = nullptr,
                                                                       const float* __restrict__ sigma_eff = nullptr,
                                                                       float* __restrict__ sig_partial = nullptr,
                                                                       vg_lik_paths lpa = vg_lik_paths{}) {
    constexpr bool PATHS = SK > 0;
    static_assert(!PATHS || LPC == 8, "the path-assembling form is the eight-lane form");
    extern __shared__ float lik_lds[];
    __shared__ float red[kLikBlock / VG_WAVE];
    constexpr int CPB = kLikBlock / LPC, kWideU = kWideSpheres / LPC;
    const int pb = blockIdx.y, tid = threadIdx.x;
    VG_T(blockIdx.x == 0 && pb == 0, 400);
    const vgpmp_robot* rb = reinterpret_cast<const vgpmp_robot*>(lik_lds);          // LDS copy
    vg_stage_16(lik_lds, rb_g, (int)(sizeof(vgpmp_robot) / 16), tid, kLikBlock);
    const int cl = tid / LPC, sub = tid % LPC;
    int s, n, n0 = 0, tile = 0;
    bool live;
    if (PATHS) {
        const int tps = (N + CPB - 1) / CPB;
        s = blockIdx.x / tps; tile = blockIdx.x - s * tps; n0 = tile * CPB;
        live = n0 + cl < N;
        n = min(n0 + cl, N - 1);                         // dead groups recompute the last time point, write nothing
    } else {
        const int idx = blockIdx.x * CPB + cl;
        live = idx < S * N;
        const int ci = live ? idx : S * N - 1;           // dead groups recompute the last configuration, write nothing
        s = ci / N; n = ci - s * N;
    }
    const size_t base = ((size_t)pb * S + s) * L * N + n;
    // path operands (PATHS), overlaid on the per-lane slots `mine` below
    float* ov = lik_lds + sizeof(vgpmp_robot) / sizeof(float) + (size_t)(wide_group_slots(L) + L) * CPB;
    float* rs = ov;                                      // [L][32]   r of the sample
    float* fasm = rs + L * 32;                           // [L][16]   the assembled f of the tile
    // PATHS: thread t < 16 L forms f of (latent t / 16, column t % 16) by ONE fmaf chain over the 32 inducing points -- the order
    // in which v_mfma_f32_16x16x4_f32 accumulates (k ascending: the same bits as paths_fwd_split_body's tiles) -- and loads its
    // column of A^T and its slab entries straight from memory (L2 resident); thread t < 32 L forms r[t] for everybody
    constexpr int kFT = 16 * 8 <= kLikBlock ? 1 : 2;     // columns per thread if 16 L exceeded the workgroup (it does not: L <= 8)
    static_assert(kFT == 1, "one (latent, column) per thread");
    float atc[32], f0c[SK > 0 ? SK : 1];
    float ru[2], re2[2], rz[2][SK > 0 ? SK : 1];
    const int fl = tid >> 4, fj = tid & 15, fjc = min(fj, N - 1 - n0);
    if (PATHS) {
        if constexpr (PATHS) {
            const int J = N + 32;
            const size_t sl = (size_t)pb * S + s;
            const int l = min(fl, L - 1);
            const float* atp = lpa.AT + ((size_t)(pb * L + l) * 32) * N + n0 + fjc;
#pragma unroll
            for (int k = 0; k < 32; ++k) atc[k] = atp[(size_t)k * N];
#pragma unroll
            for (int k = 0; k < SK; ++k) f0c[k] = lpa.F0[(size_t)k * lpa.slab + (sl * L + l) * J + n0 + fjc];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = min(tid + q * kLikBlock, L * 32 - 1), el = e >> 5, em = e & 31;
                ru[q] = lpa.U[(sl * L + el) * 32 + em];
                re2[q] = lpa.eps2[(sl * 32 + em) * L + el];
#pragma unroll
                for (int k = 0; k < SK; ++k) rz[q][k] = lpa.F0[(size_t)k * lpa.slab + (sl * L + el) * J + N + em];
            }
        }
    }
    // this lane's joints: sub, sub + 4, ... (at most 4 of them)
    float fv[VGPMP_MAX_DOF / LPC];
    float st0 = 0.f, ct0 = 1.f, dg0 = 0.f;               // joint `sub`: sin, cos, d g / d f
    if (!PATHS) {
#pragma unroll
        for (int k = 0; k < VGPMP_MAX_DOF / LPC; ++k) fv[k] = f[base + (size_t)min(sub + LPC * k, L - 1) * N];
    }
    static_assert(VGPMP_MAX_DOF % LPC == 0, "joints are dealt to the lanes of a group");
    // trainable sigma_obs: this problem's variances replace the table's (requested before the wait, stored after it)
    float sig_own = 0.f;
    if (SIG && tid < VGPMP_MAX_SPHERES) sig_own = sigma_eff[(size_t)pb * VGPMP_MAX_SPHERES + tid];
    const float scl = SIG ? -alpha_eff[pb] : scale;
    vg_dma_wait();
    __syncthreads();
    if (SIG) {
        if (tid < VGPMP_MAX_SPHERES) {
            const_cast<vgpmp_robot*>(rb)->sigma_obs[tid] = sig_own;
            const_cast<vgpmp_robot*>(rb)->inv_sigma_obs[tid] = 1.0f / sig_own;
        }
        __syncthreads();
    }
    if (PATHS) {
        if constexpr (PATHS) {
            const size_t sl = (size_t)pb * S + s;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + q * kLikBlock;
                if (e < L * 32) {
                    float v[SK];
#pragma unroll
                    for (int k = 0; k < SK; ++k) v[k] = rz[q][k];
#pragma unroll
                    for (int w = SK / 2; w > 0; w >>= 1)      // the fixed-order tree of gp_paths.h::sum_slabs_lds
#pragma unroll
                        for (int k = 0; k < w; ++k) v[k] += v[k + w];
                    const float r = vg_path_r(ru[q], v[0], lpa.sqrt_jitter, re2[q]);
                    rs[e] = r;
                    if (tile == 0) vg_stream(lpa.R + sl * L * 32 + e, r);
                }
            }
            __syncthreads();
            if (fl < L) {
                float v[SK];
#pragma unroll
                for (int k = 0; k < SK; ++k) v[k] = f0c[k];
#pragma unroll
                for (int w = SK / 2; w > 0; w >>= 1)
#pragma unroll
                    for (int k = 0; k < w; ++k) v[k] += v[k + w];
                float acc = v[0];
                const float* rp = rs + fl * 32;
#pragma unroll
                for (int k = 0; k < 32; ++k) acc = fmaf(rp[k], atc[k], acc);
                fasm[fl * 16 + fj] = acc;
                if (n0 + fj < N) vg_stream(lpa.f + (sl * L + fl) * N + n0 + fj, acc);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VGPMP_MAX_DOF / LPC; ++k) fv[k] = fasm[min(sub + LPC * k, L - 1) * 16 + cl];
            __syncthreads();                                 // the overlay is dead from here: `mine` may be written
        }
    }
    VG_T(blockIdx.x == 0 && pb == 0, 402);
    const int D = rb->dof, P = rb->num_spheres;
    float* grp = lik_lds + sizeof(vgpmp_robot) / sizeof(float) + cl;                 // [slot][CPB]
    float* dgdf = lik_lds + sizeof(vgpmp_robot) / sizeof(float) + (size_t)wide_group_slots(D) * CPB + cl;   // [D][CPB]
    float* mine = lik_lds + sizeof(vgpmp_robot) / sizeof(float) + (size_t)(wide_group_slots(D) + D) * CPB + tid;   // [slot][block]
    // SIG: c^2 / sigma of every (configuration, sphere) of the workgroup, [CPB][MAX_SPHERES], summed in fixed order below
    float* sgl = lik_lds + sizeof(vgpmp_robot) / sizeof(float) + (size_t)(wide_group_slots(D) + D) * CPB +
                 (size_t)wide_lane_slots(D) * kLikBlock;
    auto gs = [&](int slot) -> float& { return grp[slot * CPB]; };
    auto ms = [&](int slot) -> float& { return mine[slot * kLikBlock]; };
    // ---- joint angles: sigmoid to the limits (likelihood.py:49-52), sin / cos
#pragma unroll
    for (int k = 0; k < VGPMP_MAX_DOF / LPC; ++k) {
        const int j = sub + LPC * k;
        if (j < D) {
            const float sg = 1.0f / (1.0f + __expf(-fv[k]));
            const float span = rb->high[j] - rb->low[j];
            dgdf[j * CPB] = span * sg * (1.0f - sg);
            float st, ct;
            vg_sincos(fmaf(span, sg, rb->low[j]) + rb->twist[j], &st, &ct);
            gs(j) = st; gs(D + j) = ct;
            if (k == 0) { st0 = st; ct0 = ct; dg0 = span * sg * (1.0f - sg); }
        }
    }
    for (int k = 0; k < wide_lane_slots(D); ++k) ms(k) = 0.f;
    auto put_frame = [&](int i, const Frame& T) {
        const int o = 2 * D + 12 * i;
        gs(o) = T.cx.x; gs(o + 1) = T.cx.y; gs(o + 2) = T.cx.z;
        gs(o + 3) = T.cy.x; gs(o + 4) = T.cy.y; gs(o + 5) = T.cy.z;
        gs(o + 6) = T.cz.x; gs(o + 7) = T.cz.y; gs(o + 8) = T.cz.z;
        gs(o + 9) = T.t.x; gs(o + 10) = T.t.y; gs(o + 11) = T.t.z;
    };
    // the chain as a prefix product over the lanes of the group: lane j owns joint j, three rounds of
    // (row shift, 3x4 product) instead of dof dependent products on every lane
    const bool scan = LPC == 8 && L <= LPC;
    if (scan) {
        const Frame Pm = wide_scan_links(rb, sub, L, st0, ct0);
        const Frame B = base_frame(rb);
        if (sub < L) put_frame(sub + 1, frame_mul(B, Pm));
        if (sub == (L < LPC ? L : 0)) put_frame(0, B);
    }
    lik_wave_sync();
    // ---- every frame of the chain, frame i stored by lane i % LPC
    if (!scan) {
        Frame T = base_frame(rb);
        for (int i = 0; i <= D; ++i) {
            if (i > 0) dh_apply(rb, i - 1, gs(i - 1), gs(D + i - 1), T);
            if (i % LPC == sub) {
                const int o = 2 * D + 12 * i;
                gs(o) = T.cx.x; gs(o + 1) = T.cx.y; gs(o + 2) = T.cx.z;
                gs(o + 3) = T.cy.x; gs(o + 4) = T.cy.y; gs(o + 5) = T.cy.z;
                gs(o + 6) = T.cz.x; gs(o + 7) = T.cz.y; gs(o + 8) = T.cz.z;
                gs(o + 9) = T.t.x; gs(o + 10) = T.t.y; gs(o + 11) = T.t.z;
            }
        }
    }
    lik_wave_sync();
    VG_T(blockIdx.x == 0 && pb == 0, 403);
    // ---- spheres sub, sub + 4, ...: positions, voxel gathers (all in flight), hinge, per-frame sums
    const vg_sdf_dev sdf = load_sdf(sdfh);
    const float eps = rb->epsilon;
    const double offx = rb->scene_offset[0], offy = rb->scene_offset[1], offz = rb->scene_offset[2];
    const SdfFast fs = make_fast(sdf, offx, offy, offz);
    float acc = 0.f;
    int cur = 0;                                         // frame of the running sums
    vg_float3 F = vg_make3(0.f, 0.f, 0.f), Mo = vg_make3(0.f, 0.f, 0.f);
    auto flush = [&]() {
        const int o = 6 * cur;
        ms(o) = F.x; ms(o + 1) = F.y; ms(o + 2) = F.z; ms(o + 3) = Mo.x; ms(o + 4) = Mo.y; ms(o + 5) = Mo.z;
    };
    const int nk = (P + LPC - 1) / LPC;
    for (int k0 = 0; k0 < nk; k0 += kWideU) {
        float4 v[kWideU];
        vg_float3 pos[kWideU];
        int fr[kWideU];
#pragma unroll
        for (int u = 0; u < kWideU; ++u) {               // tail lanes repeat the last sphere (weight 0)
            const int q = min(sub + LPC * (k0 + u), P - 1);
            fr[u] = rb->sphere_frame[q];
            const int o = 2 * D + 12 * fr[u];
            const vg_float3 cx = vg_make3(gs(o), gs(o + 1), gs(o + 2)), cy = vg_make3(gs(o + 3), gs(o + 4), gs(o + 5));
            const vg_float3 cz = vg_make3(gs(o + 6), gs(o + 7), gs(o + 8)), t = vg_make3(gs(o + 9), gs(o + 10), gs(o + 11));
            pos[u] = axpy(rb->sphere_off[q][0], cx, axpy(rb->sphere_off[q][1], cy, axpy(rb->sphere_off[q][2], cz, t)));
            const Vox3 vx = voxel3(pos[u], fs, sdf, offx, offy, offz);
            v[u] = sdf.table[vg_table_offset(sdf, vx.ix, vx.iy, vx.iz)];
        }
#pragma unroll
        for (int u = 0; u < kWideU; ++u) {
            const int qq = sub + LPC * (k0 + u);
            const int q = min(qq, P - 1);
            const float wgt = qq < P ? 1.f : 0.f;
            const float c = fmaxf(eps - (v[u].x - rb->radius[q]), 0.f) * wgt;       // likelihood.py:131-143
            const float cs = c * rb->inv_sigma_obs[q];
            acc = fmaf(cs, c, acc);                                                  // likelihood.py:99
            if (SIG && qq < P) sgl[cl * VGPMP_MAX_SPHERES + q] = live ? cs * c : 0.f;
            const vg_float3 gp = vg_make3(cs * v[u].y, cs * v[u].z, cs * v[u].w);   // d logp / d pos
            if (fr[u] != cur) {                          // spheres are sorted by frame: each slot is written once
                flush();
                cur = fr[u];
                F = vg_make3(0.f, 0.f, 0.f); Mo = vg_make3(0.f, 0.f, 0.f);
            }
            F = vg_make3(F.x + gp.x, F.y + gp.y, F.z + gp.z);
            Mo = vg_cross_acc(Mo, pos[u], gp);
        }
    }
    flush();
    VG_T(blockIdx.x == 0 && pb == 0, 404);
    acc = quad_sum<LPC>(acc);
    const float lp = -0.5f * acc;
    // ---- joints, last to first: joint i moves every sphere on frames >= i; it turns about z of frame i
    //      (Craig) or of frame i-1 (classic)
    if (scan) {
        // every lane forms all torques of ITS spheres (no lane crossing in the suffix sums), then one
        // transposing reduction leaves the total of joint j on lane j: 7 shuffles in 3 rounds
        const bool craig = rb->craig != 0;
        vg_float3 Fs = vg_make3(0.f, 0.f, 0.f), Ms = vg_make3(0.f, 0.f, 0.f);
        float tq[8];
#pragma unroll
        for (int i = 8; i >= 1; --i) {
            const bool on = i <= L;
            const int ii = on ? i : L;
            const int o = 6 * ii;
            const float w = on ? 1.f : 0.f;
            Fs = vg_make3(fmaf(w, ms(o), Fs.x), fmaf(w, ms(o + 1), Fs.y), fmaf(w, ms(o + 2), Fs.z));
            Ms = vg_make3(fmaf(w, ms(o + 3), Ms.x), fmaf(w, ms(o + 4), Ms.y), fmaf(w, ms(o + 5), Ms.z));
            const int fo = 2 * D + 12 * (craig ? ii : ii - 1);
            const vg_float3 z = vg_make3(gs(fo + 6), gs(fo + 7), gs(fo + 8)), org = vg_make3(gs(fo + 9), gs(fo + 10), gs(fo + 11));
            const vg_float3 oxF = vg_cross(org, Fs);
            tq[i - 1] = w * vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z));
        }
        float t4[4], t2[2];
        const bool h4 = (sub & 4) != 0, h2 = (sub & 2) != 0, h1 = (sub & 1) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] = (h4 ? tq[k + 4] : tq[k]) + __shfl_xor(h4 ? tq[k] : tq[k + 4], 4, VG_WAVE);
#pragma unroll
        for (int k = 0; k < 2; ++k) t2[k] = (h2 ? t4[k + 2] : t4[k]) + __shfl_xor(h2 ? t4[k] : t4[k + 2], 2, VG_WAVE);
        const float val = (h1 ? t2[1] : t2[0]) + __shfl_xor(h1 ? t2[0] : t2[1], 1, VG_WAVE);
        if (sub < L && live) vg_stream(G + base + (size_t)sub * N, scl * val * dg0);
    } else {
        const bool craig = rb->craig != 0;
        vg_float3 Fs = vg_make3(0.f, 0.f, 0.f), Ms = vg_make3(0.f, 0.f, 0.f);
        for (int i = D; i >= 1; --i) {
            // the torque is linear in (F, M): every lane forms it from the sums over ITS spheres, and only that one
            // number is added across the lanes (3 shuffles per joint instead of 18)
            const int o = 6 * i;
            Fs = vg_make3(Fs.x + ms(o), Fs.y + ms(o + 1), Fs.z + ms(o + 2));
            Ms = vg_make3(Ms.x + ms(o + 3), Ms.y + ms(o + 4), Ms.z + ms(o + 5));
            const int fo = 2 * D + 12 * (craig ? i : i - 1);
            const vg_float3 z = vg_make3(gs(fo + 6), gs(fo + 7), gs(fo + 8)), org = vg_make3(gs(fo + 9), gs(fo + 10), gs(fo + 11));
            const vg_float3 oxF = vg_cross(org, Fs);
            const float val = quad_sum<LPC>(vg_dot(z, vg_make3(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z)));
            if ((i - 1) % LPC == sub && live) vg_stream(G + base + (size_t)(i - 1) * N, scl * val * dgdf[(i - 1) * CPB]);
        }
    }
    if (live && sub == 0) logp[((size_t)pb * S + s) * N + n] = lp;
    float w = vg_wave_sum(live && sub == 0 ? lp : 0.f);
    if ((tid & (VG_WAVE - 1)) == 0) red[tid / VG_WAVE] = w;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kLikBlock / VG_WAVE; ++k) t += red[k];
        lik_partial[(size_t)pb * gridDim.x + blockIdx.x] = t;
    }
    if (SIG && tid < VGPMP_MAX_SPHERES) {      // per-sphere sum over the workgroup's configurations, fixed order
        float t = 0.f;
        if (tid < P)
            for (int c = 0; c < CPB; ++c) t += sgl[c * VGPMP_MAX_SPHERES + tid];
        sig_partial[((size_t)pb * gridDim.x + blockIdx.x) * VGPMP_MAX_SPHERES + tid] = t;
    }
    VG_T(blockIdx.x == 0 && pb == 0, 401);
    VG_T(blockIdx.x == gridDim.x - 1 && pb == 0, 405);
}

// ---- stand-alone FK: q [n, dof] -> pos [n, P, 3], frames [n, dof+1, 12] ---------------------------
__global__ __launch_bounds__(kBlock) void fk_spheres_kernel(const vgpmp_robot* __restrict__ rb,
                                                             const float* __restrict__ q, int64_t n,
                                                             float* __restrict__ pos, float* __restrict__ frames) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int D = rb->dof, P = rb->num_spheres;
    Frame T = base_frame(rb);
    int p = 0;
    for (int k = 0; k <= D; ++k) {
        if (k > 0) dh_step(rb, k - 1, q[i * D + k - 1], T);
        if (frames) {
            float* o = frames + (i * (D + 1) + k) * 12;
            o[0] = T.cx.x; o[1] = T.cy.x; o[2] = T.cz.x; o[3] = T.t.x;
            o[4] = T.cx.y; o[5] = T.cy.y; o[6] = T.cz.y; o[7] = T.t.y;
            o[8] = T.cx.z; o[9] = T.cy.z; o[10] = T.cz.z; o[11] = T.t.z;
        }
        while (p < P && rb->sphere_frame[p] == k) {
            vg_float3 x = axpy(rb->sphere_off[p][0], T.cx,
                               axpy(rb->sphere_off[p][1], T.cy, axpy(rb->sphere_off[p][2], T.cz, T.t)));
            if (pos) {
                float* o = pos + (i * P + p) * 3;
                o[0] = x.x; o[1] = x.y; o[2] = x.z;
            }
            ++p;
        }
    }
}

// ---- the sphere centres the ELBO kernels form from latent paths (parity tests: include/vgpmp_debug.h) ----------
// f [P, S, L, N] -> pos [P, S, N, Q, 3], by the arithmetic of the likelihood launch that vg_launch_loglik_paths selects for the
// same (P, S, N, L, form): joint sigmoid with __expf, vg_sincos, then either the serial chain of the one-lane forms
// (loglik_config*, dh_apply) or the 8-lane prefix product of loglik_paths_wide_kernel (wide_scan_links); the sphere offsets applied
// in the kernels' order.  The nearest-voxel lookup is piecewise constant: a float64 oracle looks its voxels up at THESE centres,
// so that no query of a comparison falls into a neighbouring cell.
template <bool SCAN>
__global__ __launch_bounds__(kLikBlock) void sphere_centres_paths_kernel(const vgpmp_robot* __restrict__ rb, const float* __restrict__ f,
                                                                        int S, int L, int N, float* __restrict__ pos) {
    constexpr int LPC = SCAN ? 8 : 1, CPB = kLikBlock / LPC;
    __shared__ float fr_lds[SCAN ? CPB * 12 * (8 + 1) : 1];
    const int pb = blockIdx.y, cl = threadIdx.x / LPC, sub = threadIdx.x % LPC;
    const int idx = blockIdx.x * CPB + cl;
    const bool live = idx < S * N;
    const int ci = live ? idx : S * N - 1;
    const int s = ci / N, n = ci - s * N;
    const size_t base = ((size_t)pb * S + s) * L * N + n;
    const int D = rb->dof, P = rb->num_spheres;
    float* out = pos + (((size_t)pb * S + s) * N + n) * P * 3;
    if (SCAN) {
        float st0 = 0.f, ct0 = 1.f;
        if (sub < D) {
            const float sg = 1.0f / (1.0f + __expf(-f[base + (size_t)sub * N]));
            const float span = rb->high[sub] - rb->low[sub];
            vg_sincos(fmaf(span, sg, rb->low[sub]) + rb->twist[sub], &st0, &ct0);
        }
        const Frame Pm = wide_scan_links(rb, sub, L, st0, ct0);
        const Frame B = base_frame(rb);
        float* grp = fr_lds + cl * 12 * 9;
        auto put = [&](int i, const Frame& T) {
            float* o = grp + 12 * i;
            o[0] = T.cx.x; o[1] = T.cx.y; o[2] = T.cx.z; o[3] = T.cy.x; o[4] = T.cy.y; o[5] = T.cy.z;
            o[6] = T.cz.x; o[7] = T.cz.y; o[8] = T.cz.z; o[9] = T.t.x; o[10] = T.t.y; o[11] = T.t.z;
        };
        if (sub < L) put(sub + 1, frame_mul(B, Pm));
        if (sub == (L < LPC ? L : 0)) put(0, B);
        __syncthreads();
        for (int q = sub; q < P; q += LPC) {
            const float* o = grp + 12 * rb->sphere_frame[q];
            const vg_float3 cx = vg_make3(o[0], o[1], o[2]), cy = vg_make3(o[3], o[4], o[5]);
            const vg_float3 cz = vg_make3(o[6], o[7], o[8]), t = vg_make3(o[9], o[10], o[11]);
            const vg_float3 x = axpy(rb->sphere_off[q][0], cx, axpy(rb->sphere_off[q][1], cy, axpy(rb->sphere_off[q][2], cz, t)));
            if (live) { out[3 * q] = x.x; out[3 * q + 1] = x.y; out[3 * q + 2] = x.z; }
        }
    } else {
        Frame T = base_frame(rb);
        int cur = 0;
        for (int q = 0; q < P; ++q) {
            const float4 ca = *reinterpret_cast<const float4*>(rb->sphere_a[q]);
            const int fr = __builtin_bit_cast(int, ca.w);
            while (cur < fr) {
                const float sg = 1.0f / (1.0f + __expf(-f[base + (size_t)cur * N]));
                float st, ct;
                vg_sincos(fmaf(rb->joint_tab[cur][7], sg, rb->joint_tab[cur][5]) + rb->joint_tab[cur][4], &st, &ct);
                dh_apply(rb, cur, st, ct, T);
                ++cur;
            }
            const vg_float3 x = axpy(ca.x, T.cx, axpy(ca.y, T.cy, axpy(ca.z, T.cz, T.t)));
            if (live) { out[3 * q] = x.x; out[3 * q + 1] = x.y; out[3 * q + 2] = x.z; }
        }
    }
}

// ---- stand-alone SDF query on float64 relative positions -----------------------------------------
__global__ __launch_bounds__(kBlock) void sdf_query_kernel(vgpmp_sdf sdfh, const double* __restrict__ rel, int64_t n,
                                                            int32_t* __restrict__ idx, float* __restrict__ dist,
                                                            float* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const vg_sdf_dev sdf = load_sdf(sdfh);
    int ix, iy, iz;
    size_t vi = vg_voxel_index(sdf, rel[3 * i], rel[3 * i + 1], rel[3 * i + 2], ix, iy, iz);
    float4 v = sdf.table[vi];
    if (idx) { idx[3 * i] = ix; idx[3 * i + 1] = iy; idx[3 * i + 2] = iz; }
    if (dist) dist[i] = v.x;
    if (grad) { grad[3 * i] = v.y; grad[3 * i + 1] = v.z; grad[3 * i + 2] = v.w; }
}

// the ELBO kernels' own index path (voxel3) on float32 positions in the robot frame: test entry vgpmp_sdf_index_float
__global__ __launch_bounds__(kBlock) void sdf_index_f32_kernel(vgpmp_sdf sdfh, double offx, double offy, double offz,
                                                                const float* __restrict__ pos, int64_t n, int32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const vg_sdf_dev sdf = load_sdf(sdfh);
    const SdfFast fs = make_fast(sdf, offx, offy, offz);
    const Vox3 v = voxel3(vg_make3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), fs, sdf, offx, offy, offz);
    idx[3 * i] = v.ix; idx[3 * i + 1] = v.iy; idx[3 * i + 2] = v.iz;
    // the eight-lane form's per-axis path must agree
    const int ax = voxel_axis(pos[3 * i], fs.chx, fs.clx, fs.inv_delta, sdf.nx, offx, sdf.ox, sdf.delta);
    const int ay = voxel_axis(pos[3 * i + 1], fs.chy, fs.cly, fs.inv_delta, sdf.ny, offy, sdf.oy, sdf.delta);
    const int az = voxel_axis(pos[3 * i + 2], fs.chz, fs.clz, fs.inv_delta, sdf.nz, offz, sdf.oz, sdf.delta);
    if (ax != v.ix || ay != v.iy || az != v.iz) idx[3 * i] = -1 - idx[3 * i];
}

// ---- voxel table: {d, gx, gy, gz} from float64 rows of the grid (utils/sdf_utils.py:100-136) -------
// One lane per table ELEMENT (so the 16-byte stores of a wave are 1 KiB contiguous under both layouts); under
// BRICK4 a wave is exactly one brick and also leaves the brick's smallest distance.
__global__ __launch_bounds__(kBlock) void sdf_pack_kernel(vgpmp_sdf sdfh, const double* __restrict__ rows, int row_lo,
                                                           int x0, int x1) {
    const vg_sdf_dev s = vg_load_sdf(sdfh);
    const int nx = s.nx, ny = s.ny, nz = s.nz;
    float4* __restrict__ table = const_cast<float4*>(s.table);
    auto at = [&](int x, int y, int z) { return rows[((size_t)(x - row_lo) * ny + y) * nz + z]; };
    auto record = [&](int ix, int iy, int iz) {
        const int xp = min(ix + 1, nx - 1), xm = max(ix - 1, 0);
        const int yp = min(iy + 1, ny - 1), ym = max(iy - 1, 0);
        const int zp = min(iz + 1, nz - 1), zm = max(iz - 1, 0);
        double gx = (at(xp, iy, iz) - at(xm, iy, iz)) / (2.0 * s.delta);
        double gy = (at(ix, yp, iz) - at(ix, ym, iz)) / (2.0 * s.delta);
        double gz = (at(ix, iy, zp) - at(ix, iy, zm)) / (2.0 * s.delta);
        gx = gx == 0.0 ? 0.1 : gx;
        gy = gy == 0.0 ? 0.1 : gy;
        gz = gz == 0.0 ? 0.1 : gz;
        return make_float4((float)at(ix, iy, iz), (float)gx, (float)gy, (float)gz);
    };
    if (s.layout == VGPMP_SDF_BRICK4) {
        const size_t bricks_per_x = (size_t)s.nby * s.nbz;
        const size_t b0 = (size_t)(x0 >> 2) * bricks_per_x, b1 = (size_t)((x1 + 3) >> 2) * bricks_per_x;
        const int lane = threadIdx.x & (VG_WAVE - 1);
        const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) / VG_WAVE, nwaves = (size_t)gridDim.x * kBlock / VG_WAVE;
        // lane -> local coordinates: the inverse of vg_morton_in_brick
        const int lz = (lane & 1) | ((lane >> 2) & 2), ly = ((lane >> 1) & 1) | ((lane >> 3) & 2), lx = ((lane >> 2) & 1) | ((lane >> 4) & 2);
        for (size_t b = b0 + wave; b < b1; b += nwaves) {
            const int bz = (int)(b % s.nbz), by = (int)((b / s.nbz) % s.nby), bx = (int)(b / bricks_per_x);
            const int ix = 4 * bx + lx, iy = 4 * by + ly, iz = 4 * bz + lz;
            const bool in = ix < nx && iy < ny && iz < nz;
            const float4 r = in ? record(ix, iy, iz) : make_float4(0.f, 0.f, 0.f, 0.f);    // padding is never indexed
            table[b * 64 + lane] = r;
            float m = in ? r.x : __builtin_inff();
#pragma unroll
            for (int o = VG_WAVE / 2; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, VG_WAVE));
            if (lane == 0 && s.brick_min) const_cast<float*>(s.brick_min)[b] = m;
        }
    } else {
        const size_t v0 = (size_t)x0 * ny * nz, v1 = (size_t)x1 * ny * nz;
        for (size_t v = v0 + (size_t)blockIdx.x * kBlock + threadIdx.x; v < v1; v += (size_t)gridDim.x * kBlock) {
            const int iz = (int)(v % nz);
            const int iy = (int)((v / nz) % ny);
            const int ix = (int)(v / ((size_t)nz * ny));
            table[v] = record(ix, iy, iz);
        }
    }
}

}  // namespace

// lanes per configuration: 8 while the launch is too small to fill the chip (bound by the length of one
// configuration's dependent chain), else 1
static int lik_lpc(int P, int S, int N) {
    const long long n = (long long)P * S * N;
    // measured on config 2 shapes (12 800 configurations per problem; tools/lpc_scan.sh): 8 lanes 10.8 / 20.6 / 26.5 us at
    // 1 / 2 / 3 problems, the batch form 22.2 / 23.0 / 23.2 us
    return n <= 28672 ? 8 : 1;
}
int vg_loglik_blocks_per_problem(int S, int N) { return S * ((N * 8 + kLikBlock - 1) / kLikBlock); }   // upper bound (one sample per workgroup row in the path-assembling form)

static size_t lik_lds_bytes(int dof, bool with_dgdf) {
    return (size_t)(2 * dof + 6 * (dof + 1) + (with_dgdf ? dof : 0)) * kLikBlock * sizeof(float);
}

int vg_launch_sdf_pack(const vgpmp_sdf* sdf, const double* rows, int row_lo, int row_hi, int x0, int x1, hipStream_t st) {
    (void)row_hi;
    const size_t elems = sdf->layout == VGPMP_SDF_BRICK4
        ? (size_t)((x1 + 3) / 4 - x0 / 4) * ((sdf->ny + 3) / 4) * ((sdf->nz + 3) / 4) * 64
        : (size_t)(x1 - x0) * sdf->ny * sdf->nz;
    unsigned blocks = (unsigned)((elems + kBlock - 1) / kBlock);
    if (blocks > 16384u) blocks = 16384u;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sdf_pack_kernel, dim3(blocks), dim3(kBlock), 0, st, *sdf, rows, row_lo, x0, x1);
    return (int)hipGetLastError();
}

int vg_launch_sdf_free_mask(const vgpmp_sdf* sdf, hipStream_t st) {
    const size_t bits = (size_t)sdf->mask_words * 32;
    hipLaunchKernelGGL(sdf_free_mask_kernel, dim3((unsigned)((bits + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, *sdf);
    return (int)hipGetLastError();
}

int vg_launch_fk_spheres(const vgpmp_robot* rb, const float* q, int64_t n, float* pos, float* frames, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fk_spheres_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rb, q, n,
                       pos, frames);
    return (int)hipGetLastError();
}

int vg_launch_sphere_centres(const vgpmp_robot* rb, const float* f, int P, int S, int L, int N, int form, float* pos, hipStream_t st) {
    if (P == 0 || S == 0 || N == 0) return 0;
    int lpc = lik_lpc(P, S, N);      // as vg_launch_loglik_paths
    if (form != 0) lpc = 1;
    const bool scan = lpc == 8 && L <= 8;
    const int cpb = kLikBlock / (scan ? 8 : 1);
    const dim3 grid((unsigned)((S * N + cpb - 1) / cpb), (unsigned)P);
    if (scan) hipLaunchKernelGGL((sphere_centres_paths_kernel<true>), grid, dim3(kLikBlock), 0, st, rb, f, S, L, N, pos);
    else hipLaunchKernelGGL((sphere_centres_paths_kernel<false>), grid, dim3(kLikBlock), 0, st, rb, f, S, L, N, pos);
    return (int)hipGetLastError();
}

int vg_launch_sdf_index_f32(const vgpmp_sdf* sdf, const double* offset, const float* pos, int64_t n, int32_t* idx, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sdf_index_f32_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, *sdf, offset[0],
                       offset[1], offset[2], pos, n, idx);
    return (int)hipGetLastError();
}

int vg_launch_sdf_query(const vgpmp_sdf* sdf, const double* rel, int64_t n, int32_t* idx, float* dist, float* grad,
                        hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sdf_query_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, *sdf, rel, n,
                       idx, dist, grad);
    return (int)hipGetLastError();
}

int vg_launch_log_prob_impl(const vgpmp_robot* rb, int dof, const vgpmp_sdf* sdf, const float* g, int64_t n,
                            float* logp, float* dlogp, hipStream_t st) {
    if (n == 0) return 0;
    dim3 grid((unsigned)((n + kLikBlock - 1) / kLikBlock)), block(kLikBlock);
    const size_t lds = lik_lds_bytes(dof, false);
    int rc = vg_grant_dyn_lds(dlogp ? (const void*)log_prob_kernel<true> : (const void*)log_prob_kernel<false>, lds);
    if (rc) return rc;
    if (dlogp) hipLaunchKernelGGL((log_prob_kernel<true>), grid, block, lds, st, rb, *sdf, g, n, logp, dlogp);
    else hipLaunchKernelGGL((log_prob_kernel<false>), grid, block, lds, st, rb, *sdf, g, n, logp, dlogp);
    return (int)hipGetLastError();
}

bool vg_lik_paths_fit(int L, int SK) { return L <= 8 && wide_paths_words(L, SK) <= 6 * (L + 1) * kLikBlock; }

#define VG_GO(fn_, ...) fn_(__VA_ARGS__, #__VA_ARGS__)
int vg_launch_loglik_paths(const vgpmp_robot* rb, const vgpmp_sdf* sdf, const float* f, int P, int S, int L, int N,
                           float scale, float* G, float* logp, float* lik_partial, int* nblk_out, hipStream_t st,
                           hipEvent_t k0, hipEvent_t k1, const float* alpha_eff, const float* sigma_eff,
                           float* sig_partial, int form, const vg_lik_paths* paths) {
    const bool sig = alpha_eff != nullptr;      // trainable likelihood constants: per-problem alpha / sigma_obs, per-sphere sums
    if (sig && (!sigma_eff || !sig_partial)) return VGPMP_E_ARG;
    int lpc = lik_lpc(P, S, N);
    // the brick summary (when the caller provides one) lets the batch forms leave out the table access of spheres in free space
    const bool far = sdf->layout == VGPMP_SDF_BRICK4 && sdf->brick_min != nullptr;
    if (form != 0) lpc = 1;                            // measurement: the batch form whatever the batch size
    int dbg = 0;
#ifdef VGPMP_BISECT
    dbg = lik_bisect_mode();
#endif
    const int blk = lpc > 1 ? kLikBlock : kLikBatchBlock;
    if (paths && (lpc != 8 || !vg_lik_paths_fit(L, paths->SK) || (N & 3))) return VGPMP_E_ARG;
    // the path-assembling form: one sample per row of ceil(N / 16) workgroups
    const int nblk = paths ? S * ((N + kLikBlock / 8 - 1) / (kLikBlock / 8)) : (S * N * lpc + blk - 1) / blk;
    if (nblk_out) *nblk_out = nblk;
    if (P == 0 || nblk == 0) return 0;
    size_t lds = lpc > 1 ? wide_lds_bytes(L, lpc) + (sig ? (size_t)(kLikBlock / lpc) * VGPMP_MAX_SPHERES * sizeof(float) : 0)
                               : lik_lds_bytes(L, true) * kLikBatchBlock / kLikBlock;
    // k0 / k1 (profiler): events stamped with the kernel's own start and end on the device
    if (lpc == 8) {
        const vg_lik_paths lp = paths ? *paths : vg_lik_paths{};
        auto go = [&](auto kern, const char* name) {
            int rc = vg_grant_dyn_lds((const void*)kern, lds);
            if (rc) return rc;
            vg_sched_note(name);
            hipExtLaunchKernelGGL(kern, dim3(nblk, P), dim3(kLikBlock), lds, st, k0, k1, 0, rb, *sdf, f, S, L, N, scale, G, logp,
                                  lik_partial, alpha_eff, sigma_eff, sig_partial, lp);
            return (int)hipGetLastError();
        };
        if (paths) {
            switch (paths->SK) {
                case 2: return sig ? VG_GO(go, loglik_paths_wide_kernel<8, true, 2>) : VG_GO(go, loglik_paths_wide_kernel<8, false, 2>);
                case 4: return sig ? VG_GO(go, loglik_paths_wide_kernel<8, true, 4>) : VG_GO(go, loglik_paths_wide_kernel<8, false, 4>);
                case 8: return sig ? VG_GO(go, loglik_paths_wide_kernel<8, true, 8>) : VG_GO(go, loglik_paths_wide_kernel<8, false, 8>);
                default: return VGPMP_E_ARG;
            }
        }
        return sig ? VG_GO(go, loglik_paths_wide_kernel<8, true, 0>) : VG_GO(go, loglik_paths_wide_kernel<8, false, 0>);
    }
    auto go = [&](auto kern, const char* name) {
        int rc = vg_grant_dyn_lds((const void*)kern, lds);
        if (rc) return rc;
        vg_sched_note(name);
        hipExtLaunchKernelGGL(kern, dim3(nblk, P), dim3(kLikBatchBlock), lds, st, k0, k1, 0, rb, *sdf, f, S, L, N, scale, G, logp,
                              lik_partial, dbg, alpha_eff, sigma_eff, sig_partial);
        return (int)hipGetLastError();
    };
    const bool regs = L <= 15 && form != 2;              // per-frame sums in registers (form 2, measurement: in LDS)
    if (regs) {
        // free-space masks in LDS (four-wave workgroups; the per-wave partial sums and their count stay those of the one-wave form)
        const size_t lds_mask = (sdf->free_mask ? (size_t)sdf->mask_count * sdf->mask_words * 4 : 0) + (size_t)3 * L * kLikMaskBlock * sizeof(float);
        const bool masks = sdf->layout == VGPMP_SDF_BRICK4 && sdf->free_mask && sdf->mask_count > 0;
        if (!sig && 2 * lds_mask <= 160 * 1024) {      // the pipelined form: every robot of up to 15 joints (round 6: the 7-joint arms as well)
            auto gom = [&](auto kern, const char* name) {
                int rc = vg_grant_dyn_lds((const void*)kern, lds_mask);
                if (rc) return rc;
                vg_sched_note(name);
                hipExtLaunchKernelGGL(kern, dim3((nblk + 3) / 4, P), dim3(kLikMaskBlock), lds_mask, st, k0, k1, 0, rb, *sdf, f, S, L, N,
                                      scale, G, logp, lik_partial, nblk);
                return (int)hipGetLastError();
            };
            // masks in LDS where the scene has them (one dependent global load per sphere), else the summary, else every sphere
            if (L <= 7)      // frames 0 .. 7: the 8-wide sums, three waves per SIMD (64 Franka problems: 378 -> 354 us per step)
                return masks ? VG_GO(gom, loglik_paths_mask_kernel<2, true>) : far ? VG_GO(gom, loglik_paths_mask_kernel<1, true>) : VG_GO(gom, loglik_paths_mask_kernel<0, true>);
            // (both template arguments spelled out: the schedule log carries the launch site's text, and a profiler prints `<2, false>`)
            return masks ? VG_GO(gom, loglik_paths_mask_kernel<2, false>) : far ? VG_GO(gom, loglik_paths_mask_kernel<1, false>) : VG_GO(gom, loglik_paths_mask_kernel<0, false>);
        }
        lds = (size_t)3 * L * kLikBatchBlock * sizeof(float);
        if (sig) return far ? VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, true, true, true>) : VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, true, false, true>);
        return far ? VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, false, true, true>) : VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, false, false, true>);
    }
    if (sig) return far ? VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, true, true, false>) : VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, true, false, false>);
    return far ? VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, false, true, false>) : VG_GO(go, loglik_paths_kernel<1, kLikBatchBlock, false, false, false>);
}
#undef VG_GO
