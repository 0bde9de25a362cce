// The DH chain of utils/sampler.py:103-120,142-214 as 3x4 rigid transforms in registers (shared by the likelihood kernels
// and the plan-extraction kernels).
#pragma once
#include "vgpmp_device.h"

// every fused multiply-add below is written as fmaf (the including files forbid implicit contraction)
struct Frame {
    vg_float3 cx, cy, cz, t;   // rotation columns and origin
};

__device__ __forceinline__ vg_float3 axpy(float a, vg_float3 x, vg_float3 y) {
    return vg_make3(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z));
}
__device__ __forceinline__ vg_float3 lin2(float a, vg_float3 x, float b, vg_float3 y) {
    return vg_make3(fmaf(a, x.x, b * y.x), fmaf(a, x.y, b * y.y), fmaf(a, x.z, b * y.z));
}

// T_i = T_{i-1} * A_i for joint j = i-1 (0-based table index), given sin/cos of theta_j + twist_j
// (jt = {cos alpha, sin alpha, d, a}: the first four entries of the joint's row of vgpmp_robot::joint_tab, wherever the caller keeps it)
__device__ __forceinline__ void dh_apply_row(const float4 jt, bool craig, float st, float ct, Frame& T) {
    const float ca = jt.x, sa = jt.y, d = jt.z, a = jt.w;
    if (craig) {
        // Rx(alpha) Tx(a) Rz(theta) Tz(d)      (utils/sampler.py:190-214)
        vg_float3 y1 = lin2(ca, T.cy, sa, T.cz);
        vg_float3 z1 = lin2(-sa, T.cy, ca, T.cz);
        T.t = axpy(a, T.cx, T.t);
        vg_float3 x2 = lin2(ct, T.cx, st, y1);
        vg_float3 y2 = lin2(-st, T.cx, ct, y1);
        T.cx = x2; T.cy = y2; T.cz = z1;
        T.t = axpy(d, z1, T.t);
    } else {
        // Rz(theta) Tz(d) Tx(a) Rx(alpha)      (utils/sampler.py:142-168)
        vg_float3 x1 = lin2(ct, T.cx, st, T.cy);
        vg_float3 y1 = lin2(-st, T.cx, ct, T.cy);
        T.t = axpy(d, T.cz, axpy(a, x1, T.t));
        vg_float3 y2 = lin2(ca, y1, sa, T.cz);
        vg_float3 z2 = lin2(-sa, y1, ca, T.cz);
        T.cx = x1; T.cy = y2; T.cz = z2;
    }
}

__device__ __forceinline__ void dh_apply(const vgpmp_robot* __restrict__ rb, int j, float st, float ct, Frame& T) {
    dh_apply_row(*reinterpret_cast<const float4*>(rb->joint_tab[j]), rb->craig != 0, st, ct, T);
}

__device__ __forceinline__ void dh_step(const vgpmp_robot* __restrict__ rb, int j, float theta, Frame& T) {
    float st, ct;
    sincosf(theta + rb->twist[j], &st, &ct);
    dh_apply(rb, j, st, ct, T);
}

__device__ __forceinline__ Frame base_frame(const vgpmp_robot* __restrict__ rb) {
    Frame T;
    T.cx = vg_make3(rb->base[0], rb->base[4], rb->base[8]);
    T.cy = vg_make3(rb->base[1], rb->base[5], rb->base[9]);
    T.cz = vg_make3(rb->base[2], rb->base[6], rb->base[10]);
    T.t = vg_make3(rb->base[3], rb->base[7], rb->base[11]);
    return T;
}

