// Measurement aid (profiles/r06/flake.md): a HAND-REDUCED copy of the loop whose results differed under preemption -- the second sweep of
// the likelihood's gradient as round 5 wrote it (no operand fence): per lane, seven joints; per joint six per-frame sums and sin / cos
// read from lane-strided LDS, the joint's DH row by a scalar load from a table in global memory, one step of the rigid chain, a cross
// and a dot product.  No library, no Python in this process: plain HIP.  The kernel repeats the sweep REP times per launch on operands it
// derives from the lane's index (deterministic), folds every gradient into a per-lane checksum, and the host compares every launch's
// checksums with the first launch's.  VARIANT 0: the loop as round 5 had it; 1: with the operand fence of round 6.
//   hipcc --offload-arch=gfx950 -O3 -DVARIANT=0 tools/sweep_probe.hip -o tools/sweep_probe && tools/sweep_probe [seconds] [rep] [workgroups]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef MODE
#define MODE 0      // 0 the sweep; 1 the sweep without LDS (operands in registers); 2 without the scalar load; 3 LDS reads only; 4 as 1 with explicit two-wide float vectors;
                    // 5 + packed subtraction; 6 + half-swapping shuffles; 7 as 1 with the DH row's scalar-register PAIRS as packed operands (two different halves)
#endif
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Robot {
    int dof, craig, pad0, pad1;
    float base[12];
    float joint_tab[16][8];      // {cos alpha, sin alpha, d, a, ...}
};
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
#if MODE >= 4
typedef float f2v __attribute__((ext_vector_type(2)));
#endif
#if MODE >= 4 && MODE <= 6
// MODE 4 (built WITHOUT the compiler's vectorisers): the x / y components of the chain's products as explicit two-wide float vectors --
// the compiler emits v_pk_mul_f32 / v_pk_fma_f32 for them whatever the vectorisers are told: packed instructions by construction
__device__ __forceinline__ f3 axpy(float a, f3 x, f3 y) {
    const f2v r = __builtin_elementwise_fma((f2v){a, a}, (f2v){x.x, x.y}, (f2v){y.x, y.y});
    return mk(r[0], r[1], fmaf(a, x.z, y.z));
}
__device__ __forceinline__ f3 lin2(float a, f3 x, float b, f3 y) {
    const f2v t = (f2v){b, b} * (f2v){y.x, y.y};
    const f2v r = __builtin_elementwise_fma((f2v){a, a}, (f2v){x.x, x.y}, t);
    return mk(r[0], r[1], fmaf(a, x.z, b * y.z));
}
#else
__device__ __forceinline__ f3 axpy(float a, f3 x, f3 y) { return mk(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z)); }
__device__ __forceinline__ f3 lin2(float a, f3 x, float b, f3 y) { return mk(fmaf(a, x.x, b * y.x), fmaf(a, x.y, b * y.y), fmaf(a, x.z, b * y.z)); }
#endif
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
struct Frame { f3 cx, cy, cz, t; };
__device__ __forceinline__ void dh_row(const float4 jt, bool craig, float st, float ct, Frame& T) {
    const float ca = jt.x, sa = jt.y, d = jt.z, a = jt.w;
    if (craig) {
        f3 y1 = lin2(ca, T.cy, sa, T.cz), z1 = lin2(-sa, T.cy, ca, T.cz);
        T.t = axpy(a, T.cx, T.t);
        f3 x2 = lin2(ct, T.cx, st, y1), y2 = lin2(-st, T.cx, ct, y1);
        T.cx = x2; T.cy = y2; T.cz = z1;
        T.t = axpy(d, z1, T.t);
    } else {
        f3 x1 = lin2(ct, T.cx, st, T.cy), y1 = lin2(-st, T.cx, ct, T.cy);
        T.t = axpy(d, T.cz, axpy(a, x1, T.t));
        f3 y2 = lin2(ca, y1, sa, T.cz), z2 = lin2(-sa, y1, ca, T.cz);
        T.cx = x1; T.cy = y2; T.cz = z2;
    }
}
__device__ __forceinline__ float hashf(unsigned x) {      // a float in (-1, 1) from an integer
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

#if MODE == 7 || MODE == 8
// MODE 7 (built WITHOUT the vectorisers): what the SLP build has and modes 4-6 lack -- packed instructions whose scalar-register operand is a
// PAIR with two different halves ({cos alpha, sin alpha}, {d, a}: v_pk_mul_f32 / v_pk_fma_f32 v, s[n:n+1], v ... with no op_sel_hi broadcast)
__device__ __forceinline__ void dh_row_pairs(const float4 jt, float st, float ct, Frame& T) {
    f2v cs = (f2v){jt.x, jt.y}, da = (f2v){jt.z, jt.w};
#if MODE == 8
    asm volatile("" : "+v"(cs), "+v"(da));      // MODE 8: the same arithmetic with the pairs copied to vector registers first
#endif
    f3 y1, z1, t1;
    {   const f2v p = cs * (f2v){T.cy.x, T.cz.x}, q = cs * (f2v){T.cz.x, T.cy.x}; y1.x = p[0] + p[1]; z1.x = q[0] - q[1]; }
    {   const f2v p = cs * (f2v){T.cy.y, T.cz.y}, q = cs * (f2v){T.cz.y, T.cy.y}; y1.y = p[0] + p[1]; z1.y = q[0] - q[1]; }
    {   const f2v p = cs * (f2v){T.cy.z, T.cz.z}, q = cs * (f2v){T.cz.z, T.cy.z}; y1.z = p[0] + p[1]; z1.z = q[0] - q[1]; }
    {   const f2v r = __builtin_elementwise_fma(da, (f2v){z1.x, T.cx.x}, (f2v){T.t.x, 0.0f}); t1.x = r[0] + r[1]; }
    {   const f2v r = __builtin_elementwise_fma(da, (f2v){z1.y, T.cx.y}, (f2v){T.t.y, 0.0f}); t1.y = r[0] + r[1]; }
    {   const f2v r = __builtin_elementwise_fma(da, (f2v){z1.z, T.cx.z}, (f2v){T.t.z, 0.0f}); t1.z = r[0] + r[1]; }
    const f3 x2 = lin2(ct, T.cx, st, y1), y2 = lin2(-st, T.cx, ct, y1);
    T.cx = x2; T.cy = y2; T.cz = z1; T.t = t1;
}
#endif

constexpr int kBlock = 128;
__global__ __launch_bounds__(kBlock) void sweep_kernel(const Robot* __restrict__ rb, int rep, unsigned* __restrict__ out) {
    extern __shared__ float lds[];
    const int D = rb->dof;
    float* sc = lds + threadIdx.x;      // slot s of this lane: sc[s * kBlock]; [0, D) sin, [D, 2 D) cos, then 6 per frame
    const unsigned gid = blockIdx.x * kBlock + threadIdx.x;
    f3 Ft = mk(0.f, 0.f, 0.f), Mt = mk(0.f, 0.f, 0.f);
    for (int j = 0; j < D; ++j) {
        const float ang = 3.0f * hashf(gid * 31u + (unsigned)j);
        sc[j * kBlock] = __sinf(ang); sc[(D + j) * kBlock] = __cosf(ang);
    }
    for (int k = 0; k <= D; ++k) {
        const int o = 2 * D + 6 * k;
        float v[6];
        for (int c = 0; c < 6; ++c) { v[c] = 10.0f * hashf(gid * 131u + (unsigned)(k * 6 + c) + 7u); sc[(o + c) * kBlock] = v[c]; }
        Ft = mk(Ft.x + v[0], Ft.y + v[1], Ft.z + v[2]);
        Mt = mk(Mt.x + v[3], Mt.y + v[4], Mt.z + v[5]);
    }
    const bool craig = rb->craig != 0;
    unsigned chk = 0u;
#if MODE == 3
    // LDS only: every repetition reads every slot of the lane back and folds the bits (no arithmetic on them)
#pragma nounroll
    for (int r = 0; r < rep; ++r) {
        sc[0] = sc[0] * 0.999f + 1e-3f * (float)(r & 7);
        for (int k = 0; k < 2 * D + 6 * (D + 1); ++k) chk = chk * 1664525u + __float_as_uint(sc[k * kBlock]) + (unsigned)k;
    }
#elif MODE == 1 || MODE >= 4
    // NO LDS in the loop: the lane's operands in registers (the sweep unrolled: seven joints)
    float rs[7], rc[7], rm[8][6];
#pragma unroll
    for (int j = 0; j < 7; ++j) { rs[j] = sc[j * kBlock]; rc[j] = sc[(7 + j) * kBlock]; }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) rm[k][c] = sc[(14 + 6 * k + c) * kBlock];
#pragma nounroll
    for (int r = 0; r < rep; ++r) {
        rs[0] = rs[0] * 0.999f + 1e-3f * (float)(r & 7);
        Frame T;
        T.cx = mk(rb->base[0], rb->base[4], rb->base[8]); T.cy = mk(rb->base[1], rb->base[5], rb->base[9]);
        T.cz = mk(rb->base[2], rb->base[6], rb->base[10]); T.t = mk(rb->base[3], rb->base[7], rb->base[11]);
        f3 Fs = Ft, Ms = Mt;
#pragma unroll
        for (int i = 1; i <= 7; ++i) {
#if MODE == 5 || MODE == 6
            {   // the x / y components of the prefix subtraction as two-wide vectors: v_pk_add_f32 with negated operands
                f2v fxy = (f2v){Fs.x, Fs.y} - (f2v){rm[i - 1][0], rm[i - 1][1]}, mxy = (f2v){Ms.x, Ms.y} - (f2v){rm[i - 1][3], rm[i - 1][4]};
#if MODE >= 6
                // ... and a half-swapping shuffle of two pairs (v_pk_mov_b32 with op_sel), undone again: the values do not change
                f2v sw = __builtin_shufflevector(fxy, mxy, 1, 2), sw2 = __builtin_shufflevector(fxy, mxy, 0, 3);
                asm volatile("" : "+v"(sw), "+v"(sw2));
                fxy = __builtin_shufflevector(sw2, sw, 0, 2); mxy = __builtin_shufflevector(sw, sw2, 1, 3);
#endif
                Fs = mk(fxy[0], fxy[1], Fs.z - rm[i - 1][2]);
                Ms = mk(mxy[0], mxy[1], Ms.z - rm[i - 1][5]);
            }
#else
            Fs = mk(Fs.x - rm[i - 1][0], Fs.y - rm[i - 1][1], Fs.z - rm[i - 1][2]);
            Ms = mk(Ms.x - rm[i - 1][3], Ms.y - rm[i - 1][4], Ms.z - rm[i - 1][5]);
#endif
            f3 z = T.cz, org = T.t;
#if MODE == 7 || MODE == 8
            dh_row_pairs(*reinterpret_cast<const float4*>(rb->joint_tab[i - 1]), rs[i - 1], rc[i - 1], T);
#else
            dh_row(*reinterpret_cast<const float4*>(rb->joint_tab[i - 1]), craig, rs[i - 1], rc[i - 1], T);
#endif
            if (craig) { z = T.cz; org = T.t; }
            const f3 oxF = cross(org, Fs);
            const float g = dot(z, mk(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z));
            chk = chk * 1664525u + __float_as_uint(g) + (unsigned)i;
        }
    }
#else
#pragma nounroll
    for (int r = 0; r < rep; ++r) {
        sc[0] = sc[0] * 0.999f + 1e-3f * (float)(r & 7);      // (the operands change with the repetition: nothing hoists out of this loop)
        Frame T;
        T.cx = mk(rb->base[0], rb->base[4], rb->base[8]); T.cy = mk(rb->base[1], rb->base[5], rb->base[9]);
        T.cz = mk(rb->base[2], rb->base[6], rb->base[10]); T.t = mk(rb->base[3], rb->base[7], rb->base[11]);
        f3 Fs = Ft, Ms = Mt;
#pragma nounroll
        for (int i = 1; i <= D; ++i) {
            const int o = 2 * D + 6 * (i - 1);
#if MODE == 2
            // no scalar load in the loop: the rows of the table out of constants (a Craig chain with alternating twists)
            const float4 jrow = make_float4((i & 1) ? 1.0f : 0.0f, (i & 1) ? 0.0f : ((i & 2) ? 1.0f : -1.0f), 0.1f * (float)i, 0.01f * (float)i);
#else
            const float4 jrow = *reinterpret_cast<const float4*>(rb->joint_tab[i - 1]);
#endif
#if VARIANT == 1
            float4 jt = jrow;
            float st = sc[(i - 1) * kBlock], ct = sc[(D + i - 1) * kBlock];
            float m0 = sc[o * kBlock], m1 = sc[(o + 1) * kBlock], m2 = sc[(o + 2) * kBlock], m3 = sc[(o + 3) * kBlock], m4 = sc[(o + 4) * kBlock], m5 = sc[(o + 5) * kBlock];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jt.x), "+s"(jt.y), "+s"(jt.z), "+s"(jt.w), "+v"(st), "+v"(ct), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5));
            Fs = mk(Fs.x - m0, Fs.y - m1, Fs.z - m2);
            Ms = mk(Ms.x - m3, Ms.y - m4, Ms.z - m5);
            f3 z = T.cz, org = T.t;
            dh_row(jt, craig, st, ct, T);
#else
            Fs = mk(Fs.x - sc[o * kBlock], Fs.y - sc[(o + 1) * kBlock], Fs.z - sc[(o + 2) * kBlock]);
            Ms = mk(Ms.x - sc[(o + 3) * kBlock], Ms.y - sc[(o + 4) * kBlock], Ms.z - sc[(o + 5) * kBlock]);
            f3 z = T.cz, org = T.t;
            dh_row(jrow, craig, sc[(i - 1) * kBlock], sc[(D + i - 1) * kBlock], T);
#endif
            if (craig) { z = T.cz; org = T.t; }
            const f3 oxF = cross(org, Fs);
            const float g = dot(z, mk(Ms.x - oxF.x, Ms.y - oxF.y, Ms.z - oxF.z));
            chk = chk * 1664525u + __float_as_uint(g) + (unsigned)i;
        }
    }
#endif
    out[gid] = chk;
}
__global__ void count_diff(const unsigned* __restrict__ a, const unsigned* __restrict__ b, size_t n, unsigned* bad, unsigned* first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) { if (atomicAdd(bad, 1u) == 0u) *first = (unsigned)i; }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 12.0;
    const int rep = argc > 2 ? atoi(argv[2]) : 40, wgs = argc > 3 ? atoi(argv[3]) : 4096;
    Robot h{};
    h.dof = 7; h.craig = 1;
    const float base[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; ++i) h.base[i] = base[i];
    const double alpha[7] = {0, -1.5708, 1.5708, 1.5708, -1.5708, 1.5708, 1.5708}, dd[7] = {0.333, 0, 0.316, 0, 0.384, 0, 0}, aa[7] = {0, 0, 0, 0.0825, -0.0825, 0, 0.088};
    for (int j = 0; j < 7; ++j) { h.joint_tab[j][0] = (float)cos(alpha[j]); h.joint_tab[j][1] = (float)sin(alpha[j]); h.joint_tab[j][2] = (float)dd[j]; h.joint_tab[j][3] = (float)aa[j]; }
    Robot* rb; unsigned *ref, *out, *bad, *first;
    const size_t n = (size_t)wgs * kBlock;
    CHECK(hipMalloc(&rb, sizeof(Robot))); CHECK(hipMemcpy(rb, &h, sizeof(Robot), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ref, n * 4)); CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&first, 4));
    const size_t lds = (size_t)(2 * 7 + 6 * 8) * kBlock * sizeof(float);
    CHECK(hipFuncSetAttribute((const void*)sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    // SWEEP_HSACO=<code object>: the kernel comes from that file instead (tools/depack_pk.py: the failing build's assembly with chosen
    // packed instructions rewritten as two plain ones, assembled with clang -x assembler + ld.lld) -- same symbol, same arguments
    hipFunction_t modf = nullptr;
    if (const char* path = getenv("SWEEP_HSACO")) {
        hipModule_t mod; CHECK(hipModuleLoad(&mod, path)); CHECK(hipModuleGetFunction(&modf, mod, "_Z12sweep_kernelPK5RobotiPj"));
        printf("kernel from %s\n", path);
    }
    auto launch = [&](unsigned* dst) -> hipError_t {
        if (!modf) { hipLaunchKernelGGL(sweep_kernel, dim3(wgs), dim3(kBlock), lds, st, rb, rep, dst); return hipGetLastError(); }
        struct { const Robot* rb; int rep; int pad; unsigned* out; } a{rb, rep, 0, dst};
        size_t sz = sizeof(a);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(modf, wgs, 1, 1, kBlock, 1, 1, (unsigned)lds, st, nullptr, cfg);
    };
    CHECK(launch(ref));
    {   // a checksum of the reference launch: equal across code objects that compute the same thing (the rewritten ones must)
        CHECK(hipStreamSynchronize(st));
        std::vector<unsigned> r0(n); CHECK(hipMemcpy(r0.data(), ref, n * 4, hipMemcpyDeviceToHost));
        unsigned long long hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { hsh ^= r0[i]; hsh *= 1099511628211ull; }
        printf("reference launch checksum %016llx\n", hsh);
    }
    CHECK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    unsigned launches = 0, events = 0;
    while (true) {
        ++launches;
        CHECK(hipMemsetAsync(out, 0xff, n * 4, st)); CHECK(hipMemsetAsync(bad, 0, 4, st));
        CHECK(launch(out));
        hipLaunchKernelGGL(count_diff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, ref, n, bad, first);
        unsigned hb = 0, hf = 0;
        CHECK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, st)); CHECK(hipMemcpyAsync(&hf, first, 4, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (hb) {
            ++events;
            std::vector<unsigned> o(n), r2(n);
            CHECK(hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), ref, n * 4, hipMemcpyDeviceToHost));
            printf("SWEEP DIFFERS launch %u (t = %.2f s): %u lanes; first lane %u; lanes:", launches, t, hb, hf);
            int shown = 0;
            for (size_t i = 0; i < n && shown < 24; ++i) if (o[i] != r2[i]) { printf(" %zu", i); ++shown; }
            printf("\n"); fflush(stdout);
        }
        if (t > seconds) break;
    }
    printf("sweep_probe (variant %d, mode %d): %u launches of %d repetitions on %d workgroups, %u launches differed from the first\n", VARIANT, MODE, launches, rep, wgs, events);
    return events ? 2 : 0;
}
