// Shared device helpers of the vGPMP HIP kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "vgpmp_debug.h"

#define VG_WAVE 64

#define VG_CHECK_HIP(expr)                      \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

struct vg_float3 {
    float x, y, z;
};

__device__ __forceinline__ vg_float3 vg_make3(float x, float y, float z) { return vg_float3{x, y, z}; }
// explicit fused forms: the rounding of these does not depend on what the optimiser decides to contract in a given
// instantiation (results must be bit-identical between kernel variants that differ only in where a voxel is fetched from)
__device__ __forceinline__ vg_float3 vg_cross(vg_float3 a, vg_float3 b) {
    return vg_float3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float vg_dot(vg_float3 a, vg_float3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// acc + a x b
__device__ __forceinline__ vg_float3 vg_cross_acc(vg_float3 acc, vg_float3 a, vg_float3 b) {
    return vg_float3{fmaf(a.y, b.z, fmaf(-a.z, b.y, acc.x)), fmaf(a.z, b.x, fmaf(-a.x, b.z, acc.y)),
                     fmaf(a.x, b.y, fmaf(-a.y, b.x, acc.z))};
}

__device__ __forceinline__ float vg_wave_sum(float v) {
#pragma unroll
    for (int o = VG_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, VG_WAVE);
    return v;
}
__device__ __forceinline__ double vg_wave_sum(double v) {
#pragma unroll
    for (int o = VG_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, VG_WAVE);
    return v;
}

// Pin a loaded value in a register at this point: clang otherwise sinks an unconditional load into the branch of the
// select that consumes it, turning "request everything, then choose" into one memory round trip per element.
__device__ __forceinline__ void vg_pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void vg_pin(float& v) { asm volatile("" : "+v"(v)); }

// Streaming store (nontemporal): bulk outputs that the NEXT launch reads -- from another XCD, so not through this L2
// anyway -- leave no dirty lines behind; the write-back of dirty L2 lines at the end of a kernel was measured to add
// ~1 us to the hand-over after a launch that wrote 7.6 MB.
typedef float vg_f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void vg_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void vg_stream(float4* p, float4 v) {
    __builtin_nontemporal_store((vg_f32x4_t){v.x, v.y, v.z, v.w}, reinterpret_cast<vg_f32x4_t*>(p));
}

// ---- global -> LDS staging without registers (global_load_lds, gfx950) ----------------------------
#define VG_DMA_AUX 0      // cache policy bits of the staging loads (plain)
// A rolled `lds[e] = g[e]` loop compiles to load / wait / store per iteration: one memory round trip per
// 256 elements (measured 10 us for the 74 KB of the reverse pass; 3 us with the form below).  Here every
// request of the workgroup is issued back to back and lands in LDS by itself; ONE vg_dma_wait() + barrier
// before the first read.  Hardware rule: a wave instruction writes lane k at wave_base + k * size, so the
// LDS image is linear in the lane index; the GLOBAL address is per lane (gathers, padding and transposes
// go on the source side).
typedef __attribute__((address_space(1))) const void vg_gmem;
typedef __attribute__((address_space(3))) void vg_lmem;
__device__ __forceinline__ void vg_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0)

// LDS words [0, nwords): word i comes from the 4-byte global address map(i); map(i) == nullptr stores 0.
template <typename Map>
__device__ __forceinline__ void vg_stage_words(void* lds, int nwords, int tid, int nt, Map map) {
    uint32_t* w = reinterpret_cast<uint32_t*>(lds);
    const int lane = tid & (VG_WAVE - 1);
    for (int c = (tid & ~(VG_WAVE - 1)); c < nwords; c += nt) {
        const int i = c + lane;
        if (i < nwords) {
            const void* g = map(i);
            if (g) __builtin_amdgcn_global_load_lds((vg_gmem*)g, (vg_lmem*)(w + c), 4, 0, VG_DMA_AUX);
            else w[i] = 0u;
        }
    }
}
// LDS image [nrows][row_floats] (linear); row r comes from the global row row_src(r), nullptr = a row of zeros.
// Rows whose length is a multiple of 4 floats move in 16-byte units (sources and `lds` must then be 16-byte
// aligned -- true for every tensor here whose inner extents are multiples of 4); other lengths fall back to words.
template <typename Map>
__device__ __forceinline__ void vg_stage_rows(void* lds, int nrows, int row_floats, int tid, int nt, Map row_src) {
    const int lane = tid & (VG_WAVE - 1);
    if ((row_floats & 3) == 0) {
        const int upr = row_floats >> 2, total = nrows * upr;
        const float iupr = 1.0f / (float)upr;
        for (int c = (tid & ~(VG_WAVE - 1)); c < total; c += nt) {
            const int i = c + lane;
            if (i < total) {
                const int r = (int)(((float)i + 0.5f) * iupr), u = i - r * upr;
                const float* g = row_src(r);
                if (g) __builtin_amdgcn_global_load_lds((vg_gmem*)(g + 4 * u), (vg_lmem*)((char*)lds + 16 * (size_t)c), 16, 0, VG_DMA_AUX);
                else reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    } else {
        const float irf = 1.0f / (float)row_floats;
        vg_stage_words(lds, nrows * row_floats, tid, nt, [&](int i) -> const void* {
            const int r = (int)(((float)i + 0.5f) * irf);
            const float* g = row_src(r);
            return g ? g + (i - r * row_floats) : nullptr;
        });
    }
}
// Float64 matrix [rows x cols] (row-major, dense) -> LDS image with row stride `ld` doubles, zero outside
// [0, rows) x [0, cols) up to `prows` rows; `row0`/`col0` shift the matrix inside the image (padding in front).
// keep(r, c) == false leaves a zero (triangular masks).  Two 4-byte requests per double.
template <typename Keep>
__device__ __forceinline__ void vg_stage_f64(double* lds, int prows, int ld, const double* g, int rows, int cols,
                                             int row0, int col0, int tid, int nt, Keep keep) {
    const float ild = 1.0f / (float)ld;
    vg_stage_words(lds, 2 * prows * ld, tid, nt, [&](int w) -> const void* {
        const int d = w >> 1, r = (int)(((float)d + 0.5f) * ild), c = d - r * ld;
        const int gr = r - row0, gc = c - col0;
        if (gr < 0 || gc < 0 || gr >= rows || gc >= cols || !keep(gr, gc)) return nullptr;
        return reinterpret_cast<const uint32_t*>(g + (size_t)gr * cols + gc) + (w & 1);
    });
}
// Dense float64 n x n matrix (n even, 16-byte aligned) -> LDS rows of `ld` doubles (ld even, >= n), in 16-byte
// units; the pad units of a row repeat its last valid unit (never read by the consumers, no out-of-range access).
__device__ __forceinline__ void vg_stage_f64_square(double* lds, int ld, const double* g, int n, int tid, int nt) {
    const int lane = tid & (VG_WAVE - 1), upr = ld >> 1, total = n * upr, last = (n >> 1) - 1;
    const float iupr = 1.0f / (float)upr;
    for (int c = (tid & ~(VG_WAVE - 1)); c < total; c += nt) {
        const int i = c + lane;
        if (i < total) {
            const int r = (int)(((float)i + 0.5f) * iupr), u = min(i - r * upr, last);
            __builtin_amdgcn_global_load_lds((vg_gmem*)(g + (size_t)r * n + 2 * u), (vg_lmem*)((char*)lds + 16 * (size_t)c), 16, 0, VG_DMA_AUX);
        }
    }
}
// Dense float64 rows x cols matrix (cols even, 16-byte aligned) -> the [prows][ld] LDS image (ld even), ZERO outside the matrix:
// units inside come by DMA in 16-byte units, the others are zeros written directly (each lane writes only its own unit of the
// linear image, so nothing races with the DMA).  The word-granular vg_stage_f64 moves the same bytes in four times the
// requests -- and a CU's staging goes by the number of wave requests (lesson 41): at Mz = 26 it was most of stage B's time.
__device__ __forceinline__ void vg_stage_f64_even(double* lds, int prows, int ld, const double* g, int rows, int cols, int tid, int nt) {
    const int lane = tid & (VG_WAVE - 1), upr = ld >> 1, total = prows * upr, cu = cols >> 1;
    const float iupr = 1.0f / (float)upr;
    for (int c = (tid & ~(VG_WAVE - 1)); c < total; c += nt) {
        const int i = c + lane;
        if (i < total) {
            const int r = (int)(((float)i + 0.5f) * iupr), u = i - r * upr;
            if (r < rows && u < cu)
                __builtin_amdgcn_global_load_lds((vg_gmem*)(g + (size_t)r * cols + 2 * u), (vg_lmem*)((char*)lds + 16 * (size_t)c), 16, 0, VG_DMA_AUX);
            else
                reinterpret_cast<double2*>(lds)[i] = make_double2(0.0, 0.0);
        }
    }
}
// contiguous copy of n16 16-byte units (both sides 16-byte aligned)
__device__ __forceinline__ void vg_stage_16(void* lds, const void* g, int n16, int tid, int nt) {
    const int lane = tid & (VG_WAVE - 1);
    for (int c = (tid & ~(VG_WAVE - 1)); c < n16; c += nt)
        if (c + lane < n16)
            __builtin_amdgcn_global_load_lds((vg_gmem*)((const char*)g + 16 * (size_t)(c + lane)),
                                             (vg_lmem*)((char*)lds + 16 * (size_t)c), 16, 0, VG_DMA_AUX);
}

// ---- nearest-voxel signed distance lookup (utils/sdf_utils.py:62-66,73-76) ------------------
// Index arithmetic in float64 in the reference's operation order ((p - offset) - origin) / delta,
// truncate, clamp, so that indices are bit-identical to the float64 reference on equal inputs.
struct vg_sdf_dev {
    const float4* table;
    const float* brick_min;     // nullptr: no summary
    int nx, ny, nz, layout;
    int nby, nbz;               // bricks along y and z (BRICK4)
    double ox, oy, oz, delta;
    // free-space masks (include/vgpmp.h): block edge 2^mshift voxels, mby x mbz blocks along y, z
    const uint32_t* free_mask;
    int mshift, mcount, mwords, mby, mbz;
    float mclr[VGPMP_MAX_MASKS];
};

__device__ __forceinline__ vg_sdf_dev vg_load_sdf(const vgpmp_sdf& s) {
    vg_sdf_dev d;
    d.table = reinterpret_cast<const float4*>(s.table);
    d.brick_min = reinterpret_cast<const float*>(s.brick_min);
    d.nx = s.nx; d.ny = s.ny; d.nz = s.nz; d.layout = s.layout;
    d.nby = (s.ny + 3) >> 2; d.nbz = (s.nz + 3) >> 2;
    d.ox = s.origin[0]; d.oy = s.origin[1]; d.oz = s.origin[2];
    d.delta = s.delta;
    d.free_mask = reinterpret_cast<const uint32_t*>(s.free_mask);
    d.mshift = s.mask_shift; d.mcount = s.free_mask ? s.mask_count : 0; d.mwords = s.mask_words;
    d.mby = (s.ny + (1 << s.mask_shift) - 1) >> s.mask_shift; d.mbz = (s.nz + (1 << s.mask_shift) - 1) >> s.mask_shift;
#pragma unroll
    for (int k = 0; k < VGPMP_MAX_MASKS; ++k) d.mclr[k] = s.mask_clearance[k];
    return d;
}

__device__ __forceinline__ int vg_voxel_axis(double rel, double origin, double delta, int n) {
    double q = (rel - origin) / delta;
    int hi = n - 1;
    return q < 0.0 ? 0 : (q > (double)hi ? hi : (int)q);
}

// brick of voxel (ix, iy, iz) and the voxel's place inside it (include/vgpmp.h, VGPMP_SDF_BRICK4)
__device__ __forceinline__ size_t vg_brick_of(const vg_sdf_dev& s, int ix, int iy, int iz) {
    return ((size_t)(ix >> 2) * s.nby + (iy >> 2)) * s.nbz + (iz >> 2);
}
__device__ __forceinline__ int vg_morton_in_brick(int ix, int iy, int iz) {
    return (iz & 1) | ((iy & 1) << 1) | ((ix & 1) << 2) | ((iz & 2) << 2) | ((iy & 2) << 3) | ((ix & 2) << 4);
}
// element of the table that holds voxel (ix, iy, iz)
__device__ __forceinline__ size_t vg_table_offset(const vg_sdf_dev& s, int ix, int iy, int iz) {
    if (s.layout == VGPMP_SDF_BRICK4) return vg_brick_of(s, ix, iy, iz) * 64 + vg_morton_in_brick(ix, iy, iz);
    return ((size_t)ix * s.ny + iy) * s.nz + iz;
}

__device__ __forceinline__ size_t vg_voxel_index(const vg_sdf_dev& s, double rx, double ry, double rz, int& ix,
                                                 int& iy, int& iz) {
    ix = vg_voxel_axis(rx, s.ox, s.delta, s.nx);
    iy = vg_voxel_axis(ry, s.oy, s.delta, s.ny);
    iz = vg_voxel_axis(rz, s.oz, s.delta, s.nz);
    return vg_table_offset(s, ix, iy, iz);
}

// r = u - f0(Z) - sqrt(jitter) eps'  of the Matheron update, in ONE written-out form: the path kernels (gp_path.hip, contraction
// allowed) and the likelihood that assembles its own paths (fk_sdf.hip, contraction off) must round identically
__device__ __forceinline__ float vg_path_r(float u, float f0z, float sqrt_jitter, float e2) { return fmaf(-sqrt_jitter, e2, u - f0z); }

// ---- Philox-4x32-10 (same schedule as oracle/vgpmp_oracle.py::philox4x32) --------------------
// One v_mad_u64_u32 per 32 x 32 -> 64 product (both halves from one instruction; the compiler emits v_mul_lo_u32 + v_mul_hi_u32
// for the C form  hi = __umulhi(M, c), lo = M * c) and one v_bitop3_b32 per three-way xor (two v_xor_b32 otherwise): 6 vector
// instructions per round instead of 10.  Every generator role uses this form: the draws of a large batch are a quarter of a step.
__device__ __forceinline__ uint4 vg_philox(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c.x), "s"(0xD2511F53u) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c.z), "s"(0xCD9E8D57u) : "vcc");
        c = make_uint4(__builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k.x, 0x96), (uint32_t)p1,
                       __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k.y, 0x96), (uint32_t)p0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float vg_u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }

// four standard normals of counter (i, stream, 0, 0): Box-Muller on lanes (0,1) and (2,3)
// Hardware transcendentals: v_log_f32 (log2) and v_sin/v_cos_f32, whose argument is in REVOLUTIONS --
// exactly the uniform of Box-Muller, so no range reduction and very little code.
__device__ __forceinline__ float4 vg_normal4(uint32_t i, uint32_t stream, uint2 key) {
    uint4 r = vg_philox(make_uint4(i, stream, 0u, 0u), key);
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(vg_u01(r.z)));
    const float u1 = vg_u01(r.y), u3 = vg_u01(r.w);
    return make_float4(r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1),
                       r1 * __builtin_amdgcn_cosf(u3), r1 * __builtin_amdgcn_sinf(u3));
}
// The W stream (prior weights: by far the largest draw of a step, S L B values per problem) spends ONE Philox counter per EIGHT
// weights and no transcendental at all: every 16-bit half of the block's four words is one weight, w = +-T[h & 0x1fff] with the sign
// from bit 15 of the half (bits 13, 14 unused).  T (csrc/gp_wtable.h, generated by tools/make_w_table.py) holds the means of |z|,
// z ~ N(0, 1), over 8192 equally probable bins of the half-normal distribution, as float16: a stratified inverse-CDF draw with
// 16 384 equally likely values -- E[w] = 0 exactly, Var[w] = 1 - 5e-6, |w| <= 4.074.  (Element 8 i + 2 j + {0, 1} of the stream
// comes from the {low, high} half of word j of counter i.)  The values are float16 on purpose: the prior kernels of large batches
// multiply W on the f16 matrix pipe (gp_prior_split.h), and a weight that IS a float16 has no low half to carry -- two MFMAs per
// float32-accurate product instead of three -- and the draw itself is forty Philox instructions and eight table reads where
// Box-Muller on hardware log / sqrt / sin / cos was a third of that kernel's vector time.  The reference draws float64 normals
// from TensorFlow's generator (models/vgpmp.py:281 through GPflowSampling); no implementation can match those bits, the
// restatement of THIS stream is oracle/vgpmp_oracle.py::philox_normals8 -- bit for bit, there is no rounding to disagree on.
// (vg_w8 and the table live in gp_common.h / gp_wtable.h: only the gp_path.hip translation unit draws W.)
__device__ __forceinline__ float vg_lane(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ float vg_normal1(uint32_t e, uint32_t stream, uint2 key) {
    return vg_lane(vg_normal4(e >> 2, stream, key), (int)(e & 3u));
}
__device__ __forceinline__ uint2 vg_key(uint32_t seed, uint32_t problem, uint32_t step) {
    return make_uint2(seed ^ (problem * 0x9E3779B1u), step);
}

enum { VG_STREAM_OMEGA = 0, VG_STREAM_CHI = 1, VG_STREAM_BETA = 2, VG_STREAM_W = 3, VG_STREAM_EPS = 4, VG_STREAM_EPS2 = 5 };

// Raises a kernel's dynamic-LDS limit when needed.  hipFuncSetAttribute is a slow host call and applies to the
// CURRENT device only, so the largest size granted is remembered per (device, kernel).  The table is guarded by a
// mutex: one process may drive several GPUs from several threads (DeviceScene._stream switches the device).
inline int vg_grant_dyn_lds(const void* fn, size_t bytes) {
    if (bytes > 160 * 1024) return VGPMP_E_SHAPE;
    if (bytes <= 48 * 1024) return 0;
    constexpr int kSlots = 256;
    static const void* fns[kSlots];
    static int devs[kSlots];
    static size_t granted[kSlots];
    static std::mutex mu;
    int dev = 0;
    VG_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int slot = -1;
    for (int i = 0; i < kSlots; ++i) {
        if (fns[i] == fn && devs[i] == dev) { slot = i; break; }
        if (fns[i] == nullptr) { fns[i] = fn; devs[i] = dev; slot = i; break; }
    }
    if (slot >= 0 && granted[slot] >= bytes) return 0;
    VG_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (slot >= 0) granted[slot] = bytes;
    return 0;
}

// ---- schedule log (include/vgpmp_debug.h: vgpmp_debug_last_schedule) ---------------------------------------------
// Per host thread: the names of the kernels a step enqueued, as written at the launch site (template arguments included; the
// profiler prints the same text).  A push of a string pointer per launch -- nothing on the device.
void vg_sched_clear();                                      // a new step begins
void vg_sched_note(const char* name);                       // "(kernel<...>)" or "kernel<...>"
const void* vg_fn_reg(const void* fn, const char* name);    // remembers the name of a kernel's host pointer, returns the pointer
void vg_sched_note_fn(const void* fn);                      // a launch through such a pointer
#define VG_FN(...) ([]() -> const void* { static const void* p_ = vg_fn_reg((const void*)(__VA_ARGS__), #__VA_ARGS__); return p_; }())
#define VG_GGL(kernel, ...) do { vg_sched_note(#kernel); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)
#define VG_EXT_GGL(kernel, ...) (vg_sched_note(#kernel), (void)hipExtLaunchKernelGGL(kernel, __VA_ARGS__))
int vg_launch_sphere_centres(const vgpmp_robot* rb, const float* f, int P, int S, int L, int N, int form, float* pos, hipStream_t st);

// launchers implemented in the .hip files
int vg_launch_sdf_pack(const vgpmp_sdf* sdf, const double* rows, int row_lo, int row_hi, int x0, int x1, hipStream_t st);
int vg_launch_sdf_free_mask(const vgpmp_sdf* sdf, hipStream_t st);
int vg_launch_fk_spheres(const vgpmp_robot* rb, const float* q, int64_t n, float* pos, float* frames, hipStream_t st);
int vg_launch_sdf_index_f32(const vgpmp_sdf* sdf, const double* offset, const float* pos, int64_t n, int32_t* idx, hipStream_t st);
int vg_launch_sdf_query(const vgpmp_sdf* sdf, const double* rel, int64_t n, int32_t* idx, float* dist, float* grad,
                        hipStream_t st);
int vg_launch_log_prob_impl(const vgpmp_robot* rb, int dof, const vgpmp_sdf* sdf, const float* g, int64_t n,
                            float* logp, float* dlogp, hipStream_t st);
// ELBO-path likelihood: f [P,S,L,N] -> G [P,S,L,N] (dloss/df), logp [P,S,N], lik_partial [P, nblk]
// ---- measurement builds (-DVGPMP_BISECT): in-kernel time stamps -----------------------------------
// VG_T(cond, id) records (id, 100 MHz wall clock) from thread 0 of the workgroups that satisfy `cond`,
// giving a timeline of one step across launches (tools/step_trace.py).  Compiled out of the product.
#ifdef VGPMP_BISECT
// one fixed slot per stamp id: a plain store, nothing to wait for (an atomic slot counter costs the stamping
// thread a memory round trip per stamp and stretches the phases it is meant to measure)
static __device__ unsigned long long vg_tr_buf[2048];
#define VG_T(cond, id) do { if (threadIdx.x == 0 && (cond)) vg_tr_buf[(id) & 2047] = wall_clock64(); } while (0)
// latest end over ALL workgroups that pass here (the slowest workgroup of a role)
#define VG_TMAX(id) do { __syncthreads(); if (threadIdx.x == 0) atomicMax(&vg_tr_buf[(id) & 2047], (unsigned long long)wall_clock64()); } while (0)
static int vg_trace_take(unsigned long long* host, int cap) {      // (id, stamp) pairs of this translation unit; clears them
    static unsigned long long tmp[2048];
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(vg_tr_buf), sizeof(tmp)) != hipSuccess) return -1;
    int n = 0;
    for (int i = 0; i < 2048 && n < cap; ++i)
        if (tmp[i]) { host[2 * n] = (unsigned long long)i; host[2 * n + 1] = tmp[i]; ++n; }
    for (int i = 0; i < 2048; ++i) tmp[i] = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vg_tr_buf), tmp, sizeof(tmp));
    return n;
}
int vg_trace_take_gp(unsigned long long* host, int cap);
int vg_trace_take_lik(unsigned long long* host, int cap);
#else
#define VG_T(cond, id) do { } while (0)
#define VG_TMAX(id) do { } while (0)
#endif

// The few-problem likelihood assembling the paths it needs itself (loglik_paths_wide_kernel<8, SIG, SK>, Mz = 32): what it reads
// besides the robot and the voxel table, and where r and f go.  U = m + C eps comes from stage B (cov_b role 0).
struct vg_lik_paths {
    int SK;                  // split-K slabs of the prior draws (2, 4 or 8)
    size_t slab;
    float sqrt_jitter;
    const float *AT, *F0, *U, *eps2;
    float *R, *f;
};

int vg_launch_loglik_paths(const vgpmp_robot* rb, const vgpmp_sdf* sdf, const float* f, int P, int S, int L, int N,
                           float scale, float* G, float* logp, float* lik_partial, int* nblk_out, hipStream_t st,
                           hipEvent_t kernel_start = nullptr, hipEvent_t kernel_end = nullptr,
                           const float* alpha_eff = nullptr, const float* sigma_eff = nullptr, float* sig_partial = nullptr,
                           int form = 0,       // 0: by batch size; 1: batch form; 2: batch form with all per-frame state in LDS
                           const vg_lik_paths* paths = nullptr);      // the eight-lane form assembles f (and r) itself: `f` is unused
// LDS the eight-lane form has to overlay the path operands on (bytes), and what they need; paths fit iff need <= room
bool vg_lik_paths_fit(int L, int SK);
int vg_loglik_blocks_per_problem(int S, int N);
int vg_launch_kernel_derivative(int kind, int order, const double* x, int n, const double* y, int m, double ell, double var,
                                double* out, hipStream_t st);
int vg_launch_cov_matrices(int kind, const double* Z, int nz, const double* X, int nx, int L, const double* ell,
                           const double* var, double jitter, double* out, hipStream_t st);
int vg_launch_velocity_kuu_kuf(int kind, const double* Zy, const double* X, int Mz, int N, int L, int D, const double* ell,
                               const double* var, double jitter, double* Kuu, double* Kuf, hipStream_t st);
int vg_launch_mesh_sdf(const double* tri, const int* part, int T, int nx, int ny, int nz, const double* origin,
                       double delta, double* grid, hipStream_t st);

// ---- race hunt build (-DVGPMP_RACE; tools/build_variant.sh race WORK -DVGPMP_RACE): behind EVERY workgroup barrier each wave sleeps a
// pseudo-random 0 .. ~4000 cycles, so the waves of a workgroup leave barriers in scrambled order and at scrambled distances -- what a
// busy neighbour on the device does to them once in a while (round 5's hunt for a race; there was none: DESIGN section 4).  Code that is correct only because its waves
// happen to run in lock step between two barriers then differs from run to run (tools/dbg_rep.py with VGPMP_HIP_LIB set).
#ifdef VGPMP_RACE
static __device__ __forceinline__ void vg_real_syncthreads() { __syncthreads(); }
static __device__ __forceinline__ void vg_jitter() {
    const unsigned h = ((unsigned)wall_clock64() * 2654435761u) ^ ((unsigned)(threadIdx.x >> 6) * 40503u) ^ ((unsigned)blockIdx.x * 7919u);
    const int n = (int)((h >> 9) & 63u);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#define __syncthreads() do { vg_real_syncthreads(); vg_jitter(); } while (0)
#endif
