#!/bin/bash
# Measurement aid (profiles/r06/flake.md, "What triggers it"): the library AS ROUND 5 SHIPPED IT (git archive d16839a, its own compiler
# flags: the vectorisers on, 10 438 packed-FP32 instructions) plus vgpmp_debug_mfma_load, so that tests/test_gpu_attach.py can be run
# on it (VGPMP_HIP_LIB=tools/libvgpmp_r5hook.so): the tests' teeth.  -> tools/libvgpmp_r5hook.so (git-ignored; travels to the GPU box).
set -euo pipefail
cd "$(dirname "$0")/.."
rev=${1:-d16839a}; src=/tmp/vgpmp_r5hook; rm -rf $src; mkdir -p $src/obj
git archive $rev vgpmp_amd/csrc include | tar -x -C $src
python - "$src/vgpmp_amd/csrc/capi.hip" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
hook = '''namespace {
typedef _Float16 vg_dbg_h8 __attribute__((ext_vector_type(8)));
typedef float vg_dbg_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void debug_mfma_load_kernel(float* __restrict__ sink, int iterations) {
    const float seed = (float)((blockIdx.x * 256u + threadIdx.x) & 1023u) * (1.0f / 1024.0f);
    vg_dbg_h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(seed + 0.125f * k); b[k] = (_Float16)(0.5f - seed); }
    vg_dbg_f4 c = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iterations * 16; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (c[0] + c[1] + c[2] + c[3] == 123.456f) sink[0] = c[0];
}
}  // namespace
extern "C" int vgpmp_debug_mfma_load(float* dev_sink, int32_t workgroups, int32_t iterations, vgpmp_stream stream) {
    hipLaunchKernelGGL(debug_mfma_load_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, dev_sink, (int)iterations);
    return (int)hipGetLastError();
}

'''
mark = 'extern "C" {\n\nconst char* vgpmp_version(void)'
assert mark in s
open(p, "w").write(s.replace(mark, hook + mark, 1))
PY
for f in fk_sdf gp_path mesh_sdf deriv_kernels plan inducing comm capi; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I$src/include -I$src/vgpmp_amd/csrc -c $src/vgpmp_amd/csrc/$f.hip -o $src/obj/$f.o &
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared $src/obj/*.o -ldl -o tools/libvgpmp_r5hook.so
python tools/audit_packed.py tools/libvgpmp_r5hook.so | head -1 | cut -c1-220
