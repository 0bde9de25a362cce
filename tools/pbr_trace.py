"""Measurement aid: phases of ONE workgroup of paths_bwd_regs at the config-5 share (measurement build, VG_PBT stamps of
csrc/gp_paths.h).   VGPMP_HIP_LIB=tools/libvgpmp_bisect.so python tools/pbr_trace.py [bench.py arguments]
Per pair of chunks: us from kernel-side stamp to stamp -- requests issued | operands landed | five products | chunk 0 | chunk 1."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vgpmp_amd import capi  # noqa: E402


def main():
    args = bench.resolve(bench.parse_args(["--workload", "stress", "--no-cpu-baseline", "--no-solve"] + sys.argv[1:]))
    ps, spec, grid, scene, planner = bench.build_problem(0, args, 1)
    lib = capi.load()
    lib.vgpmp_debug_trace.argtypes = [C.c_void_p, C.c_int32]
    lib.vgpmp_debug_trace.restype = C.c_int
    buf = np.zeros(2 * 8192, dtype=np.uint64)
    planner.run_steps(10)
    torch.cuda.synchronize()
    lib.vgpmp_debug_trace(buf.ctypes.data, 8192)
    planner.run_steps(3)
    torch.cuda.synchronize()
    n = lib.vgpmp_debug_trace(buf.ctypes.data, 8192)
    st = {int(buf[2 * i]): int(buf[2 * i + 1]) for i in range(n)}
    names = {9: "B fragments requested (kernel start ~ here)", 0: "pair: top", 1: "staging requests issued", 2: "operands landed", 3: "five products done",
             4: "chunk 0 done", 5: "chunk 1 done"}
    ev = sorted((t, k) for k, t in st.items() if 1200 <= k < 1300)
    t0 = ev[0][0] if ev else 0
    prev = t0
    for t, k in ev:
        pair, ph = (k - 1200) // 10, (k - 1200) % 10
        print(f"{(t - t0) / 100:7.2f} us  (+{(t - prev) / 100:5.2f})  pair {pair}  {names.get(ph, ph)}")
        prev = t


if __name__ == "__main__":
    main()
