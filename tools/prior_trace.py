"""Measurement aid: phase timeline of ONE workgroup of prior_fused_split_kernel at the config-5 share (measurement build with the
VG_PT stamps of csrc/gp_prior_split.h).   VGPMP_HIP_LIB=tools/libvgpmp_bisect.so python tools/prior_trace.py
Per K step (8..15) and wave: microseconds spent in  frequency-tile store | features + W draw | wait at barrier 1 | products | wait at
barrier 2.  (The first column is an artefact of the stamps: the store of the tile waits on vmcnt(0), which in this build also drains the
stamp's own global store -- ~0.3 us; in the product build the tile's values were requested a whole product phase earlier.)"""
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
    names = ("om store*", "features + W", "barrier 1", "products", "barrier 2")
    tot = np.zeros((8, 5))
    steps = 0
    for ki in range(8):
        row = []
        ok = all(640 + 48 * ki + 8 * ph + w in st for ph in range(6) for w in range(8))
        if not ok:
            continue
        steps += 1
        for w in range(8):
            t = [st[640 + 48 * ki + 8 * ph + w] for ph in range(6)]
            assert all(b >= a for a, b in zip(t, t[1:])), f"stamps of K step {8 + ki}, wave {w} out of order (ids shared with another kernel?): {t}"
            tot[w] += np.diff(t) / 100.0
        t0 = min(st[640 + 48 * ki + w] for w in range(8))
        t1 = max(st[640 + 48 * ki + 40 + w] for w in range(8))
        print(f"K step {8 + ki}: {(t1 - t0) / 100.0:6.2f} us")
    print("mean us per K step and wave:  " + " | ".join(names))
    for w in range(8):
        print(f"  wave {w}: " + " | ".join(f"{v / max(steps, 1):6.2f}" for v in tot[w]) + f"   sum {tot[w].sum() / max(steps, 1):6.2f}")
    print("* includes the drain of the stamp's own store (vmcnt(0)): an artefact of this build")


if __name__ == "__main__":
    main()
