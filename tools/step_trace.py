"""Measurement aid: timeline of one training step from in-kernel time stamps (measurement build).

    VGPMP_HIP_LIB=tools/libvgpmp_bisect.so python tools/step_trace.py [problems]

Prints, for the last full step of a short run, every stamp (kernel*100 + role*10 + phase) with its time in
microseconds relative to the first stamp of the step."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgpmp_amd import capi, engine, robots as rb, scenes  # noqa: E402

NAMES = {100: "cov_a start", 101: "cov_a Kuu built", 102: "cov_a factorised", 103: "cov_a end",
         110: "final start", 115: "final requests issued", 114: "final operands landed", 111: "final partials summed", 112: "final grads", 113: "final end",
         150: "final (in stage 1) start", 155: "final (in stage 1) requests issued", 154: "final (in stage 1) operands landed",
         151: "final (in stage 1) partials summed", 152: "final (in stage 1) grads", 153: "final (in stage 1) end",
         160: "cov_a slowest wg end", 161: "final slowest wg end", 162: "eps slowest wg end", 163: "features slowest wg end",
         104: "cov_a panel 1 loaded", 105: "cov_a panel 1 eliminated", 106: "cov_a panel 1 stored", 107: "cov_a second block row formed",
         108: "cov_a panel 2 loaded", 109: "cov_a panel 2 eliminated", 140: "cov_a panel 2 stored", 141: "cov_a workgroup joined",
         120: "eps start", 130: "features first wg start", 131: "features first wg end", 135: "features last wg end",
         200: "cov_b KL start", 201: "cov_b KL staged", 202: "cov_b KL end",
         250: "cov_b q_sqrt role start", 251: "cov_b q_sqrt role staged", 252: "cov_b q_sqrt role end",
         210: "cov_b d/dell start", 211: "cov_b d/dell staged", 212: "cov_b d/dell matmuls", 213: "cov_b d/dell end",
         220: "cov_b d/dvar start", 221: "cov_b d/dvar staged", 222: "cov_b d/dvar matmuls", 223: "cov_b d/dvar end",
         214: "d/dell af done", 215: "d/dell matmul 1", 216: "d/dell matmul 2", 217: "d/dell matmul 3",
         224: "d/dvar af done", 225: "d/dvar matmul 1", 226: "d/dvar matmul 2", 227: "d/dvar matmul 3",
         233: "rows Kfu built", 234: "rows A", 235: "rows tangent 1",
         230: "rows start", 232: "rows staged", 231: "rows end",
         240: "gemm first wg start", 241: "gemm first wg mfma done", 242: "gemm first wg end", 245: "gemm last wg end",
         300: "paths_fwd start", 301: "paths_fwd staged", 302: "paths_fwd end", 305: "paths_fwd last wg end",
         310: "rng basis start", 320: "rng w start", 321: "rng w first wg end", 325: "rng w last wg end",
         400: "loglik first wg start", 402: "loglik robot + f loaded", 403: "loglik frames done", 404: "loglik spheres done", 401: "loglik first wg end", 405: "loglik last wg end",
         500: "paths_bwd start", 501: "paths_bwd staged", 502: "paths_bwd loops", 503: "paths_bwd end",
         505: "paths_bwd last wg end", 506: "paths_bwd requests issued", 507: "paths_bwd operands landed", 600: "hyper start", 601: "hyper end",
         1300: "prior small16 first wg start", 1301: "prior small16 points staged", 1302: "prior small16 K loop done", 1303: "prior small16 first wg end",
         **{1500 + 8 * o + i: f"cov_b order {o} ({('d/dvar', 'd/dell', 'KL', 'q_sqrt', 'rows 0', 'rows 1', 'rows 2', 'rows 3')[o]}) latent {64 * i} start" for o in range(8) for i in range(8)},
         **{1400 + 8 * o + i: f"cov_b order {o} ({('d/dvar', 'd/dell', 'KL', 'q_sqrt', 'rows 0', 'rows 1', 'rows 2', 'rows 3')[o]}) latent {64 * i} end" for o in range(8) for i in range(8)},
         1110: "cov_b KL: a = Lk^-1 delta", 1111: "cov_b KL: Q terms", 1112: "cov_b KL: summed",
         1100: "cov_a diagonal scales ready", 1101: "cov_a factors written", 142: "cov_a inverse formed", 143: "rows tail: dKuu/dell requested", 144: "rows tail: Kfu of tile 0", 145: "rows tail: operands stand", 146: "rows tail: tile 0 products start",
         147: "rows tail: tile 0 products done", 148: "rows tail: tile 4 start", 149: "rows tail: tile 4 products done",
         1600: "rows wave: loads issued", 1601: "rows wave: staged", 1602: "rows wave: Kfu formed", 1603: "rows wave: products done", 1604: "rows wave: stored",
         **{1700 + 16 * r + k: f"mid_stage1 {('gradient assembly', 'stage A', 'basis draws', 'normal draws', 'eps draws')[r]} wg {k}/8 start" for r in range(5) for k in range(8)},
         **{1800 + 16 * r + k: f"mid_stage1 {('gradient assembly', 'stage A', 'basis draws', 'normal draws', 'eps draws')[r]} wg {k}/8 end" for r in range(5) for k in range(8)},
         **{1900 + 2 * i + x: f"paths_bwd_regs wg (x={x}, latent 0, problem {16 * i}) start" for i in range(8) for x in range(2)},
         **{1930 + 2 * i + x: f"paths_bwd_regs wg (x={x}, latent 0, problem {16 * i}) end" for i in range(8) for x in range(2)},
         **{1960 + 2 * i + x: f"paths_fwd_regs wg (x={x}, latent 0, problem {16 * i}) start" for i in range(8) for x in range(2)},
         **{1990 + 2 * i + x: f"paths_fwd_regs wg (x={x}, latent 0, problem {16 * i}) end" for i in range(8) for x in range(2)},
         **{1320 + 4 * i + y: f"prior small16 wg ({64 * i}, {y}) start" for i in range(8) for y in range(4)},
         **{1360 + 4 * i + y: f"prior small16 wg ({64 * i}, {y}) end" for i in range(8) for y in range(4)}}


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib = capi.load()
    lib.vgpmp_debug_trace.argtypes = [C.c_void_p, C.c_int32]
    lib.vgpmp_debug_trace.restype = C.c_int
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=128, delta=1.6 / 128, origin=(-0.8, -0.8, -0.2), seed=0)
    pp = ps.planner_params
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    env = lambda k, d: int(os.environ.get(k, d))      # SAMPLES / INDUCING / TIMESTEPS: other shapes than config 2's
    pl = engine.PlannerBatch(sc, qs, num_samples=env("SAMPLES", 128), num_inducing=env("INDUCING", 30), num_data=env("TIMESTEPS", 100), num_bases=env("BASES", 1024),
                             lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                             learning_rate=pp["learning_rate"], seed=1)
    pl.fuse = os.environ.get("NO_FUSE") is None
    buf = np.zeros(2 * 8192, dtype=np.uint64)
    pl.run_steps(30)
    torch.cuda.synchronize()
    lib.vgpmp_debug_trace(buf.ctypes.data, 8192)          # discard
    pl.run_steps(6)
    torch.cuda.synchronize()
    n = lib.vgpmp_debug_trace(buf.ctypes.data, 8192)
    # every stamp id keeps its LAST value: the stamps of the final steps of the run.  The last step of a call ends with
    # the stand-alone hyper / final launches; shift times so that the latest cov_a start is zero and show one period
    ev = sorted((int(buf[2 * i + 1]), int(buf[2 * i])) for i in range(n))
    t_cov = max(t for t, k in ev if k == 100)
    period = None
    print(f"{P} problem(s), fuse={pl.fuse}; times relative to the last cov_a start (negative = previous step)")
    for t, k in ev:
        print(f"{(t - t_cov) / 100:8.2f} us  {k:4d}  {NAMES.get(k, '')}")


if __name__ == "__main__":
    main()
