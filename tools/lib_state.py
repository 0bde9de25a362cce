"""Validation aid: the variables after a few steps of a bench.py workload, written to an .npz -- run once per library
(VGPMP_HIP_LIB=tools/libvgpmp_<name>.so) and compare:   python tools/lib_state.py out.npz [steps] -- [bench.py arguments]
                                                        python tools/lib_state.py --compare a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
        for k in a.files:
            print(k, "equal" if k not in bad else f"DIFFER max |d| = {np.abs(a[k] - b[k]).max():.3e}")
        sys.exit(1 if bad else 0)
    import torch
    import bench
    out = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "--" else 5
    extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
    args = bench.resolve(bench.parse_args(["--no-cpu-baseline", "--no-solve"] + extra))
    ps, spec, grid, scene, planner = bench.build_problem(0, args, 1)
    planner.run_steps(steps)
    torch.cuda.synchronize()
    np.savez(out, q_mu=planner.q_mu.cpu().numpy(), q_sqrt=planner.q_sqrt.cpu().numpy(), raw_ell=planner.raw_ell.cpu().numpy(),
             raw_var=planner.raw_var.cpu().numpy(), m_q=planner.adam_m[1].cpu().numpy(), v_ell=planner.adam_v[2].cpu().numpy(),
             grad_q=planner.grad[1].cpu().numpy(), lik=planner.lik.cpu().numpy(), kl=planner.kl.cpu().numpy())
    print("wrote", out, "finite", bool(np.isfinite(planner.q_mu.cpu().numpy()).all()))


if __name__ == "__main__":
    main()
