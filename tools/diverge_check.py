import sys, os, argparse
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
args = bench.resolve(bench.parse_args(["--workload", "config3"]))
ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
print("split_k", pl.dims.split_k, "P", pl.P)
for blk in range(40):
    pl.run_steps(65)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(pl.q_mu).all(dim=(1, 2))
    ell = pl.lengthscales(); var = pl.variances()
    print(blk, "steps", pl.t, "bad problems", int(bad.sum()), "ell range", float(ell.min()), float(ell.max()), "var range", float(var.min()), float(var.max()),
          "q_sqrt absmax", float(pl.q_sqrt.abs().max()), "loss", float((-(pl.lik - pl.kl)).mean()))
    if bad.any():
        print("bad idx", torch.nonzero(bad).flatten().tolist())
        break
