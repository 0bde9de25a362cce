import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.resolve(bench.parse_args(["--workload", "stress"]))
ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
pl.loss_and_grad(generate=True, step=0)
torch.cuda.synchronize()
G = pl.view("G")
print("step0: lik", float(pl.lik.sum()), "G checksum", float(G.double().abs().sum()), "grad checksums", [float(g.double().abs().sum()) for g in pl.grad])
pl.reset()
for blk in range(4):
    pl.run_steps(50)
    pl.elbo(generate=True, step=10**6)
    torch.cuda.synchronize()
    ell = pl.lengthscales(); 
    print("after", 50 * (blk + 1), "steps: lik", float(pl.lik.sum()), "kl", float(pl.kl.sum()), "ell mean", float(ell.mean()), "q_sqrt absmean", float(pl.q_sqrt.abs().mean()))
