"""Measurement aid: does the 14-joint batch likelihood give the same bits from fresh scenes / planners, again and again?
(One full-suite run of round 5 saw 39 of 12 800 log-densities of tests/test_gpu_config5.py's full-size case off by a voxel.)"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import synthetic_problem
from vgpmp_amd import engine
import test_gpu_config5 as T
S, N, M, B, P = 128, 100, 30, 256, 6
pb = synthetic_problem(dof=14, S=S, N=N, M=M, B=B, seed=13, n_grid=48, n_problems=P)
ref = None
bad = 0
junk = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    junk.append(torch.full((int(np.random.randint(1, 64)) << 18,), float("nan"), device="cuda"))      # stir the allocator, poison what is freed
    if len(junk) > 3:
        junk.pop(0)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"], free_space_summary=True)
    pl, nz = T._batch(pb, sc, S, N, M, B)
    pl.loss_and_grad(generate=False)
    cen = pl.sphere_centres()
    torch.cuda.synchronize()
    cur = (pl.logp.clone(), pl.f.clone(), cen.clone(), pl.grad[1].clone())
    if ref is None:
        ref = cur
    else:
        same = [bool(torch.equal(a, b)) for a, b in zip(ref, cur)]
        if not all(same):
            bad += 1
            d = (ref[0] - cur[0]).abs()
            print("rep", rep, "same (logp, f, centres, grad)", same, "logp pairs off", int((d > 0).sum()), "max", float(d.max()))
    del sc, pl
print("repetitions with different bits:", bad)
