import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import vgpmp_oracle as orc
from helpers import synthetic_problem, device_centres
from vgpmp_amd import engine
import test_gpu_config5 as T
S, N, M, B, P = 128, 100, 30, 256, 6
pb = synthetic_problem(dof=14, S=S, N=N, M=M, B=B, seed=13, n_grid=48, n_problems=P)
for summary in (True, False, True, True):
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"], free_space_summary=summary)
    pl, nz = T._batch(pb, sc, S, N, M, B)
    outs = []
    for rep in range(3):
        pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        outs.append((pl.logp.clone(), pl.f.clone(), pl.sphere_centres().clone()))
    same = [bool(torch.equal(outs[0][i], outs[r][i])) for r in (1, 2) for i in range(3)]
    k = 0
    p, y = pb["params"][k], pb["ys"][k]
    cen = outs[0][2][k].cpu().numpy().astype(np.float64)
    fw = orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], y, nz[k], pb["alpha"], lookup_pos=cen, want_dell=False)
    for rep in range(3):
        d = np.abs(outs[rep][0][k].cpu().numpy() - fw["logp"])
        print("summary", summary, "rep", rep, "repeatable", same, "bad pairs", int((d > 2e-6 * np.abs(fw["logp"]).max()).sum()), "max", d.max())
