"""Measurement / QA aid: random shapes through the C ABI against the oracle (loss and q_mu / hyper-parameter gradients).
    python tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import small_problem
from oracle import vgpmp_oracle as orc
from vgpmp_amd import engine
from vgpmp_amd import robots as rb


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(ncase):
        S = int(rng.choice([5, 8, 16, 20, 24, 33, 48, 64]))
        N = int(rng.choice([7, 12, 20, 40, 50]))
        M = int(rng.choice([3, 6, 10, 14, 22, 30]))
        B = int(rng.choice([64, 128, 256]))
        P = int(rng.choice([1, 1, 2, 3, 5, 6, 10, 12, 16]))      # (10 up: beyond the few-problem schedule at any sample count)
        robot = str(rng.choice(["franka", "wam", "ur10"]))
        pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=int(rng.integers(1000)), n_grid=32)
        sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
        ps = rb.load_problemset(robot if robot in rb.AVAILABLE_ROBOTS else "franka", "industrial")
        pl = engine.PlannerBatch(sc, np.repeat(pb["y"][None], P, 0), num_samples=S, num_inducing=M, num_data=N, num_bases=B,
                                 lengthscales=ps.planner_params["lengthscales"], variance=ps.planner_params["variance"],
                                 alpha=pb["alpha"], learning_rate=pb["lr"])
        p = pb["params"]
        for k in range(P):
            pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
            pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
        r32 = lambda a: a.astype(np.float32).astype(np.float64)
        nz = pb["noise"]
        nz = orc.Noise(r32(nz.omega), r32(nz.beta), r32(nz.w), r32(nz.eps), r32(nz.eps2))
        rep = lambda a: np.repeat(a[None], P, 0)
        pl.set_noise(rep(nz.omega), rep(nz.beta), rep(nz.w), rep(nz.eps), rep(nz.eps2))
        # every third case: the batch form of the likelihood (one lane per configuration) and one launch per kernel forced on the
        # small problem -- the forms large batches run
        form = case % 3
        if form == 1:
            from vgpmp_amd import capi
            pl.extra_flags = capi.LIK_LANES
            pl.fuse = False
        elif form == 2:
            pl.fuse = False
        loss, grads = pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        fw = orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], pb["y"], nz, pb["alpha"])
        og, _ = orc.elbo_backward(p, pb["scene"], pb["X"], pb["Zy"], nz, pb["alpha"], fw)
        el = max(abs(float(loss[k]) + fw["elbo"]) / (abs(fw["elbo"]) + 1e-9) for k in range(P))
        eq = max(np.abs(grads[0][k].cpu().numpy().T - og.q_mu).max() / (np.abs(og.q_mu).max() + 1e-12) for k in range(P))
        ee = max(np.abs(grads[2][k].cpu().numpy() - og.raw_ell).max() / (np.abs(og.raw_ell).max() + 1e-12) for k in range(P))
        ev = max(np.abs(grads[3][k].cpu().numpy() - og.raw_var).max() / (np.abs(og.raw_var).max() + 1e-12) for k in range(P))
        worst = max(worst, el, eq)
        print(f"case {case:2d} form {form} {robot:6s} P={P} S={S:2d} N={N:2d} M={M:2d} B={B:3d} sk={pl.dims.split_k}: loss {el:.1e}  dq_mu {eq:.1e}  dell {ee:.1e}  dvar {ev:.1e}"
              + ("   <-- CHECK" if max(el, eq) > 2e-2 else ""))
    print("worst relative deviation (loss, dq_mu):", f"{worst:.2e}")


if __name__ == "__main__":
    main()
