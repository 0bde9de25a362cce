"""Measurement / QA aid: random shapes, robots, batch sizes and trainable subsets through the C ABI against the oracle.
Every case: (a) one injected-noise evaluation -- log-density of every (sample, time) pair, ELBO pieces, every gradient on the device's
own voxels at the fixed tolerances of tests/helpers.py; (b) a few optimisation steps on the device's own generated noise, the oracle
following step by step (tests/helpers.py::follow_device_trajectory) -- multi-step calls, so the merged launches of a call's later
steps run.      python tools/fuzz_parity.py [n_cases] [seed] [regs]
(regs: shapes of the register-resident path kernels of large batches -- Mz = 32, 64 or 128 samples, 6-12 problems -- which the uniform
draw of shapes meets once in a hundred cases)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import TOL_LIK, TOL_LOGP, assert_grads, device_centres, follow_device_trajectory, small_problem
from oracle import vgpmp_oracle as orc
from vgpmp_amd import capi, engine
from vgpmp_amd import robots as rb


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for case in range(ncase):
        S = int(rng.choice([5, 7, 8, 16, 20, 24, 33, 48, 64, 128]))
        N = int(rng.choice([7, 12, 20, 40, 50, 100]))
        M = int(rng.choice([3, 6, 7, 10, 14, 22, 24, 30, 46]))
        B = int(rng.choice([64, 128, 256]))
        P = int(rng.choice([1, 1, 2, 3, 5, 6, 10, 12, 16, 24]))
        robot = str(rng.choice(["franka", "wam", "ur10", "kuka"]))
        if len(sys.argv) > 3 and sys.argv[3] == "regs":
            S, M, N, P = int(rng.choice([64, 128])), 30, int(rng.choice([12, 20, 40])), int(rng.choice([6, 10, 12]))
        if S * N * P * (M + 2) > 2_000_000:      # keep the oracle in seconds
            P = max(1, 2_000_000 // (S * N * (M + 2)))
        pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=int(rng.integers(1000)), n_grid=32)
        D = pb["spec"].dof
        sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
        ps = rb.load_problemset(robot, "industrial")
        pp = dict(ps.planner_params, alpha=pb["alpha"], learning_rate=pb["lr"])
        ys = np.repeat(pb["y"][None], P, 0) + 0.02 * rng.standard_normal((P, 2, D))
        ys = np.clip(ys, pb["spec"].low + 0.05, pb["spec"].high - 0.05)
        tag = f"case {case}: {robot} S={S} N={N} M={M} B={B} P={P}"
        # ---- (a) injected noise
        pl = engine.PlannerBatch(sc, ys, num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=pp["lengthscales"],
                                 variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"])
        mode = int(rng.integers(3))
        if mode == 1:
            pl.extra_flags |= capi.LIK_LANES; pl.fuse = False
        X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
        var = max(float(pp["variance"]), 0.1 + 1e-6)
        params, noises = [], []
        r32 = lambda a: a.astype(np.float32).astype(np.float64)
        for k in range(P):
            p = orc.init_params(pb["scene"].robot, ys[k], M, pp["lengthscales"], var)
            p.q_sqrt = np.tril(p.q_sqrt + 0.05 * rng.standard_normal(p.q_sqrt.shape))
            p.q_mu = p.q_mu + 0.05 * rng.standard_normal(p.q_mu.shape)
            nz = orc.draw_noise(rng, S, D, D, B, M + 2)
            params.append(p); noises.append(orc.Noise(r32(nz.omega), r32(nz.beta), r32(nz.w), r32(nz.eps), r32(nz.eps2)))
            pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        st = lambda name: np.stack([getattr(nz, name) for nz in noises])
        pl.set_noise(st("omega"), st("beta"), st("w"), st("eps"), st("eps2"))
        loss, grads = pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        for k in sorted(set([0, P - 1])):
            fw = orc.elbo_forward(params[k], pb["scene"], X, Zy, ys[k], noises[k], float(pp["alpha"]), lookup_pos=device_centres(pl, k))
            og, _ = orc.elbo_backward(params[k], pb["scene"], X, Zy, noises[k], float(pp["alpha"]), fw)
            top = np.abs(fw["logp"]).max()
            np.testing.assert_allclose(pl.logp[k].cpu().numpy(), fw["logp"], rtol=0, atol=TOL_LOGP * top + 1e-30, err_msg=tag)
            np.testing.assert_allclose(float(pl.lik[k]), fw["lik"], rtol=TOL_LIK, atol=1e-12, err_msg=tag)
            np.testing.assert_allclose(float(pl.kl[k]), fw["cv"]["kl"], rtol=1e-9, err_msg=tag)
            assert_grads(f"{tag} mode={mode} k={k}", grads, og, k=k)
        # ---- (b) generated noise, multi-step calls, a random trainable subset
        tr = dict(q_mu=True, q_sqrt=bool(rng.integers(2)), lengthscales=bool(rng.integers(2)), kernel_variance=bool(rng.integers(2)))
        Pb = min(P, 4)
        pl2 = engine.PlannerBatch(sc, ys[:Pb], num_samples=S, num_inducing=M, num_data=N, num_bases=1024, lengthscales=pp["lengthscales"],
                                  variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=int(rng.integers(1 << 20)),
                                  problem_base=int(rng.integers(100)), trainable=tr)
        if tr == orc.DEFAULT_TRAINABLE:
            follow_device_trajectory(tag + " follow", pl2, pb["scene"], ys[:Pb], pp, var, 3, pl2.seed, pl2.problem_base)
        # ... and a multi-step call equals the same steps one by one (the merged launches of later steps), bit for bit
        a = engine.PlannerBatch(sc, ys, num_samples=S, num_inducing=M, num_data=N, num_bases=1024, lengthscales=pp["lengthscales"],
                                variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=11, trainable=tr)
        b = engine.PlannerBatch(sc, ys, num_samples=S, num_inducing=M, num_data=N, num_bases=1024, lengthscales=pp["lengthscales"],
                                variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=11, trainable=tr)
        a.run_steps(5)
        for _ in range(5):
            b.run_steps(1)
        torch.cuda.synchronize()
        for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var), (a.adam_v[1], b.adam_v[1]),
                     (a.grad[1], b.grad[1]), (a.lik, b.lik), (a.kl, b.kl)):
            assert torch.equal(x, y), tag + f" trainable={tr}: a 5-step call differs from five 1-step calls"
        assert torch.isfinite(a.q_mu).all(), tag
        print("ok", tag, "trainable", {k: v for k, v in tr.items() if not v} or "all", flush=True)
    print(f"{ncase} cases passed")


if __name__ == "__main__":
    main()
