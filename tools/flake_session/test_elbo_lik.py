import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_elbo_likelihood_repeats():
    """The ELBO step's own likelihood launch (the batch form: one lane per configuration) inside ONE planner's forward + reverse evaluation
    on FIXED noise (the same step index every time), over and over beside the starting rank processes; logp and G against the first
    evaluation.  With a -DVGPMP_CHK build of the library, G's joints 0-4 are checksums of the forward walk's operands (sphere positions,
    table addresses, gathered records, sphere constants, joint inputs): which of them differs says what went wrong."""
    from vgpmp_amd import capi, engine, robots as rb, scenes
    ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    S, M, N, P, L = 64, 30, 40, 12, 7
    qs = np.array([ps.queries[i % 36] for i in range(P)])
    pl = engine.PlannerBatch(sc, qs, num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
    def run():
        pl.loss_and_grad(generate=True, step=3)
        return pl.logp.clone(), pl.view("G").clone(), pl.f.clone()
    ref = run(); torch.cuda.synchronize()
    print("\nkernels:", [k for k in capi.last_schedule(pl.lib) if k.startswith("loglik")], flush=True)
    t_end = time.time() + float(os.environ.get("FLAKE_SECONDS", "12"))
    k = 0
    while time.time() < t_end:
        k += 1
        out = run(); torch.cuda.synchronize()
        if all(torch.equal(x, y) for x, y in zip(out, ref)):
            continue
        print("\nELBO LIKELIHOOD DIFFERS repetition", k, "| f equal:", torch.equal(out[2], ref[2]), flush=True)
        x, y = out[0].reshape(P, -1), ref[0].reshape(P, -1)
        for pp in range(P):
            bad = torch.nonzero(x[pp] != y[pp]).flatten()
            if len(bad):
                print(f"   logp problem {pp}: flat (s N + n) indices", [int(v) for v in bad[:40]], flush=True)
        gx, gy = out[1].reshape(P, S, L, N), ref[1].reshape(P, S, L, N)
        for pp in range(P):
            d = gx[pp] != gy[pp]                               # [S, L, N]
            if not d.any():
                continue
            cfg = torch.nonzero(d.any(dim=1).reshape(-1)).flatten()      # flat s N + n
            per_joint = [int(v) for v in d.sum(dim=(0, 2))]
            print(f"   G problem {pp}: configurations", [int(v) for v in cfg[:40]], "| differing entries per joint", per_joint, flush=True)
            for c in [int(v) for v in cfg[:4]]:
                s_, n_ = c // N, c % N
                print(f"      config {c}: got ", [float(v) for v in gx[pp, s_, :, n_]], "\n                 want", [float(v) for v in gy[pp, s_, :, n_]], flush=True)
        assert False
    print("\nelbo likelihood:", k, "repetitions, all equal", flush=True)
