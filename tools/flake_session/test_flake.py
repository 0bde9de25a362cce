import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_first_difference():
    """Two planners of the SAME schedule (both merged, or both one launch per kernel: FLAKE_MODE=aa|bb) stepped call by call; every state and
    work tensor compared after every call: which tensor differs first, and after which call."""
    from vgpmp_amd import capi, engine, robots as rb, scenes
    ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    off = ps.object_positions[0]
    if os.environ.get("FLAKE_ROBOT") == "synthetic14":      # (the 14-joint arm of config 5: the register / pipelined forms of the likelihood)
        spec, off = rb.synthetic_arm(14), (0.05, -0.03, 0.02)
    sc = engine.DeviceScene(spec, grid, off)
    S, M, N, P = 64, 30, 40, int(os.environ.get("FLAKE_P", "12"))
    L_ = 7
    qs = np.array([ps.queries[i % 36] for i in range(P)])
    if spec.dof != 7:
        L_ = spec.dof
        qs = np.random.default_rng(3).uniform(-2.0, 2.0, (P, 2, L_))
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * L_, variance=0.2, seed=4)
    if os.environ.get("FLAKE_LR"):      # (a -DVGPMP_CHK build hands checksums back as gradients: keep the variables where they are)
        kw["learning_rate"] = float(os.environ["FLAKE_LR"])
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    mode = os.environ.get("FLAKE_MODE", "aa")
    a.extra_flags |= int(os.environ.get("FLAKE_FLAGS", "0")); b.extra_flags |= int(os.environ.get("FLAKE_FLAGS", "0"))
    if mode == "bb":
        a.extra_flags |= capi.NO_FUSE; b.extra_flags |= capi.NO_FUSE
    views = ("R", "G", "A4", "C", "m", "F0", "H", "epsT", "eps2T", "kl_l", "Kinv")
    def state(p):
        d = {"q_mu": p.q_mu, "q_sqrt": p.q_sqrt, "raw_ell": p.raw_ell, "raw_var": p.raw_var, "f": p.f, "logp": p.logp, "lik": p.lik, "kl": p.kl,
             "omega": p.omega, "beta": p.beta, "eps": p.eps}
        for v in views:
            try:
                d["ws." + v] = p.view(v)
            except Exception:
                pass
        return d
    nper = int(os.environ.get("FLAKE_STEPS", "8"))
    light = True      # (keep the device busy: variables and paths only, no workspace copies, until something differs)
    import time
    t_end = time.time() + float(os.environ.get("FLAKE_SECONDS", "12"))
    call = 0
    while time.time() < t_end:
        call += 1
        a.run_steps(nper); b.run_steps(nper)
        torch.cuda.synchronize()
        if light:
            sa = {"q_mu": a.q_mu, "q_sqrt": a.q_sqrt, "raw_ell": a.raw_ell, "raw_var": a.raw_var, "f": a.f, "logp": a.logp, "lik": a.lik, "kl": a.kl}
            sb = {"q_mu": b.q_mu, "q_sqrt": b.q_sqrt, "raw_ell": b.raw_ell, "raw_var": b.raw_var, "f": b.f, "logp": b.logp, "lik": b.lik, "kl": b.kl}
        else:
            sa, sb = state(a), state(b)
        diff = [(k, float((sa[k].double() - sb[k].double()).abs().max()), int((sa[k] != sb[k]).sum())) for k in sa if not torch.equal(sa[k], sb[k])]
        if diff:
            print("\nkernels:", [k for k in capi.last_schedule(a.lib) if k.startswith("loglik")], flush=True)
            print("\nFIRST DIFFERENCE", mode, "steps per call", nper, "after call", call, diff, flush=True)
            for k in ("q_mu", "lik", "kl", "raw_ell", "f"):
                x, y = sa[k].double(), sb[k].double()
                per_problem = (x - y).abs().reshape(x.shape[0], -1).max(dim=1).values
                print("  ", k, "problems that differ:", [int(i) for i in torch.nonzero(per_problem > 0).flatten()], flush=True)
            # where: flat (sample, time) index of the log-densities that differ = lane + 64 workgroup of the likelihood launch; G by (s, l, n)
            for nm, x, y in (("logp", a.logp, b.logp), ("G", a.view("G"), b.view("G"))):
                x, y = x.reshape(P, -1), y.reshape(P, -1)
                for pp in range(P):
                    bad = torch.nonzero(x[pp] != y[pp]).flatten()
                    if len(bad) == 0:
                        continue
                    if nm == "logp":
                        print(f"   logp problem {pp}: flat (s N + n) indices", [int(v) for v in bad[:40]], "| values a", [round(float(v), 3) for v in x[pp][bad[:8]]],
                              "b", [round(float(v), 3) for v in y[pp][bad[:8]]], flush=True)
                    else:
                        sidx, rem = bad // (L_ * N), bad % (L_ * N)
                        lidx, nidx = rem // N, rem % N
                        cfg = sorted(set(int(v) for v in (sidx * N + nidx)))
                        print(f"   G problem {pp}: {len(bad)} entries on {len(cfg)} configurations, flat indices", cfg[:40], "| joints hit", sorted(set(int(v) for v in lidx)),
                              "| entries per joint", [int((lidx == j).sum()) for j in range(L_)], flush=True)
                        gx, gy = x[pp].reshape(S, L_, N), y[pp].reshape(S, L_, N)
                        for c in cfg[:3]:
                            print(f"      config {c}: a", [float(v) for v in gx[c // N, :, c % N]], "\n                  b", [float(v) for v in gy[c // N, :, c % N]], flush=True)
            wa, wb = state(a), state(b)
            for k in wa:
                if k.startswith("ws.") or k in ("omega", "beta", "eps"):
                    print("  ", k, "max |diff|", float((wa[k].double() - wb[k].double()).abs().max()), "entries", int((wa[k] != wb[k]).sum()), "of", wa[k].numel(), flush=True)
            assert False, diff
