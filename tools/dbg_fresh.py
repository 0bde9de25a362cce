"""Measurement aid: ONE fresh-process run of the merged-vs-unmerged comparison (test_merged_launches_of_the_batch_schedule_change_nothing), checked
after every call; prints the first call after which a tensor differs.  Loop it from the shell: the rare difference needs a fresh process."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vgpmp_amd import capi, engine, robots as rb, scenes
ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
S, M, N, P = 64, 30, 40, 12
qs = np.array([ps.queries[i % 36] for i in range(P)])
kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
if mode == "ab":
    b.extra_flags |= capi.NO_FUSE
elif mode == "bb":
    a.extra_flags |= capi.NO_FUSE; b.extra_flags |= capi.NO_FUSE
names = ["q_mu", "q_sqrt", "raw_ell", "raw_var", "m0", "m1", "m2", "m3", "v0", "v1", "v2", "v3", "f", "lik", "kl"]
call = 0
for blk in range(3):
    for n in [8] * 5 + ["step"]:
        if n == "step":
            a.step(); b.step()
        else:
            a.run_steps(n); b.run_steps(n)
        call += 1
        torch.cuda.synchronize()
        xs = a._variables() + a._moments() + [a.f, a.lik, a.kl]; ys = b._variables() + b._moments() + [b.f, b.lik, b.kl]
        diff = [(names[i] if i < len(names) else i, float((x.double() - y.double()).abs().max()), int((x != y).sum())) for i, (x, y) in enumerate(zip(xs, ys)) if not torch.equal(x, y)]
        if diff:
            print("DIFF", mode, "after call", call, n, diff[:8], flush=True)
            sys.exit(1)
print("same", mode)
