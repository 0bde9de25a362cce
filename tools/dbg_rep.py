"""Measurement aid: the merged-vs-unmerged bitwise comparison of tests/test_gpu_surface.py, repeated; prints which tensors differ when one does."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vgpmp_amd import capi, engine, robots as rb, scenes
ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for rep in range(reps):
    for (S, M, N, P) in ((64, 30, 40, 12), (7, 24, 70, 20)):
        qs = np.array([ps.queries[i % 36] for i in range(P)])
        kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
        a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
        b.extra_flags |= capi.NO_FUSE
        first = None
        for blk in range(3):
            a.run_steps(40); b.run_steps(40)
            a.step(); b.step()
            torch.cuda.synchronize()
            names = ["q_mu", "q_sqrt", "raw_ell", "raw_var"]
            xs = a._variables() + a._moments() + [a.f, a.lik, a.kl]; ys = b._variables() + b._moments() + [b.f, b.lik, b.kl]
            diff = [(i, float((x.double() - y.double()).abs().max())) for i, (x, y) in enumerate(zip(xs, ys)) if not torch.equal(x, y)]
            if diff and first is None:
                first = (blk, diff)
        if first:
            bad += 1
            print("rep", rep, (S, M, N, P), "first difference in block", first[0], first[1][:8], flush=True)
print("repetitions", reps, "shape-runs that differed:", bad)
