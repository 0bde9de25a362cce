import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from vgpmp_amd import engine, robots as rb, scenes
ps = rb.load_problemset("franka", "industrial")
spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
qs = np.array([ps.queries[0]])
kw = dict(num_samples=32, num_inducing=30, num_data=40, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=3)
for steps in (1, 2, 3, 25):
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.fuse = False
    a.run_steps(steps)
    for _ in range(steps):
        b.run_steps(1)
    torch.cuda.synchronize()
    print("steps", steps, {n: float((getattr(a, n).double() - getattr(b, n).double()).abs().max()) for n in ("q_mu", "q_sqrt", "raw_ell", "w", "eps", "eps2", "omega", "beta", "f", "logp", "lik")})
