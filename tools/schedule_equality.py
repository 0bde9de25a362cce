"""Validation aid: thousands of steps of the small-batch schedule (shared launches, hyper-parameter update as a
stage-1 prologue, q update riding with the next step) against one launch per kernel -- the parameters must agree
to the last bit."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgpmp_amd import capi, engine, robots as rb, scenes

ps = rb.load_problemset("franka", "industrial")
spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=64, delta=0.025, origin=(-0.8, -0.8, -0.2), seed=0)
pp = ps.planner_params
sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
CASES = ((1, 3000), (4, 1500), (8, 800))
if len(sys.argv) > 2:      # tools/schedule_equality.py <problems> <steps>
    CASES = ((int(sys.argv[1]), int(sys.argv[2])),)
for P, steps in CASES:
    qs = np.array([ps.queries[i] for i in range(P)])
    kw = dict(num_samples=128, num_inducing=30, num_data=100, num_bases=1024, lengthscales=pp["lengthscales"],
              variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=3)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.fuse = False
    # (one launch per kernel forms the prior draws by the f16-split kernel at SK = 1: the bitwise comparison needs its float32 form)
    b.extra_flags |= capi.PRIOR_F32
    if P > 4:
        a.extra_flags |= capi.PRIOR_F32      # (the large-batch schedule forms them by the f16-split kernel as well)
    worst = 0.0
    for blk in range(steps // 100):
        a.run_steps(100 if blk % 2 else 37)
        a.run_steps(0 if blk % 2 else 63)
        b.run_steps(100)
        torch.cuda.synchronize()
        d = max(float((x - y).abs().max()) for x, y in ((a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var), (a.q_mu, b.q_mu),
                                                        (a.adam_v[2], b.adam_v[2])))
        worst = max(worst, d)
        if d > 0:
            print(f"P={P}: mismatch {d:.3e} after {100 * (blk + 1)} steps")
            break
    print(f"P={P}: {steps} steps, worst |difference| between schedules {worst:.3e}")
