"""Validation aid: where two schedules of the same planner first differ.  Runs the shared-launch schedule and one launch per
kernel side by side, `chunk` steps at a time, and reports the first step count at which any parameter / moment differs, with
the differing tensors.      python tools/schedule_first_difference.py <problems> [chunk] [max_steps]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgpmp_amd import engine, robots as rb, scenes

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
ps = rb.load_problemset("franka", "industrial")
spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=64, delta=0.025, origin=(-0.8, -0.8, -0.2), seed=0)
pp = ps.planner_params
sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
qs = np.array([ps.queries[i] for i in range(P)])
kw = dict(num_samples=128, num_inducing=30, num_data=100, num_bases=1024, lengthscales=pp["lengthscales"],
          variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=3)
a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
b.fuse = False
# (one launch per kernel forms the prior draws by the f16-split kernel from SK = 1 up: bitwise comparison needs its float32 form)
from vgpmp_amd import capi
b.extra_flags |= capi.PRIOR_F32
names = ["raw_ell", "raw_var", "q_mu", "q_sqrt"]
for t in range(0, max_steps, chunk):
    a.run_steps(chunk); b.run_steps(chunk)
    torch.cuda.synchronize()
    diffs = {}
    for n in names:
        x, y = getattr(a, n), getattr(b, n)
        if not torch.equal(x, y):
            d = (x - y).abs()
            diffs[n] = (float(d.max()), [int(v) for v in torch.nonzero(d.reshape(P, -1).amax(1) > 0).flatten()])
    for k in range(len(a.adam_m)):
        if not torch.equal(a.adam_m[k], b.adam_m[k]) or not torch.equal(a.adam_v[k], b.adam_v[k]):
            diffs[f"adam[{k}]"] = (float((a.adam_v[k] - b.adam_v[k]).abs().max()), [])
    if diffs:
        print(f"P={P}: first difference after {t + chunk} steps:", diffs)
        print("  finite:", bool(torch.isfinite(a.q_mu).all()), bool(torch.isfinite(b.q_mu).all()),
              " lengthscales a", a.lengthscales().flatten()[:7].tolist(), " b", b.lengthscales().flatten()[:7].tolist())
        break
else:
    print(f"P={P}: {max_steps} steps identical")
