"""Measurement aid: forward-only ELBO evaluations in a loop (fewer distinct kernels than a training step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vgpmp_amd import engine, robots, scenes
ps = robots.load_problemset("franka", "industrial"); spec = robots.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=128, delta=0.0125, origin=(-0.8, -0.8, -0.2), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
pl = engine.PlannerBatch(sc, np.array([ps.queries[0]]), num_samples=128, num_inducing=30, num_data=100, lengthscales=[2.0]*7, variance=0.2)
pl.generate_noise(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for i in range(300):
    if mode == "fwd": pl.elbo(generate=False)
    else: pl.step()
torch.cuda.synchronize()
