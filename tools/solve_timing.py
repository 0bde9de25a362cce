"""Measurement aid: wall time of the reference-shaped solve_planning_problem() flow per query (config 2)."""
import sys, os, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ns = {}
exec("from gpflow_vgpmp.utils.miscellaneous import *", ns)
from gpflow_vgpmp.utils.simulation_manager import SimulationManager
import torch
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t0 = time.perf_counter()
    env = SimulationManager(file_path=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parameters.yaml"))
    print(f"SimulationManager: {time.perf_counter() - t0:.2f} s")
solve = ns["solve_planning_problem"]
queries = env.config["scene_params"]["queries"][:6]
for k, (start, end) in enumerate(queries):
    start = np.array(start, dtype=np.float64).reshape(1, env.robot.dof)
    end = np.array(end, dtype=np.float64).reshape(1, env.robot.dof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solved, traj = solve(env=env, start_joints=start, end_joints=end)
    torch.cuda.synchronize()
    print(f"query {k}: {1e3 * (time.perf_counter() - t0):.1f} ms solved={solved}")

# the whole problem set as ONE device batch (what benchmarking.py's loop becomes on the HIP path)
batched = ns["solve_planning_problems_batched"]
qs = env.config["scene_params"]["queries"]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = batched(env, qs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batched: {len(qs)} queries in {1e3 * dt:.1f} ms = {len(qs) / dt:.0f} plans/s, solved {sum(int(r[0]) for r in res)}/{len(qs)}")
print("batched flags :", "".join("T" if r[0] else "F" for r in res))
single = []
for (start, end) in qs[:12]:
    s_, _ = solve(env=env, start_joints=np.array(start, dtype=np.float64).reshape(1, env.robot.dof),
                  end_joints=np.array(end, dtype=np.float64).reshape(1, env.robot.dof))
    single.append(s_)
print("single  flags :", "".join("T" if x else "F" for x in single))
