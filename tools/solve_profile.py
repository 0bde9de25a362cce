"""Measurement aid: where the host time of one solve_planning_problem() call goes (cProfile, after two warm-up calls)."""
import cProfile, io, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ns = {}
exec("from gpflow_vgpmp.utils.miscellaneous import *", ns)
from gpflow_vgpmp.utils.simulation_manager import SimulationManager
import torch
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    env = SimulationManager(file_path=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parameters.yaml"))
solve = ns["solve_planning_problem"]
qs = env.config["scene_params"]["queries"]
arg = lambda q: dict(env=env, start_joints=np.array(q[0], dtype=np.float64).reshape(1, env.robot.dof),
                     end_joints=np.array(q[1], dtype=np.float64).reshape(1, env.robot.dof))
for q in qs[:2]:
    solve(**arg(q))
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for q in qs[2:6]:
    solve(**arg(q))
torch.cuda.synchronize()
pr.disable()
print(f"4 calls: {1e3 * (time.perf_counter() - t0) / 4:.2f} ms each (under the profiler)")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
