"""Are the plans plans?    Runs on the GPU box: `python tests/plan_report.py [robot] [problem set]`.

For every start-goal query of a problem set, at the reference's own planner parameters (data/problemsets/<robot>.py), it reports
the signed clearance (SDF distance minus sphere radius, minimum over spheres and time; > 0 = collision-free, the headless stand-in
for utils/robot.py:455-480) of
  * the start and the goal state themselves (a query whose end points collide can never be "solved"),
  * the initial path (the straight line in joint space the variational mean starts from, models/vgpmp.py:166-171),
  * the posterior mean and the best of 150 posterior samples after num_steps optimisation steps on the device,
and, for the first `--oracle` queries, the same two figures from the float64 oracle driven by the SAME Philox noise stream.
Test infrastructure (it imports the oracle); tests/test_gpu_plans.py asserts on the same quantities.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def device_report(robot="franka", problem="industrial", overrides=None, seed=0, n_init=100):
    import torch
    from vgpmp_amd import engine, robots, scenes
    ps = robots.load_problemset(robot, problem)
    pp = dict(ps.planner_params, **(overrides or {}))
    spec = robots.load_robot(robot, *ps.robot_pos_and_orn)
    grid = scenes.scene_sdf(problem, delta=0.0125, padding=20)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    queries = ps.queries
    qs = np.array([[a, b] for a, b in queries], dtype=np.float64)
    pl = engine.PlannerBatch(sc, qs, num_samples=pp["num_samples"], num_inducing=pp["num_inducing"], num_data=pp["time_spacing_X"],
                             lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                             learning_rate=pp["learning_rate"], seed=seed)
    dev = pl.device
    # clearance of the pinned states and of the straight line between them
    states = torch.tensor(np.asarray(ps.states), dtype=torch.float32, device=dev)
    st_clear = pl.path_clearance(states[None])[0]                                   # [n_states, P]
    lam = torch.linspace(0.0, 1.0, n_init, device=dev, dtype=torch.float64)[None, :, None]
    q = torch.tensor(qs, device=dev)
    line = (q[:, :1] + (q[:, 1:] - q[:, :1]) * lam).to(torch.float32)               # [Q, n_init, L]
    init_clear = pl.path_clearance(line).amin(dim=2)                                # [Q, n_init]
    loss0 = (-(pl.elbo(step=10**6))).cpu().numpy()
    pl.run_steps(int(pp["num_steps"]))
    loss1 = (-(pl.elbo(step=10**6))).cpu().numpy()
    Xnew = np.tile(np.linspace(0.0, 1.0, int(pp["time_spacing_Xnew"]))[:, None], (1, spec.dof))
    mean, best, _, _ = pl.sample_from_posterior(150, Xnew, step=pl.t)
    cm = pl.path_clearance(mean)                                                    # [Q, Nnew, P]
    cb = pl.path_clearance(best)
    torch.cuda.synchronize()
    st_min = st_clear.amin(dim=1).cpu().numpy()
    st_arg = st_clear.argmin(dim=1).cpu().numpy()
    idx = {tuple(np.round(s, 9)): i for i, s in enumerate(np.asarray(ps.states))}
    rows = []
    for k, (a, b) in enumerate(queries):
        ia, ib = idx[tuple(np.round(a, 9))], idx[tuple(np.round(b, 9))]
        flat = cb[k].reshape(-1)
        w = int(flat.argmin())
        rows.append(dict(query=k, start_state=ia, goal_state=ib, start=float(st_min[ia]), goal=float(st_min[ib]),
                         initial_path=float(init_clear[k].min()), mean_path=float(cm[k].min()), best_sample=float(cb[k].min()),
                         worst_time=w // spec.num_spheres, worst_sphere=w % spec.num_spheres,
                         loss_first=float(loss0[k]), loss_last=float(loss1[k])))
    states_rep = [dict(state=i, clearance=float(st_min[i]), sphere=int(st_arg[i])) for i in range(len(ps.states))]
    return dict(robot=robot, problem=problem, planner_params=pp, states=states_rep, queries=rows), (ps, spec, grid, pp, pl)


def oracle_plan(ps, spec, grid, pp, query_index, seed=0, clearance_fn=None):
    """The float64 oracle on query `query_index`, noise = the device's Philox stream of (seed, problem = query_index, step)."""
    from helpers import oracle_scene
    from oracle import vgpmp_oracle as orc
    sc = oracle_scene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    y = np.array(ps.queries[query_index], dtype=np.float64)
    S, N, M, B, D = int(pp["num_samples"]), int(pp["time_spacing_X"]), int(pp["num_inducing"]), 1024, spec.dof
    p = orc.init_params(sc.robot, y, M, pp["lengthscales"], pp["variance"])
    st = orc.adam_init(p)
    X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
    losses = []
    for t in range(int(pp["num_steps"])):
        nz = orc.philox_noise(seed, query_index, t, S, D, D, B, M + 2)
        losses.append(orc.optimization_step(p, st, sc, X, Zy, y, nz, float(pp["alpha"]), float(pp["learning_rate"])))
    Xnew = orc.init_trainset(int(pp["time_spacing_Xnew"]), D)
    mean = orc.posterior_mean(p, sc.robot, Xnew, Zy, y)
    pos = orc.sphere_positions(sc.robot, mean)                                     # [Nnew, P, 3]
    d = orc.sdf_distance(sc.sdf, pos - sc.offset) - np.asarray(spec.sphere_radii)[None, :]
    return dict(query=query_index, mean_path=float(d.min()), loss_first=float(losses[0]), loss_last=float(np.mean(losses[-10:])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("robot", nargs="?", default="franka")
    ap.add_argument("problem", nargs="?", default="industrial")
    ap.add_argument("--oracle", type=int, default=3)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    rep, (ps, spec, grid, pp, pl) = device_report(a.robot, a.problem)
    print(f"{a.robot} / {a.problem}: planner parameters {rep['planner_params']}")
    print("states (clearance of the state itself, worst sphere):")
    for s in rep["states"]:
        print(f"  state {s['state']}: {s['clearance']:+.4f} m (sphere {s['sphere']})")
    print("query  start   goal    | initial path | mean path  best sample (worst t, sphere) | loss first -> last")
    for r in rep["queries"]:
        print(f"{r['query']:3d}   {r['start']:+.3f}  {r['goal']:+.3f}  |   {r['initial_path']:+.4f}    |  {r['mean_path']:+.4f}    {r['best_sample']:+.4f}"
              f"   ({r['worst_time']:3d}, {r['worst_sphere']:2d})      | {r['loss_first']:.4g} -> {r['loss_last']:.4g}")
    q = rep["queries"]
    free = [r for r in q if r["start"] > 0 and r["goal"] > 0]
    print(f"queries with collision-free end points: {len(free)} of {len(q)}; solved (best sample clear): {sum(r['best_sample'] > 0 for r in q)}"
          f" of {len(q)}, {sum(r['best_sample'] > 0 for r in free)} of those with free end points; initial straight line clear: "
          f"{sum(r['initial_path'] > 0 for r in q)}; improved or kept (best >= initial - 1 mm): {sum(r['best_sample'] >= r['initial_path'] - 1e-3 for r in q)}")
    rep["oracle"] = []
    for k in range(a.oracle):
        o = oracle_plan(ps, spec, grid, pp, k)
        rep["oracle"].append(o)
        print(f"oracle query {k}: mean path clearance {o['mean_path']:+.4f} (device {q[k]['mean_path']:+.4f}); loss {o['loss_first']:.4g} -> {o['loss_last']:.4g}")
    if a.json:
        json.dump(rep, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
