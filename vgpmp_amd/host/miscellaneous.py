"""Orchestration of one planning problem behind the reference's names (utils/miscellaneous.py).

The module is star-imported by the reference's driver, which then uses `gpflow`, `np`, `p`, `time`,
`get_root_package_path` and `solve_planning_problem` (benchmarking.py:3-97), so those names are bound
here on purpose.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from .environment import get_root_package_path
from .shims import gpflow, p, set_trainable


def init_trainset(grid_spacing_X, grid_spacing_Xnew, input_dimension, degree_of_freedom, start_joints, end_joints,
                  scale=100, end_time=1):
    """utils/miscellaneous.py:115-127: time grids X / Xnew (rows t * 1_D) and y = [start; end]."""
    X = np.array([np.full(input_dimension, t) for t in np.linspace(0, end_time * scale, grid_spacing_X)])
    Xnew = np.array([np.full(input_dimension, t) for t in np.linspace(0, end_time * scale, grid_spacing_Xnew)])
    y = np.concatenate([np.asarray(start_joints, dtype=np.float64).reshape(1, degree_of_freedom),
                        np.asarray(end_joints, dtype=np.float64).reshape(1, degree_of_freedom)], axis=0)
    return X, y, Xnew


def disable_param_opt(planner, trainable_params):
    """utils/miscellaneous.py:324-343: apply the `trainable_params` flags of parameters.yaml.
    Trainable inducing locations (utils/miscellaneous.py:338) become per-problem variables of the device batch with the
    Sigmoid(0.09, 0.91) bijector of models/vgpmp.py:29-42 (vgpmp.h: vgpmp_inducing_params);
    sigma_obs / alpha become per-problem variables of the device batch (vgpmp.h: vgpmp_lik_params);
    the Normal priors the reference attaches here are centred on the parameters themselves, so they contribute only
    the bijectors' log-det-Jacobians to the loss -- that term is part of the device update."""
    for kern in planner.kernel.kernels:
        set_trainable(kern.variance, trainable_params["kernel_variance"])
        set_trainable(kern.lengthscales, trainable_params["lengthscales"])
    set_trainable(planner.alpha, bool(trainable_params.get("alpha", False)))
    set_trainable(planner.likelihood.variance, bool(trainable_params.get("sigma_obs", False)))
    planner.trainable = {"q_mu": bool(trainable_params["q_mu"]), "q_sqrt": bool(trainable_params["q_sqrt"]),
                         "lengthscales": bool(trainable_params["lengthscales"]),
                         "kernel_variance": bool(trainable_params["kernel_variance"]),
                         "sigma_obs": bool(trainable_params.get("sigma_obs", False)),
                         "alpha": bool(trainable_params.get("alpha", False)),
                         "inducing_variable": bool(trainable_params.get("inducing_variable", False))}
    if planner._planner is not None:
        want_lik = planner.trainable["sigma_obs"] or planner.trainable["alpha"]
        if want_lik != planner._planner.lik_variables or planner.trainable["inducing_variable"] != planner._planner.z_variables:
            planner._planner = None          # the device batch is rebuilt with / without those variables
        else:
            planner._planner.trainable = dict(planner.trainable)


def optimization_step(model, closure=None, optimizer=None, data=None):
    """utils/miscellaneous.py:68-84: one Adam step on -ELBO; returns the loss of that step's paths."""
    pl = model._ensure(model._n_train if data is None else model._check_data(data))
    pl.step()
    model.optimizer.iterations = pl.t
    return float(-(pl.lik - pl.kl)[0])


def training_loop(model, data, num_steps, print_summary=False, randomize=False):
    """utils/miscellaneous.py:87-112."""
    print("Starting training....")
    model.optimization_steps(data, int(num_steps))
    if print_summary:
        for kern in model.kernel.kernels:
            print(f"model lengthscale: {kern.lengthscales.numpy()} \\nmodel variance: {kern.variance.numpy()}")


def solve_planning_problem(env, start_joints, end_joints, run=0, k=0):
    """utils/miscellaneous.py:141-321 without the GUI branches: build the model for this start-goal
    pair, optimise, draw 150 posterior paths, keep the most likely one and check it.
    Returns (solved, best_sample[Nnew, D])."""
    from .model import VGPMP
    planner_params = env.config["planner_params"]
    trainable_params = env.config["trainable_params"]
    dof = env.robot.dof
    X, y, Xnew = init_trainset(planner_params["time_spacing_X"], planner_params["time_spacing_Xnew"], dof, dof,
                               start_joints, end_joints, scale=1)
    planner = VGPMP.initialize(sdf=env.sdf, robot=env.robot, sampler=env.sampler, query_states=y,
                               scene_offset=env.scene.position, q_mu=None, interpolation_method="linear",
                               **planner_params)
    disable_param_opt(planner, trainable_params)
    env.robot.set_current_joint_config(np.squeeze(start_joints))
    training_loop(model=planner, num_steps=planner_params["num_steps"], data=X)
    sample_mean, best_sample, samples, uncertainties = planner.sample_from_posterior(Xnew, env.robot)
    env.robot.set_current_joint_config(np.squeeze(start_joints))
    env.robot.enable_collision_active_links(-1)
    pl = planner._planner
    import torch
    env.robot.clearance_fn = lambda path: float(pl.path_clearance(torch.as_tensor(path, device=pl.device)).min())
    res = env.robot.move_to_ee_config(best_sample)
    return res, best_sample


def solve_planning_problems_batched(env, queries, seed: int = 0, report: dict = None):
    """All start-goal queries of a problem set optimised in lock step as ONE device batch (the reference
    loops over them one by one, benchmarking.py:70-85).  Returns [(solved, best_sample[Nnew, D]), ...].
    `report` (a dict) receives per query the signed clearances of the start state, the goal state, the initial straight line
    and the best sample: the headless success check (clearance > 0 everywhere, utils/robot.py:455-480 without the physics)
    cannot pass for a query whose own end states touch the obstacles by the sphere model, whatever the planner does."""
    import torch
    from .. import engine
    from .model import VariationalMonteCarloLikelihood
    pp = env.config["planner_params"]
    tp = env.config["trainable_params"]
    dof = env.robot.dof
    lik = VariationalMonteCarloLikelihood(sigma_obs=pp["sigma_obs"], robot=env.robot, sampler=env.sampler, sdf=env.sdf,
                                          offset=env.scene.position, epsilon=pp["epsilon"])
    qs = np.array([[np.asarray(a, dtype=np.float64).reshape(dof), np.asarray(b, dtype=np.float64).reshape(dof)]
                   for a, b in queries])
    pl = engine.PlannerBatch(lik.device_scene, qs, num_samples=pp["num_samples"], num_inducing=pp["num_inducing"],
                             num_data=pp["time_spacing_X"], lengthscales=pp["lengthscales"], variance=pp["variance"],
                             alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=seed,
                             trainable={"q_mu": bool(tp["q_mu"]), "q_sqrt": bool(tp["q_sqrt"]),
                                        "lengthscales": bool(tp["lengthscales"]),
                                        "kernel_variance": bool(tp["kernel_variance"]),
                                        "sigma_obs": bool(tp.get("sigma_obs", False)), "alpha": bool(tp.get("alpha", False)),
                                        "inducing_variable": bool(tp.get("inducing_variable", False))})
    if report is not None:
        c0, c1, cl = pl.query_clearances(int(pp["time_spacing_Xnew"]))
    pl.run_steps(int(pp["num_steps"]))
    Xnew = np.tile(np.linspace(0.0, 1.0, int(pp["time_spacing_Xnew"]))[:, None], (1, dof))
    _, best, _, _ = pl.sample_from_posterior(150, Xnew, step=pl.t)
    clear = pl.path_clearance(best).amin(dim=(1, 2)).cpu().numpy()
    if report is not None:
        report.update(start=c0.cpu().numpy().astype(float).tolist(), goal=c1.cpu().numpy().astype(float).tolist(),
                      initial_path=cl.cpu().numpy().astype(float).tolist(), best_sample=clear.astype(float).tolist())
    best = best.cpu().numpy().astype(np.float64)
    low, high = env.robot.spec.low, env.robot.spec.high
    inside = ((best >= low - 1e-9) & (best <= high + 1e-9)).all(axis=(1, 2))
    return [(bool(clear[i] > 0.0 and inside[i]), best[i]) for i in range(len(queries))]
