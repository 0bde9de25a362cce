"""Shared builders for the tests: oracle objects from the product's robot tables, synthetic
scenes, small problem instances.  (The oracle is the checker; see oracle/vgpmp_oracle.py.)"""
import math

import numpy as np

from oracle import vgpmp_oracle as orc
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes


def oracle_robot(spec: rb.RobotSpec) -> orc.RobotTable:
    return orc.RobotTable(name=spec.name, dh=spec.dh.copy(), twist=spec.twist.copy(), craig=spec.craig,
                          base_pose=spec.base_pose.copy(), fk_slice=spec.fk_slice.copy(),
                          spheres_per_link=spec.num_spheres_per_link.copy(),
                          sphere_offsets=spec.sphere_offsets.copy(), radii=spec.sphere_radii.copy(),
                          joint_limits=spec.joint_limits.copy())


def oracle_scene(spec, grid, offset, sigma_obs=0.005, epsilon=0.05) -> orc.Scene:
    data, origin, delta = grid
    return orc.Scene(robot=oracle_robot(spec), sdf=orc.SDFGrid(np.asarray(data, np.float64), np.asarray(origin, np.float64), float(delta)),
                     offset=np.asarray(offset, dtype=np.float64),
                     sigma_obs=np.full(spec.num_spheres, sigma_obs), epsilon=epsilon)


def small_problem(robot="franka", S=6, N=9, M=5, B=64, seed=0, n_grid=24, problem="industrial"):
    """A tiny planning instance with obstacles close enough that the hinge is active."""
    ps = rb.load_problemset(robot, problem)
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=n_grid, delta=2.4 / n_grid, origin=(-1.2, -1.2, -0.6), seed=seed)
    scene = oracle_scene(spec, grid, ps.object_positions[0])
    y = np.array([ps.states[0], ps.states[1]], dtype=np.float64)
    D = spec.dof
    pp = ps.planner_params
    params = orc.init_params(scene.robot, y, M, pp["lengthscales"], pp["variance"])
    rng = np.random.default_rng(seed + 1)
    params.q_sqrt = np.tril(params.q_sqrt + 0.05 * rng.standard_normal(params.q_sqrt.shape))
    params.q_mu = params.q_mu + 0.05 * rng.standard_normal(params.q_mu.shape)
    X = orc.init_trainset(N, D)
    Zy = orc.inducing_Zy(M, D)
    noise = orc.draw_noise(rng, S, D, D, B, M + 2)
    return dict(spec=spec, scene=scene, grid=grid, y=y, params=params, X=X, Zy=Zy, noise=noise,
                alpha=float(pp["alpha"]), lr=float(pp["learning_rate"]), offset=np.array(ps.object_positions[0]))


def synthetic_problem(dof=14, S=8, N=12, M=6, B=64, seed=0, n_grid=48, n_problems=1):
    """BASELINE config 5's robot (synthetic classic-DH chain, 3 spheres per frame incl. the base: P = 45 at 14 joints) on a
    small obstacle grid, random start-goal pairs in +-2 rad; `n_problems` instances sharing params and noise."""
    spec = rb.synthetic_arm(dof)
    grid = scenes.synthetic_boxes_sdf(n=n_grid, delta=2.4 / n_grid, origin=(-1.2, -1.2, -0.6), seed=seed, n_boxes=10, n_spheres=6)
    offset = np.array([0.05, -0.03, 0.02])
    scene = oracle_scene(spec, grid, offset)
    rng = np.random.default_rng(seed + 1)
    ys = rng.uniform(-2.0, 2.0, (n_problems, 2, dof))
    params = []
    for y in ys:
        p = orc.init_params(scene.robot, y, M, [2.0] * dof, 0.2)
        p.q_sqrt = np.tril(p.q_sqrt + 0.05 * rng.standard_normal(p.q_sqrt.shape))
        p.q_mu = p.q_mu + 0.05 * rng.standard_normal(p.q_mu.shape)
        params.append(p)
    X = orc.init_trainset(N, dof)
    Zy = orc.inducing_Zy(M, dof)
    noise = [orc.draw_noise(rng, S, dof, dof, B, M + 2) for _ in range(n_problems)]
    return dict(spec=spec, scene=scene, grid=grid, ys=ys, params=params, X=X, Zy=Zy, noise=noise, alpha=100.0, lr=0.02,
                offset=offset)


# ---- end-to-end comparisons on the device's own voxels (tests/test_gpu_parity.py, test_gpu_config5.py) ----------------
# Fixed tolerances of the end-to-end comparisons on the device's own voxels (float32 path against float64 oracle).
# Measured on MI355X (gpurun_out/r05/t_parity.txt, "PARITY" lines): logp <= 3e-6, lik <= 2e-6, gradients <= 4e-4 of the largest
# component at config 2's full size; the bounds below are <= 10x that.
TOL_LOGP = 2e-5       # per (sample, time) pair, relative to the largest |logp|
TOL_LIK = 2e-5        # alpha / S * sum logp, relative
TOL_GRAD = 3e-3       # every gradient component, relative to the largest component of its tensor
MAX_FLIPPED = 0.03    # share of (sample, time) pairs a float64 chain resolves to another voxel (coarse 0.05 m test grids)


def device_centres(pl, k):
    """float64 copy of the float32 sphere centres [S, N, Q, 3] the likelihood launch of the last evaluation formed for problem k."""
    return pl.sphere_centres()[k].cpu().numpy().astype(np.float64)


def flipped_share(tag, logp_dev, fw64):
    """Share of (sample, time) pairs whose log-density differs from the oracle's when the oracle looks its voxels up at its OWN
    float64 centres: pairs with a sphere within float32 rounding of a cell boundary.  Printed, bounded, and nothing else."""
    ok = np.isclose(logp_dev, fw64["logp"], rtol=2e-3, atol=1e-4)
    flipped = 1.0 - ok.mean()
    print(f"PARITY {tag} flipped_share={flipped:.5f}")
    assert flipped <= MAX_FLIPPED, (tag, flipped)
    return flipped


def assert_grads(tag, got_list, og, names=("q_mu", "q_sqrt", "raw_ell", "raw_var"), k=0):
    worst = {}
    for got, name in zip(got_list, names):
        want = getattr(og, name)
        got = got[k].cpu().numpy()
        if name == "q_mu":
            got = got.T
        scale = np.abs(want).max() + 1e-12
        worst[name] = np.abs(got - want).max() / scale
    print("PARITY", tag, " ".join(f"{n}={v:.2e}" for n, v in worst.items()))
    for name, v in worst.items():
        assert v < TOL_GRAD, (tag, name, v)
