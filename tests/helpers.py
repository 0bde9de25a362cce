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
# Measured on MI355X over every comparison of the GPU suite (gpurun_out/r05/t_all.txt, "PARITY" lines: 11 problem sets, BASELINE
# configs 1-4 at full size, the 14-joint arm, trainable inducing locations): the bounds below are within 10x of the worst case.
TOL_LOGP = 2e-6       # per (sample, time) pair, relative to the largest |logp|                   (measured: <= 1.7e-7)
TOL_LIK = 2e-6        # alpha / S * sum logp, relative                                             (measured: <= 2.2e-7)
TOL_GRAD = 3e-4       # every gradient component, relative to the largest component of its tensor  (measured: <= 5.8e-5)
MAX_FLIPPED = 5e-3    # share of (sample, time) pairs a float64 chain resolves to another voxel (measured: <= 5e-4; r05 allowed 0.03).
                      # The one assertion that notices a wrong FK chain INSIDE the ELBO launch: the logp comparison itself looks
                      # its voxels up at the device's own centres.


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


def follow_device_trajectory(tag, pl, osc, qs, pp, var, steps, seed, base, problems=None):
    """`steps` optimisation steps of the device planner `pl` on its OWN generated noise (seed, problem, step -> Philox), the oracle
    following step by step: before every step the device's variables and Adam moments are read back and handed to the oracle,
    which then takes the same step on orc.philox_noise of the same key with its voxels looked up at the device's own sphere
    centres of that step.  So every step is compared on identical inputs -- loss, every gradient, every updated variable and
    moment -- with the fixed tolerances of the one-step tests; the chaotic growth of a free-running pair of trajectories (one
    sphere centre resolving to a neighbouring voxel changes the path of everything after it) never enters.  The device itself
    runs free: its state is never overwritten.  Returns the device's per-step losses [steps, P]."""
    import torch
    P, L, M, S, N, B = pl.P, pl.L, pl.M, pl.S, pl.N, pl.B
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    lr, alpha = float(pp["learning_rate"]), float(pp["alpha"])
    problems = range(P) if problems is None else problems
    names = ("q_mu", "q_sqrt", "raw_ell", "raw_var")
    dev_loss = np.zeros((steps, P))
    worst = dict(loss=0.0, grad=0.0, var=0.0, m=0.0)

    def read(tensors, k):
        out = [t[k].cpu().numpy().copy() for t in tensors]
        out[0] = out[0].T.copy()                                   # q_mu: [L, M] on the device, [M, L] in the oracle
        return orc.Params(*out)

    for t in range(steps):
        before = {k: (read([pl.q_mu, pl.q_sqrt, pl.raw_ell, pl.raw_var], k), read(pl.adam_m, k), read(pl.adam_v, k)) for k in problems}
        pl.step()
        dev_loss[t] = (-(pl.lik - pl.kl)).cpu().numpy()
        for k in problems:
            p, m, v = before[k]
            st = orc.AdamState(m, v, t)
            nz = orc.philox_noise(seed, base + k, t, S, L, L, B, M + 2)
            want, g = orc.optimization_step(p, st, osc, X, Zy, qs[k], nz, alpha, lr, lookup_pos=device_centres(pl, k), want_grad=True)
            rel = abs(dev_loss[t, k] - want) / abs(want)
            worst["loss"] = max(worst["loss"], rel)
            assert rel <= 5e-7, (tag, "loss", k, t, dev_loss[t, k], want, rel)       # (measured: <= 7e-8)
            after, m_dev = read([pl.q_mu, pl.q_sqrt, pl.raw_ell, pl.raw_var], k), read(pl.adam_m, k)
            g_dev = read(pl.grad, k)
            for name in names:
                gw, gd = getattr(g, name), getattr(g_dev, name)
                top = np.abs(gw).max() + 1e-300
                e = np.abs(gd - gw).max() / top
                worst["grad"] = max(worst["grad"], e)
                assert e < TOL_GRAD, (tag, "gradient", name, k, t, e)
                # the first moment is linear in the gradient: same relative accuracy
                e = np.abs(getattr(m_dev, name) - getattr(st.m, name)).max() / (np.abs(getattr(st.m, name)).max() + 1e-300)
                worst["m"] = max(worst["m"], e)
                assert e < TOL_GRAD, (tag, "first moment", name, k, t, e)
                # Adam normalises every entry (lr m / (sqrt v + 1e-7)): an entry whose gradient is rounding noise may move by up to
                # lr either way, so the updated variables are compared where the gradient carries information
                big = np.abs(gw) >= 1e-3 * top
                e = np.abs(getattr(after, name) - getattr(p, name))[big].max() / lr if big.any() else 0.0
                worst["var"] = max(worst["var"], e)
                assert e < 2e-4, (tag, "updated variable (in units of lr)", name, k, t, e)      # (measured: <= 1.7e-5)
    print(f"PARITY {tag} follow: " + " ".join(f"{a}={b:.2e}" for a, b in worst.items()))
    assert np.isfinite(dev_loss).all()
    return dev_loss
