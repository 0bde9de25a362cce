"""Every problem set the reference ships, at ITS OWN planner parameters, through the device ELBO step against the float64 oracle
(VERDICT r4 item 1c).  The eleven (robot, scene) sets of data/problemsets/{franka,wam,ur10,kuka}.py -- re-entered in
vgpmp_amd/data/problemsets.json -- differ in what the covariance path and the prior kernels see: Kuka trains M = 7 inducing points
(Mz = 9, the smallest float64 tiles: data/problemsets/kuka.py:75-104), wam / lab asks for a kernel variance of 0.05, BELOW the
positive(lower=0.1) floor of models/vgpmp.py:139 (data/problemsets/wam.py:113; GPflow would refuse it, this build lifts it to the
floor + 1e-6), the `boxes` sets have two states, i.e. ONE query (a one-problem batch at few samples), UR10 has
six joints and sigma_obs down to 1e-4 (and, by the reference's own base pose diag(-1, -1, 1) -- tests/test_robot.py:70-73 -- and DH
table, faces AWAY from its bookshelf: every state of ur10 / bookshelves is 0.5 m clear, so that set exercises the KL path only).  Per set: min(4, #queries) queries in one batch, on the set's own scene (SDF generated from
the reference's collision mesh at the set's object position);
  (a) one step with injected noise: log-density of every (sample, time) pair, ELBO pieces and every gradient against the oracle on
      the device's own voxels (tests/helpers.py: fixed tolerances, no allowance for neighbouring cells);
  (b) ten optimisation steps with the device's own generated noise, the oracle following step by step on orc.philox_noise of the
      same key from the device's own state and voxels: loss, gradients, moments and updated variables of EVERY step at the
      one-step tolerances; plus a sanity bound on the free-running pair.
"""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
from helpers import TOL_LIK, TOL_LOGP, assert_grads, device_centres, flipped_share, follow_device_trajectory, oracle_scene
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes

pytestmark = pytest.mark.gpu

SETS = [("franka", "industrial"), ("franka", "bookshelves"), ("franka", "boxes"), ("wam", "industrial"), ("wam", "bookshelves"),
        ("wam", "lab"), ("ur10", "industrial"), ("ur10", "bookshelves"), ("kuka", "industrial"), ("kuka", "bookshelves"),
        ("kuka", "boxes")]
VARIANCE_FLOOR = 0.1
_grids = {}


def _scene_grid(name):
    if name not in _grids:
        _grids[name] = scenes.scene_sdf(name, delta=0.02, padding=12)
    return _grids[name]


def _setup(robot, problem):
    from vgpmp_amd import engine
    ps = rb.load_problemset(robot, problem)
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    pp = ps.planner_params
    grid = _scene_grid(problem)
    off = ps.object_positions[0]
    sc = engine.DeviceScene(spec, grid, off, sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    osc = oracle_scene(spec, grid, off, sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    S, M, N = int(pp["num_samples"]), int(pp["num_inducing"]), int(pp["time_spacing_X"])
    # the queries of the set whose initial straight line comes closest to (or deepest into) the obstacles: the ones that exercise
    # the likelihood; plus whether ANY of them reaches the hinge band at all (a set may keep clear of its scene from the start)
    allq = np.array(ps.queries, dtype=np.float64)
    probe = engine.PlannerBatch(sc, allq, num_samples=1, num_inducing=M, num_data=N, num_bases=16, lengthscales=pp["lengthscales"],
                                variance=pp["variance"])
    line = probe.query_clearances(N)[2].cpu().numpy()
    del probe
    nq = min(4, len(allq))
    qs = allq[np.argsort(line)[:nq]]
    reaches_hinge = bool(line.min() < float(pp["epsilon"]) - 0.01)
    var = max(float(pp["variance"]), VARIANCE_FLOOR + 1e-6)         # wam / lab: 0.05 is below the floor
    return engine, ps, spec, pp, sc, osc, qs, S, M, N, var, reaches_hinge


@pytest.mark.parametrize("robot,problem", SETS)
def test_injected_noise_step_against_oracle(robot, problem):
    engine, ps, spec, pp, sc, osc, qs, S, M, N, var, reaches_hinge = _setup(robot, problem)
    P, L, B = len(qs), spec.dof, 1024
    assert len(ps.queries) == len(ps.states) * (len(ps.states) - 1) // 2
    pl = engine.PlannerBatch(sc, qs, num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=pp["lengthscales"],
                             variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"])
    rng = np.random.default_rng(31)
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    params, noises = [], []
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    for k in range(P):
        p = orc.init_params(osc.robot, qs[k], M, pp["lengthscales"], var)
        p.q_sqrt = np.tril(p.q_sqrt + 0.05 * rng.standard_normal(p.q_sqrt.shape))
        p.q_mu = p.q_mu + 0.05 * rng.standard_normal(p.q_mu.shape)
        params.append(p)
        nz = orc.draw_noise(rng, S, L, L, B, M + 2)
        noises.append(orc.Noise(r32(nz.omega), r32(nz.beta), r32(nz.w), r32(nz.eps), r32(nz.eps2)))
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
    # the planner's own initial kernel variables are the oracle's (the floor included)
    np.testing.assert_allclose(pl.raw_var[0].cpu().numpy(), params[0].raw_var, rtol=1e-12)
    st = lambda name: np.stack([getattr(nz, name) for nz in noises])
    pl.set_noise(st("omega"), st("beta"), st("w"), st("eps"), st("eps2"))
    loss, grads = pl.loss_and_grad(generate=False)
    torch.cuda.synchronize()
    Mz = M + 2
    active = 0
    for k in range(P):
        tag = f"{robot}/{problem}[k={k}]"
        fw = orc.elbo_forward(params[k], osc, X, Zy, qs[k], noises[k], float(pp["alpha"]), lookup_pos=device_centres(pl, k))
        og, _ = orc.elbo_backward(params[k], osc, X, Zy, noises[k], float(pp["alpha"]), fw)
        cv = fw["cv"]
        A4 = pl.view("A4").reshape(P, L, N, Mz, 4)[k].cpu().numpy()
        np.testing.assert_allclose(A4[..., 0], cv["A"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pl.view("C").reshape(P, L, Mz, Mz)[k].cpu().numpy(), cv["C"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(float(pl.kl[k]), cv["kl"], rtol=1e-9)
        np.testing.assert_allclose(pl.f[k].cpu().numpy(), fw["f"], rtol=0, atol=1e-4)
        logp = pl.logp[k].cpu().numpy()
        top = np.abs(fw["logp"]).max()
        active += int(top > 0)
        np.testing.assert_allclose(logp, fw["logp"], rtol=0, atol=TOL_LOGP * top + 1e-30)
        np.testing.assert_allclose(float(pl.lik[k]), fw["lik"], rtol=TOL_LIK, atol=1e-12)
        np.testing.assert_allclose(float(loss[k]), -fw["elbo"], rtol=2e-5)
        assert_grads(tag, grads, og, k=k)
        if k == 0:
            flipped_share(tag, logp, orc.elbo_forward(params[k], osc, X, Zy, qs[k], noises[k], float(pp["alpha"]), want_dell=False))
    if reaches_hinge:      # (ur10 / bookshelves and kuka / boxes start clear of their scenes by more than epsilon)
        assert active > 0, "the queries closest to the obstacles must reach into the hinge band"
    print(f"PARITY {robot}/{problem} queries_in_hinge_band={active}/{P}")


@pytest.mark.parametrize("robot,problem", SETS)
def test_generated_noise_trajectory_against_oracle(robot, problem):
    """Ten optimisation steps on the device's own noise, the oracle following step by step from the device's state
    (tests/helpers.py::follow_device_trajectory): loss 5e-7, gradients and first moments 3e-4 of their largest entry, updated
    variables 2e-4 lr -- at every step of every problem; and the free-running oracle (its own state, its own voxels) stays
    within 5e-3 of the device's loss over the ten steps."""
    engine, ps, spec, pp, sc, osc, qs, S, M, N, var, _ = _setup(robot, problem)
    P, L, B, steps = len(qs), spec.dof, 1024, 10
    seed, base = 123, 17
    pl = engine.PlannerBatch(sc, qs, num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=pp["lengthscales"],
                             variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=seed,
                             problem_base=base)
    dev_loss = follow_device_trajectory(f"{robot}/{problem}", pl, osc, qs, pp, var, steps, seed, base)
    # the free-running pair: chaotic in the voxels (sigma_obs down to 1e-4 turns one neighbouring cell into 1e-3 of the loss), so
    # only a sanity bound -- the step-by-step comparison above is the parity statement
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    worst = 0.0
    for k in range(P):
        p = orc.init_params(osc.robot, qs[k], M, pp["lengthscales"], var)
        st = orc.adam_init(p)
        for t in range(steps):
            nz = orc.philox_noise(seed, base + k, t, S, L, L, B, M + 2)
            want = orc.optimization_step(p, st, osc, X, Zy, qs[k], nz, float(pp["alpha"]), float(pp["learning_rate"]))
            worst = max(worst, abs(dev_loss[t, k] - want) / abs(want))
    print(f"PARITY {robot}/{problem} free-running worst loss deviation = {worst:.2e}")
    assert worst <= 5e-3
