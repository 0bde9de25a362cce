"""Parity of the HIP path (through the C ABI) with the float64 oracle.  Run with -m gpu on MI355X.

Tolerances: integer voxel indices bit-exact on identical inputs; float32 quantities within the tolerance written at
each assertion.  The SDF is nearest-voxel (piecewise constant): a float32 sphere centre within rounding of a cell
boundary may read the neighbouring voxel of what a float64 chain reads.  The end-to-end comparisons therefore hand the
oracle the device's OWN float32 sphere centres for the voxel lookup (include/vgpmp_debug.h, vgpmp_debug_sphere_centres;
oracle `lookup_pos`): both sides read the same voxels, every (sample, time) pair must agree, and the gradient tolerances are
fixed numbers -- no allowance that grows with the number of disagreeing pairs.  How many pairs a float64 chain WOULD
resolve differently is measured, printed and bounded separately (`_flipped_share`).
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
from helpers import (MAX_FLIPPED, TOL_GRAD, TOL_LIK, TOL_LOGP, assert_grads as _assert_grads, device_centres as _device_centres,
                     flipped_share as _flipped_share, oracle_robot, oracle_scene, small_problem)
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _engine():
    from vgpmp_amd import engine
    return engine


def _scene(spec, grid, offset, sigma_obs=0.005, epsilon=0.05):
    return _engine().DeviceScene(spec, grid, offset, sigma_obs=sigma_obs, epsilon=epsilon)


POSES = {"franka": ((0, 0, 0), (0, 0, 0, 1)), "wam": ((0, 0, 0.346), (0, 0, 0, 1)),
         "ur10": ((0, 0, 0), (0, 0, -1, 0)), "kuka": ((0.1, -0.2, 0.3), (0, 0, 0.38268343, 0.92387953))}


@pytest.mark.parametrize("name", sorted(POSES))
def test_fk_matches_reference_golden_and_oracle(name):
    spec = rb.load_robot(name, *POSES[name])
    grid = scenes.synthetic_boxes_sdf(n=8, delta=0.3, origin=(-1.2, -1.2, -1.2))
    sc = _scene(spec, grid, (0, 0, 0))
    z = np.load(GOLD / "fk_reference.npz")
    q = z[f"{name}_q"]
    pos, frames = sc.fk_spheres(torch.tensor(q), want_frames=True)
    # float32 FK vs the reference's own float64 numpy FK (golden): abs 5e-6 m on a ~1 m arm
    np.testing.assert_allclose(frames.cpu().numpy(), z[f"{name}_frames"][:, :, :3, :], atol=5e-6, rtol=0)
    rng = np.random.default_rng(0)
    q2 = rng.uniform(spec.low, spec.high, (257, spec.dof))
    want = orc.sphere_positions(oracle_robot(spec), q2.astype(np.float32).astype(np.float64))
    got = sc.fk_spheres(torch.tensor(q2, dtype=torch.float32)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-6, rtol=0)


def test_sdf_indices_bit_exact_and_golden():
    z = np.load(GOLD / "sdf_reference.npz")
    spec = rb.load_robot("franka")
    grid = (z["data"], z["origin"], float(z["delta"]))
    sc = _scene(spec, grid, (0, 0, 0))
    idx, dist, grad = sc.sdf_query(torch.tensor(z["pos"]))
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), z["idx"])          # reference numpy twin, bit-exact
    assert np.array_equal(dist.cpu().numpy(), z["dist"].astype(np.float32))
    want_g = np.where(z["grad"] == 0, 0.1, z["grad"]).astype(np.float32)
    assert np.array_equal(grad.cpu().numpy(), want_g)
    # a large random sweep against the oracle, including cell boundaries hit exactly
    big = scenes.synthetic_boxes_sdf(n=40, delta=0.05, origin=(-1.0, -1.0, -1.0), seed=4)
    sc2 = _scene(spec, big, (0, 0, 0))
    rng = np.random.default_rng(1)
    pos = rng.uniform(-1.3, 1.3, (200000, 3))
    pos[:5000] = -1.0 + 0.05 * rng.integers(-2, 43, (5000, 3))                    # exact lattice points
    og = orc.SDFGrid(*big)
    idx2, dist2, grad2 = sc2.sdf_query(torch.tensor(pos))
    assert np.array_equal(idx2.cpu().numpy().astype(np.int64), orc.sdf_index(og, pos))
    assert np.array_equal(dist2.cpu().numpy(), orc.sdf_distance(og, pos).astype(np.float32))
    assert np.array_equal(grad2.cpu().numpy(), orc.sdf_gradient(og, pos).astype(np.float32))


@pytest.mark.parametrize("name,problem", [("franka", "industrial"), ("wam", "industrial"), ("ur10", "industrial")])
def test_log_prob_and_gradient(name, problem):
    ps = rb.load_problemset(name, problem)
    spec = rb.load_robot(name, *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    off = ps.object_positions[0]
    sc = _scene(spec, grid, off)
    osc = oracle_scene(spec, grid, off)
    rng = np.random.default_rng(2)
    g = rng.uniform(spec.low, spec.high, (4096, spec.dof)).astype(np.float32)
    logp, dl = sc.log_prob(torch.tensor(g), want_grad=True)
    want_lp, want_dl = orc.log_prob(osc, g.astype(np.float64), want_grad=True)
    logp, dl = logp.cpu().numpy(), dl.cpu().numpy()
    assert (want_lp < 0).mean() > 0.2, "scene must put spheres inside the hinge band"
    ok = np.isclose(logp, want_lp, rtol=2e-4, atol=1e-5)
    assert ok.mean() > 0.995, f"only {ok.mean():.4f} of configurations agree"       # voxel flips are rare
    scale = np.abs(want_dl).max(axis=1, keepdims=True) + 1e-6
    okg = (np.abs(dl - want_dl) / scale).max(axis=1) < 5e-4
    assert (okg | ~ok).mean() > 0.995
    # forward-only entry gives the same values
    lp2 = sc.log_prob(torch.tensor(g)).cpu().numpy()
    assert np.array_equal(lp2, logp)


def _planner(pb, sc, S, N, M, B, split_k=None, trainable=None):
    eng = _engine()
    pp = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=B, alpha=pb["alpha"], learning_rate=pb["lr"])
    ps = rb.load_problemset(pb["spec"].name if pb["spec"].name in rb.AVAILABLE_ROBOTS else "franka", "industrial")
    pl = eng.PlannerBatch(sc, pb["y"][None], lengthscales=ps.planner_params["lengthscales"],
                          variance=ps.planner_params["variance"], split_k=split_k, trainable=trainable, **pp)
    p = pb["params"]
    pl.q_mu.copy_(torch.tensor(p.q_mu.T[None]))
    pl.q_sqrt.copy_(torch.tensor(p.q_sqrt[None]))
    pl.raw_ell.copy_(torch.tensor(p.raw_ell[None]))
    pl.raw_var.copy_(torch.tensor(p.raw_var[None]))
    return pl


def _inject(pl, noise):
    pl.set_noise(noise.omega[None], noise.beta[None], noise.w[None], noise.eps[None], noise.eps2[None])


def _noise32(noise):
    r = lambda a: a.astype(np.float32).astype(np.float64)
    return orc.Noise(r(noise.omega), r(noise.beta), r(noise.w), r(noise.eps), r(noise.eps2))


@pytest.mark.parametrize("robot,S,N,M,B,split_k", [("franka", 6, 9, 5, 64, 1), ("franka", 37, 50, 10, 256, 4),
                                                    ("wam", 20, 33, 12, 128, 2), ("ur10", 16, 20, 6, 64, 1),
                                                    ("franka", 24, 40, 14, 128, 4),        # Mz = 16, N % 4 == 0: two-workgroup path kernels
                                                    ("franka", 8, 12, 46, 64, 2),          # Mz = 48: the largest inducing set
                                                    ("franka", 12, 16, 20, 64, 2),         # Mz = 22: two-panel elimination on a zero-padded 32 x 32 image
                                                    ("wam", 10, 14, 25, 64, 1),            # Mz = 27: odd, padded; (Kuu + jI)^-1 of the row tiles by dot products
                                                    ("franka", 128, 100, 30, 1024, 4)])     # BASELINE config 2, full size
def test_elbo_forward_backward_against_oracle(robot, S, N, M, B, split_k):
    pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=11, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    pl = _planner(pb, sc, S, N, M, B, split_k=split_k)
    noise = _noise32(pb["noise"])
    _inject(pl, noise)
    loss, grads = pl.loss_and_grad(generate=False)
    # the oracle looks its voxels up at the device's own float32 sphere centres: no query falls into a neighbouring cell
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, pb["alpha"], lookup_pos=_device_centres(pl, 0))
    og, G = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], noise, pb["alpha"], fw)
    L, Mz, J = pb["spec"].dof, M + 2, N + M + 2
    cv = fw["cv"]
    # covariance path is float64 on the device, rounded to float32 at the hand-over
    A4 = pl.view("A4").reshape(L, N, Mz, 4).cpu().numpy()
    np.testing.assert_allclose(A4[..., 0], cv["A"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pl.view("C").reshape(L, Mz, Mz).cpu().numpy(), cv["C"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pl.view("Kinv").reshape(L, Mz, Mz).cpu().numpy(), cv["Kinv"], rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(pl.view("kl_l").cpu().numpy().sum(), cv["kl"], rtol=1e-9)
    F0 = pl.view("F0").reshape(split_k, S, L, J).sum(0).cpu().numpy()
    np.testing.assert_allclose(F0, fw["F0"], rtol=0, atol=2e-5 * np.abs(fw["F0"]).max())
    H = pl.view("H").reshape(split_k, S, L, J).sum(0).cpu().numpy()
    np.testing.assert_allclose(H, fw["H"], rtol=0, atol=2e-5 * (np.abs(fw["H"]).max() + 1e-9))
    np.testing.assert_allclose(pl.view("R").reshape(S, L, Mz).cpu().numpy(), fw["R"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pl.f[0].cpu().numpy(), fw["f"], rtol=0, atol=1e-4)
    logp = pl.logp[0].cpu().numpy()
    assert (fw["logp"] < 0).any()
    tag = f"elbo[{robot},S={S},N={N},M={M}]"
    top = np.abs(fw["logp"]).max()
    print(f"PARITY {tag} logp={np.abs(logp - fw['logp']).max() / top:.2e} lik={abs(float(pl.lik[0]) - fw['lik']) / abs(fw['lik']):.2e}")
    # same voxels on both sides: EVERY (sample, time) pair agrees to float32 accuracy, and so do the sums and the gradients
    np.testing.assert_allclose(logp, fw["logp"], rtol=0, atol=TOL_LOGP * top)
    np.testing.assert_allclose(float(pl.kl[0]), cv["kl"], rtol=1e-9)
    np.testing.assert_allclose(float(pl.lik[0]), fw["lik"], rtol=TOL_LIK)
    np.testing.assert_allclose(float(loss[0]), -fw["elbo"], rtol=2e-5)
    _assert_grads(tag, grads, og)
    # ... and how far a float64 chain's own voxels are from that: a measured, bounded share of the pairs
    fw64 = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, pb["alpha"], want_dell=False)
    _flipped_share(tag, logp, fw64)


def test_kl_only_gradient_is_float64_exact():
    """With alpha = 0 the likelihood drops out and the whole reverse pass is the float64 covariance
    path: gradients must match the oracle to 1e-9 relative."""
    pb = small_problem(robot="franka", S=4, N=7, M=9, B=32, seed=5, n_grid=24)
    pb["alpha"] = 0.0
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    pl = _planner(pb, sc, 4, 7, 9, 32, split_k=1)
    noise = _noise32(pb["noise"])
    _inject(pl, noise)
    loss, grads = pl.loss_and_grad(generate=False)
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, 0.0)
    og, _ = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], noise, 0.0, fw)
    np.testing.assert_allclose(float(loss[0]), fw["cv"]["kl"], rtol=1e-10)
    np.testing.assert_allclose(grads[0][0].cpu().numpy().T, og.q_mu, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(grads[1][0].cpu().numpy(), og.q_sqrt, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(grads[2][0].cpu().numpy(), og.raw_ell, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(grads[3][0].cpu().numpy(), og.raw_var, rtol=1e-6, atol=1e-9)


def test_adam_trajectory_matches_oracle():
    """Five optimisation steps with injected noise: parameters track the oracle's Keras-Adam."""
    S, N, M, B = 8, 12, 6, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=7, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    pl = _planner(pb, sc, S, N, M, B, split_k=1)
    p = pb["params"].copy(); st = orc.adam_init(p)
    rng = np.random.default_rng(3)
    for step in range(5):
        noise = _noise32(orc.draw_noise(rng, S, 7, 7, B, M + 2))
        _inject(pl, noise)
        pl.step(generate=False)
        orc.optimization_step(p, st, pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, pb["alpha"], pb["lr"])
    # Adam normalises the gradient, so a float32-level gradient difference moves a parameter by
    # ~lr * 1e-3 per step at most; anything larger is a real disagreement.
    tol = 5 * pb["lr"] * 2e-2
    assert np.abs(pl.q_mu[0].cpu().numpy().T - p.q_mu).max() < tol
    assert np.abs(pl.q_sqrt[0].cpu().numpy() - p.q_sqrt).max() < tol
    assert np.abs(pl.raw_ell[0].cpu().numpy() - p.raw_ell).max() < tol
    assert np.abs(pl.raw_var[0].cpu().numpy() - p.raw_var).max() < tol


@pytest.mark.parametrize("S,N,P", [(8, 12, 1), (40, 50, 1), (128, 100, 6)])   # 8 / 8 / 1 lanes per configuration
def test_likelihood_constants_gradient_against_oracle(S, N, P):
    """trainable_params.sigma_obs / alpha (vgpmp_lik_params): gradient of the training loss wrt the two raw variables,
    every problem of the batch with its own values."""
    import dataclasses
    M, B = 6, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=9, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    eng = _engine()
    ps = rb.load_problemset("franka", "industrial")
    tr = dict(q_mu=True, q_sqrt=True, lengthscales=True, kernel_variance=True, sigma_obs=True, alpha=True)
    pl = eng.PlannerBatch(sc, np.repeat(pb["y"][None], P, 0), lengthscales=ps.planner_params["lengthscales"],
                          variance=ps.planner_params["variance"], trainable=tr, num_samples=S, num_inducing=M,
                          num_data=N, num_bases=B, alpha=pb["alpha"], learning_rate=pb["lr"])
    assert pl.lik_variables
    p = pb["params"]
    nsph = pb["spec"].num_spheres
    rng = np.random.default_rng(4)
    lps = []
    for k in range(P):
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
        lp = orc.init_lik_params(pb["alpha"] * (1.0 + 0.3 * k), pb["scene"].sigma_obs * (1.0 + rng.uniform(0, 1, nsph)))
        pl.raw_alpha[k] = float(lp.raw_alpha)
        pl.raw_sigma[k, :nsph] = torch.tensor(lp.raw_sigma)
        lps.append(lp)
    noise = _noise32(pb["noise"])
    rep = lambda a: np.repeat(a[None], P, 0)
    pl.set_noise(rep(noise.omega), rep(noise.beta), rep(noise.w), rep(noise.eps), rep(noise.eps2))
    loss, grads = pl.loss_and_grad(generate=False)
    torch.cuda.synchronize()
    for k in range(P):
        alpha, sigma = orc.lik_constrained(lps[k])
        osc = dataclasses.replace(pb["scene"], sigma_obs=sigma)
        fw = orc.elbo_forward(p, osc, pb["X"], pb["Zy"], pb["y"], noise, alpha)
        assert (fw["logp"] < 0).any()
        gl = orc.lik_backward(lps[k], osc, fw)
        np.testing.assert_allclose(float(loss[k]), -fw["elbo"], rtol=3e-4)
        # float32 FK flips a few voxels: tolerance on the scale of the gradient, as for the other variables
        np.testing.assert_allclose(float(pl.lik_grad[0][k]), float(gl.raw_alpha), rtol=2e-3)
        got = pl.lik_grad[1][k, :nsph].cpu().numpy()
        assert np.abs(got - gl.raw_sigma).max() <= 2e-3 * np.abs(gl.raw_sigma).max() + 1e-6, (got, gl.raw_sigma)
        assert np.all(pl.lik_grad[1][k, nsph:].cpu().numpy() == 0.0)
        og, _ = orc.elbo_backward(p, osc, pb["X"], pb["Zy"], noise, alpha, fw)
        gq = grads[0][k].cpu().numpy().T
        assert np.abs(gq - og.q_mu).max() <= 5e-3 * np.abs(og.q_mu).max() + 1e-6


def test_likelihood_constants_adam_trajectory_matches_oracle():
    """Five optimisation steps with sigma_obs and alpha among the variables (one optimizer, shared step count)."""
    import dataclasses
    S, N, M, B = 8, 12, 6, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=9, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    tr = dict(orc.DEFAULT_TRAINABLE, sigma_obs=True, alpha=True)
    pl = _planner(pb, sc, S, N, M, B, split_k=1, trainable=tr)
    nsph = pb["spec"].num_spheres
    p = pb["params"].copy(); st = orc.adam_init(p)
    lp = orc.init_lik_params(pb["alpha"], pb["scene"].sigma_obs)
    np.testing.assert_allclose(pl.raw_alpha[0].item(), float(lp.raw_alpha), rtol=1e-12)
    np.testing.assert_allclose(pl.raw_sigma[0, :nsph].cpu().numpy(), lp.raw_sigma, rtol=1e-12)
    lp = orc.init_lik_params(3.0, pb["scene"].sigma_obs)          # small alpha: the bijector's slope matters
    pl.raw_alpha[0] = float(lp.raw_alpha)
    lp0 = lp.copy()
    zl = lambda: orc.LikParams(np.zeros(()), np.zeros(nsph))
    st_lik = dict(m=zl(), v=zl())
    rng = np.random.default_rng(3)
    active = False
    for step in range(5):
        noise = _noise32(orc.draw_noise(rng, S, 7, 7, B, M + 2))
        _inject(pl, noise)
        pl.step(generate=False)
        alpha, sigma = orc.lik_constrained(lp)
        fw = orc.elbo_forward(p, dataclasses.replace(pb["scene"], sigma_obs=sigma), pb["X"], pb["Zy"], pb["y"], noise, alpha)
        active = active or bool((fw["logp"] < 0).any())
        orc.optimization_step_lik(p, lp, st, st_lik, pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, pb["lr"], tr)
    assert active, "fixture must have active hinge terms"
    tol = 5 * pb["lr"] * 2e-2
    assert abs(pl.raw_alpha[0].item() - float(lp.raw_alpha)) < tol
    assert np.abs(pl.raw_sigma[0, :nsph].cpu().numpy() - lp.raw_sigma).max() < tol
    assert np.abs(pl.q_mu[0].cpu().numpy().T - p.q_mu).max() < tol
    assert np.abs(pl.raw_ell[0].cpu().numpy() - p.raw_ell).max() < tol
    # the variables did move
    assert abs(float(lp.raw_alpha) - float(lp0.raw_alpha)) > pb["lr"]
    assert np.abs(lp.raw_sigma - lp0.raw_sigma).min() > pb["lr"]


def test_device_philox_matches_oracle_stream():
    S, N, M, B = 5, 6, 4, 32
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=1, n_grid=16)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    eng = _engine()
    pl = eng.PlannerBatch(sc, np.stack([pb["y"], pb["y"][::-1]]), num_samples=S, num_inducing=M, num_data=N,
                          num_bases=B, lengthscales=[2.0] * 7, variance=0.2, seed=123, problem_base=40)
    pl.generate_noise(step=9)
    for p in range(2):
        nz = orc.philox_noise(123, 40 + p, 9, S, 7, 7, B, M + 2)
        # the prior weights come out of Philox bits and a table of float16 values: integer work, the oracle's weights BIT FOR BIT
        assert np.array_equal(pl.w[p].cpu().numpy().astype(np.float64), nz.w)
        np.testing.assert_allclose(pl.eps[p].cpu().numpy(), nz.eps, rtol=0, atol=2e-5)
        np.testing.assert_allclose(pl.eps2[p].cpu().numpy(), nz.eps2, rtol=0, atol=2e-5)
        np.testing.assert_allclose(pl.beta[p].cpu().numpy(), nz.beta, rtol=0, atol=1e-6)
        np.testing.assert_allclose(pl.omega[p].cpu().numpy(), nz.omega, rtol=2e-4, atol=2e-5)


def test_problem_batch_is_independent_and_generated_noise_runs():
    """Three problems in one batch: each equals the same problem solved alone (no cross-talk)."""
    S, N, M, B = 8, 10, 5, 64
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = _scene(spec, grid, ps.object_positions[0])
    qs = np.array([[ps.states[0], ps.states[1]], [ps.states[2], ps.states[3]], [ps.states[4], ps.states[6]]])
    eng = _engine()
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, seed=5)
    batch = eng.PlannerBatch(sc, qs, **kw)
    for _ in range(3):
        batch.step()
    for p in range(3):
        solo = eng.PlannerBatch(sc, qs[p:p + 1], problem_base=p, **kw)
        for _ in range(3):
            solo.step()
        assert torch.equal(solo.q_mu[0], batch.q_mu[p]) and torch.equal(solo.q_sqrt[0], batch.q_sqrt[p])
        assert torch.equal(solo.raw_ell[0], batch.raw_ell[p])
    assert torch.isfinite(batch.elbo()).all()


def test_full_size_properties():
    """BASELINE config 2 shape (S=128, M=30, N=100, B=1024): size-independent properties --
    deterministic replay, finite outputs, KL non-negative, loss decreases over 30 steps."""
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=128, delta=0.0125, origin=(-0.8, -0.8, -0.2), seed=0)
    sc = _scene(spec, grid, ps.object_positions[0])
    eng = _engine()
    kw = dict(num_samples=128, num_inducing=30, num_data=100, num_bases=1024, lengthscales=[2.0] * 7, variance=0.2,
              learning_rate=0.02, seed=1)
    q = np.array([[ps.states[0], ps.states[1]]])
    a = eng.PlannerBatch(sc, q, **kw)
    b = eng.PlannerBatch(sc, q, **kw)
    l0 = float(-a.elbo(step=10**6)[0])
    for _ in range(30):
        a.step(); b.step()
    assert torch.equal(a.q_mu, b.q_mu) and torch.equal(a.q_sqrt, b.q_sqrt)     # bitwise replay
    l1 = float(-a.elbo(step=10**6)[0])
    assert np.isfinite([l0, l1]).all() and float(a.kl[0]) >= 0.0
    assert l1 < l0
    g = a.samples()
    lo, hi = torch.tensor(spec.low, device=g.device), torch.tensor(spec.high, device=g.device)
    assert bool(((g >= lo) & (g <= hi)).all())
    # paths are pinned to start/goal by the 1e-6 conditioning: sample spread at t=0 and t=1 is tiny
    y = torch.tensor(q[0], device=g.device, dtype=g.dtype)
    assert float((g[0, :, 0, :] - y[0]).abs().max()) < 5e-2 and float((g[0, :, -1, :] - y[1]).abs().max()) < 5e-2


@pytest.mark.parametrize("robot,S,N,M,B,P", [("franka", 8, 12, 6, 64, 1), ("ur10", 24, 20, 10, 128, 2), ("franka", 128, 100, 30, 1024, 1)])
def test_inducing_location_gradient_against_oracle(robot, S, N, M, B, P):
    """trainable_params.inducing_variable (utils/miscellaneous.py:338; Z = 0.09 + 0.82 sigmoid(raw_Z), models/vgpmp.py:29-42):
    d loss / d raw_Z of every problem from the device's reverse pass (float64 covariance / Cholesky adjoint, float32 sums over
    the samples, the random-feature prior at the rows of Zy) against the oracle, at perturbed, column-wise different
    locations; the other gradients are unaffected by where Zy comes from."""
    pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=11, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    D = pb["spec"].dof
    tr = dict(orc.DEFAULT_TRAINABLE, inducing_variable=True)
    eng = _engine()
    ps = rb.load_problemset(robot, "industrial")
    pl = eng.PlannerBatch(sc, np.repeat(pb["y"][None], P, 0), lengthscales=ps.planner_params["lengthscales"],
                          variance=ps.planner_params["variance"], trainable=tr, num_samples=S, num_inducing=M, num_data=N,
                          num_bases=B, alpha=pb["alpha"], learning_rate=pb["lr"])
    assert pl.z_variables
    np.testing.assert_allclose(pl.raw_Z[0].cpu().numpy(), orc.init_raw_Z(M, D), rtol=1e-12)
    rng = np.random.default_rng(6)
    raws = [orc.init_raw_Z(M, D) + 0.25 * rng.standard_normal((M, D)) for _ in range(P)]
    p = pb["params"]
    for k in range(P):
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
        pl.raw_Z[k].copy_(torch.tensor(raws[k]))
    noise = _noise32(pb["noise"])
    rep = lambda a: np.repeat(a[None], P, 0)
    pl.set_noise(rep(noise.omega), rep(noise.beta), rep(noise.w), rep(noise.eps), rep(noise.eps2))
    loss, grads = pl.loss_and_grad(generate=False)
    torch.cuda.synchronize()
    for k in range(P):
        Zy = orc.zy_from_raw(raws[k])
        np.testing.assert_allclose(pl.Zy_all[k].cpu().numpy(), Zy, rtol=1e-13, atol=1e-15)
        fw = orc.elbo_forward(p, pb["scene"], pb["X"], Zy, pb["y"], noise, pb["alpha"], lookup_pos=_device_centres(pl, k))
        og, _, g_zy = orc.elbo_backward(p, pb["scene"], pb["X"], Zy, noise, pb["alpha"], fw, want_z=True)
        want = orc.z_backward(raws[k], g_zy)
        assert (fw["logp"] < 0).any()
        tag = f"inducing[{robot},S={S},k={k}]"
        # the device's own voxels on both sides: every pair agrees, fixed gradient tolerances
        np.testing.assert_allclose(pl.logp[k].cpu().numpy(), fw["logp"], rtol=0, atol=TOL_LOGP * np.abs(fw["logp"]).max())
        np.testing.assert_allclose(float(pl.kl[k]), fw["cv"]["kl"], rtol=1e-9)
        got = pl.z_grad[k].cpu().numpy()
        scale = np.abs(want).max()
        assert scale > 1e-3
        print(f"PARITY {tag} raw_Z={np.abs(got - want).max() / scale:.2e}")
        assert np.abs(got - want).max() / scale < TOL_GRAD, (np.abs(got - want).max(), scale)
        _assert_grads(tag, grads, og, k=k)


def test_inducing_location_kl_gradient_is_float64_exact():
    """alpha = 0: only the KL depends on Z, and that path is float64 on the device: 1e-7."""
    S, N, M, B = 4, 7, 9, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=5, n_grid=24)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    tr = dict(orc.DEFAULT_TRAINABLE, inducing_variable=True)
    pl = _planner(dict(pb, alpha=0.0), sc, S, N, M, B, split_k=1, trainable=tr)
    rng = np.random.default_rng(2)
    raw = orc.init_raw_Z(M, 7) + 0.3 * rng.standard_normal((M, 7))
    pl.raw_Z[0].copy_(torch.tensor(raw))
    noise = _noise32(pb["noise"])
    _inject(pl, noise)
    pl.loss_and_grad(generate=False)
    Zy = orc.zy_from_raw(raw)
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], Zy, pb["y"], noise, 0.0)
    _, _, g_zy = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], Zy, noise, 0.0, fw, want_z=True)
    want = orc.z_backward(raw, g_zy)
    np.testing.assert_allclose(pl.z_grad[0].cpu().numpy(), want, rtol=1e-7, atol=1e-9 * np.abs(want).max())


def test_inducing_location_adam_trajectory_matches_oracle():
    """Five optimisation steps with the inducing locations among the variables (one optimizer, shared step count); four chained
    steps of one vgpmp_elbo_steps call reproduce four one-step calls bit for bit."""
    S, N, M, B = 8, 12, 6, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=7, n_grid=48)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    tr = dict(orc.DEFAULT_TRAINABLE, inducing_variable=True)
    pl = _planner(pb, sc, S, N, M, B, split_k=1, trainable=tr)
    p = pb["params"].copy(); st = orc.adam_init(p)
    raw = orc.init_raw_Z(M, 7)
    st_z = dict(m=np.zeros_like(raw), v=np.zeros_like(raw))
    rng = np.random.default_rng(3)
    for step in range(5):
        noise = _noise32(orc.draw_noise(rng, S, 7, 7, B, M + 2))
        _inject(pl, noise)
        pl.step(generate=False)
        orc.optimization_step_z(p, raw, st, st_z, pb["scene"], pb["X"], pb["y"], noise, pb["alpha"], pb["lr"], tr)
    tol = 5 * pb["lr"] * 2e-2
    assert np.abs(pl.raw_Z[0].cpu().numpy() - raw).max() < tol
    assert np.abs(pl.q_mu[0].cpu().numpy().T - p.q_mu).max() < tol
    assert np.abs(pl.raw_ell[0].cpu().numpy() - p.raw_ell).max() < tol
    assert np.abs(raw - orc.init_raw_Z(M, 7)).min() > 1e-3 and np.abs(raw - orc.init_raw_Z(M, 7)).max() > pb["lr"]    # the locations moved
    Z = pl.inducing_locations()[0].cpu().numpy()
    assert (Z > 0.09).all() and (Z < 0.91).all()
    # generated noise: 4 chained steps == 4 single-step calls
    a = _planner(pb, sc, S, N, M, B, split_k=1, trainable=tr)
    b = _planner(pb, sc, S, N, M, B, split_k=1, trainable=tr)
    for _ in range(4):
        a.run_steps(1)
    b.run_steps(4)
    for x, y in ((a.raw_Z, b.raw_Z), (a.q_mu, b.q_mu), (a.raw_ell, b.raw_ell), (a.z_adam_v, b.z_adam_v)):
        assert torch.equal(x, y)
    assert float((a.raw_Z - torch.tensor(orc.init_raw_Z(M, 7), device=a.raw_Z.device)).abs().max()) > 0


def _compare_with_oracle(pl, k, p, scene, X, Zy, y, noise, alpha, S, N, M, L, split_k, lik_scale=1.0, kl_scale=1.0, tag="batch"):
    """Problem k of the batch `pl` against the oracle on the device's own voxels, with the tolerances of
    test_elbo_forward_backward_against_oracle.
    lik_scale / kl_scale: a rank's share of a sharded sample axis (S local of S_total samples, KL on rank 0 only)."""
    Mz, J = M + 2, N + M + 2
    fw = orc.elbo_forward(p, scene, X, Zy, y, noise, alpha * lik_scale, lookup_pos=_device_centres(pl, k))
    og, _ = orc.elbo_backward(p, scene, X, Zy, noise, alpha * lik_scale, fw)
    if kl_scale != 1.0:
        fw0 = orc.elbo_forward(p, scene, X, Zy, y, noise, 0.0)
        g0, _ = orc.elbo_backward(p, scene, X, Zy, noise, 0.0, fw0)            # the KL's own gradient
        for name in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
            setattr(og, name, getattr(og, name) - (1.0 - kl_scale) * getattr(g0, name))
    cv = fw["cv"]
    A4 = pl.view("A4").reshape(pl.P, L, N, Mz, 4)[k].cpu().numpy()
    np.testing.assert_allclose(A4[..., 0], cv["A"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pl.view("C").reshape(pl.P, L, Mz, Mz)[k].cpu().numpy(), cv["C"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pl.view("kl_l").reshape(pl.P, -1)[k].cpu().numpy().sum(), cv["kl"], rtol=1e-9)
    F0 = pl.view("F0").reshape(split_k, pl.P, S, L, J).sum(0)[k].cpu().numpy()
    np.testing.assert_allclose(F0, fw["F0"], rtol=0, atol=2e-5 * np.abs(fw["F0"]).max())
    H = pl.view("H").reshape(split_k, pl.P, S, L, J).sum(0)[k].cpu().numpy()
    np.testing.assert_allclose(H, fw["H"], rtol=0, atol=2e-5 * (np.abs(fw["H"]).max() + 1e-9))
    np.testing.assert_allclose(pl.view("R").reshape(pl.P, S, L, Mz)[k].cpu().numpy(), fw["R"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pl.f[k].cpu().numpy(), fw["f"], rtol=0, atol=1e-4)
    logp = pl.logp[k].cpu().numpy()
    # the device's own voxels on both sides: every (sample, time) pair, fixed tolerances
    top = np.abs(fw["logp"]).max()
    np.testing.assert_allclose(logp, fw["logp"], rtol=0, atol=TOL_LOGP * top + 1e-30)
    np.testing.assert_allclose(float(pl.kl[k]), kl_scale * cv["kl"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(float(pl.lik[k]), fw["lik"], rtol=TOL_LIK, atol=1e-12)
    _assert_grads(f"{tag}[k={k}]", pl.grad, og, k=k)
    if k == 0:      # one un-overridden forward per configuration: the share of pairs a float64 chain resolves differently
        _flipped_share(tag, logp, orc.elbo_forward(p, scene, X, Zy, y, noise, alpha * lik_scale, want_dell=False))
    return fw


@pytest.mark.parametrize("config", ["config1", "config3", "config4", "ur10_few_samples", "wam_twenty_samples"])
def test_baseline_configs_at_full_size_against_oracle(config):
    """BASELINE configs 1, 3 and 4 at their FULL (S, M, N, B = 1024) against the oracle, not only
    through size-independent properties.  config1 = WAM / industrial, S=50 M=10 N=70 (data/problemsets/wam.py:93-106);
    config3 = Franka / bookshelves, S=7 M=24 N=70 with 10 of the 55 start-goal pairs in ONE batch (70 latents: beyond the 64 the
    few-problem schedule takes at this sample count), so the few-sample prior kernel -- its 4 x 4 x 1 MFMA form for up to 8
    samples -- and the batch schedule run at batch size (data/problemsets/franka.py:91-104); config4 = one rank's share of UR10 /
    industrial: S=128 of 1024 samples at sample_offset 256, KL owned by another rank (data/problemsets/ur10.py:71-84)."""
    robot, problem, S, M, N, P, extra = {
        "config1": ("wam", "industrial", 50, 10, 70, 1, {}),
        "config3": ("franka", "bookshelves", 7, 24, 70, 10, {}),
        "config4": ("ur10", "industrial", 128, 18, 70, 1, dict(samples_total=1024, sample_offset=256, kl_scale=0.0)),
        # the few-sample prior kernel's 6-joint form / its two-tile form at the reference's other sample count (20), both as batches
        # beyond the few-problem schedule (data/problemsets/ur10.py:71-84, wam.py:107-120)
        "ur10_few_samples": ("ur10", "industrial", 7, 18, 70, 12, {}),
        "wam_twenty_samples": ("wam", "bookshelves", 20, 15, 100, 10, {}),
    }[config]
    _injected_batch_against_oracle(config, robot, problem, S, M, N, P, extra)


def _injected_batch_against_oracle(tag, robot, problem, S, M, N, P, extra, check=None):
    """A batch of P problems of a reference problem set at B = 1024 on INJECTED noise (arbitrary float32 draws of the host) against the
    oracle: every problem, or the problems `check` names.  Returns the kernels the library ran."""
    B = 1024
    ps = rb.load_problemset(robot, problem)
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    pp = ps.planner_params
    grid = scenes.synthetic_boxes_sdf(n=64, delta=2.4 / 64, origin=(-1.2, -1.2, -0.6), seed=3)
    off = ps.object_positions[0]
    sc = _engine().DeviceScene(spec, grid, off, sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    osc = oracle_scene(spec, grid, off, sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    L = spec.dof
    step = max(1, len(ps.queries) // P)
    qs = np.array([ps.queries[(i * step) % len(ps.queries)] for i in range(P)], dtype=np.float64)
    pl = _engine().PlannerBatch(sc, qs, num_samples=S, num_inducing=M, num_data=N, num_bases=B,
                                lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                                learning_rate=pp["learning_rate"], **extra)
    split_k = pl.dims.split_k
    rng = np.random.default_rng(21)
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    check = list(range(P)) if check is None else list(check)
    params, noises = {}, {}
    for k in range(P):
        p = orc.init_params(osc.robot, qs[k], M, pp["lengthscales"], max(pp["variance"], 0.1 + 1e-6))
        p.q_sqrt = np.tril(p.q_sqrt + 0.05 * rng.standard_normal(p.q_sqrt.shape))
        p.q_mu = p.q_mu + 0.05 * rng.standard_normal(p.q_mu.shape)
        nz = _noise32(orc.draw_noise(rng, S, L, L, B, M + 2))
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
        # (one problem's draws at a time: 64 problems of config 2's shape are 0.5 GB of float64 on the host)
        for dst, src in ((pl.omega, nz.omega), (pl.beta, nz.beta), (pl.w, nz.w), (pl.eps, nz.eps), (pl.eps2, nz.eps2)):
            dst[k].copy_(torch.as_tensor(src, dtype=torch.float32).reshape(dst[k].shape))
        if k in check:
            params[k], noises[k] = p, nz
    pl.noise_ahead_step = None
    pl.loss_and_grad(generate=False)
    torch.cuda.synchronize()
    from vgpmp_amd import capi
    ran = capi.last_schedule(pl.lib)
    lik_scale = S / float(extra.get("samples_total", S))
    active = False
    for k in check:
        fw = _compare_with_oracle(pl, k, params[k], osc, X, Zy, qs[k], noises[k], float(pp["alpha"]), S, N, M, L, split_k,
                                  lik_scale=lik_scale, kl_scale=float(extra.get("kl_scale", 1.0)), tag=tag)
        active = active or bool((fw["logp"] < 0).any())
    assert active, "the scene must put spheres inside the hinge band for at least one problem"
    return ran


@pytest.mark.parametrize("shape", ["config3_x64", "config2_x64"])
def test_callers_own_weights_at_batch_size_against_oracle(shape):
    """the library's own W stream is a float16 table draw (two MFMAs per product); a binder that supplies its OWN weights
    hands over arbitrary float32 normals (generate = False).  That path at batch size, 64 problems: config 3's shape (S = 7: the
    few-sample prior kernel in its three-MFMA form, WX = false -- W split into two f16 halves) and config 2's shape (S = 128: the
    stored-W schedule -- features kernel + float32-MFMA tiled GEMM), four of the 64 problems against the oracle with the fixed
    tolerances of the one-step tests (models/vgpmp.py:281-282)."""
    robot, problem, S, M, N = {"config3_x64": ("franka", "bookshelves", 7, 24, 70),
                               "config2_x64": ("franka", "industrial", 128, 30, 100)}[shape]
    ran = _injected_batch_against_oracle(shape, robot, problem, S, M, N, 64, {}, check=(0, 21, 42, 63))
    print(f"PARITY {shape} kernels: {ran}")
    prior = [k for k in ran if k.startswith(("prior_", "mid_cov_b_prior16"))]
    assert prior, ran
    if shape == "config3_x64":
        # (template arguments <MT, DM, DELL, WX>: the caller's weights take the split form)
        assert any(k.startswith("prior_fused_small16_kernel") and k.rstrip(">").endswith("false") for k in prior), prior
    else:
        assert any(k.startswith("prior_gemm") for k in prior) and not any("split" in k for k in prior), prior


def test_adam_update_arithmetic_is_float64_exact():
    """with alpha = 0 the whole optimisation step is float64 on the device (covariance path, KL, its
    reverse, the update), so a five-step trajectory pins Keras' Adam(lr, 0.8, 0.95, epsilon 1e-7) of models/vgpmp.py:77
    exactly -- moment recursion, bias correction and the place of epsilon (outside the square root): 1e-9, where the
    float32 trajectories above can only afford steps * lr * 2e-2."""
    S, N, M, B = 4, 9, 7, 32
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=5, n_grid=24)
    sc = _scene(pb["spec"], pb["grid"], pb["offset"])
    pl = _planner(dict(pb, alpha=0.0), sc, S, N, M, B, split_k=1)
    p = pb["params"].copy(); st = orc.adam_init(p)
    p0 = pb["params"].copy()
    rng = np.random.default_rng(8)
    for step in range(5):
        noise = _noise32(orc.draw_noise(rng, S, 7, 7, B, M + 2))
        _inject(pl, noise)
        pl.step(generate=False)
        orc.optimization_step(p, st, pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, 0.0, pb["lr"])
    for got, want, start in ((pl.q_mu[0].cpu().numpy().T, p.q_mu, p0.q_mu), (pl.q_sqrt[0].cpu().numpy(), p.q_sqrt, p0.q_sqrt),
                             (pl.raw_ell[0].cpu().numpy(), p.raw_ell, p0.raw_ell), (pl.raw_var[0].cpu().numpy(), p.raw_var, p0.raw_var)):
        assert np.abs(want - start).max() > pb["lr"], "the variable must have moved"
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    # the moments themselves (float64 state)
    np.testing.assert_allclose(pl.adam_m[0][0].cpu().numpy().T, st.m.q_mu, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(pl.adam_v[1][0].cpu().numpy(), st.v.q_sqrt, rtol=1e-7, atol=1e-14)
