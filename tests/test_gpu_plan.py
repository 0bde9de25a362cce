"""Plan extraction on the device (vgpmp_sample_paths; models/vgpmp.py:312-339) against the oracle with injected noise:
posterior mean, the pathwise samples as joint angles, get_best_sample's arg-max, the best path, the end-effector
variance of compute_uncertainty=True, and the headless clearance check that stands in for the simulated execution."""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
from helpers import small_problem

pytestmark = pytest.mark.gpu


def _setup(robot, S, N, M, B, seed=31):
    from vgpmp_amd import engine
    pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=seed, n_grid=40)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    pl = engine.PlannerBatch(sc, np.stack([pb["y"], pb["y"][::-1]]), num_samples=4, num_inducing=M, num_data=7, num_bases=B,
                             lengthscales=[2.0] * pb["spec"].dof, variance=0.2, alpha=pb["alpha"])
    p = pb["params"]
    for k in range(2):
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
    return pb, sc, pl


@pytest.mark.parametrize("robot", ["franka", "ur10"])
def test_sample_from_posterior_against_oracle(robot):
    S, N, M, B = 150, 23, 6, 64
    pb, sc, pl = _setup(robot, S, N, M, B)
    D = pb["spec"].dof
    Xnew = orc.init_trainset(N, D)
    sp = pl.posterior_sampler(S, Xnew)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(5)
    nz = orc.draw_noise(rng, S, D, D, B, M + 2)
    nz = orc.Noise(r32(nz.omega), r32(nz.beta), r32(nz.w), r32(nz.eps), r32(nz.eps2))
    rep = lambda a: np.stack([a, a])
    sp.set_noise(rep(nz.omega), rep(nz.beta), rep(nz.w), rep(nz.eps), rep(nz.eps2))
    sp.elbo(generate=False)
    mean, best_path, samples, best, ee = sp.extract_plans(True, True)
    torch.cuda.synchronize()
    ys = [pb["y"], pb["y"][::-1]]
    for k in range(2):
        w_mean, w_best_path, w_samples, w_best = orc.sample_from_posterior(pb["params"], pb["scene"], Xnew, pb["Zy"], ys[k], nz)
        # posterior mean: float32 A (rounded from float64) times float32 q_mu, then the joint sigmoid
        np.testing.assert_allclose(mean[k].cpu().numpy(), w_mean, rtol=0, atol=1e-5)
        np.testing.assert_allclose(samples[k].cpu().numpy(), w_samples, rtol=0, atol=2e-4)
        fw = orc.elbo_forward(pb["params"], pb["scene"], Xnew, pb["Zy"], ys[k], nz, 1.0, want_dell=False)
        score = fw["logp"].sum(-1)
        got = int(best[k])
        # the device sums float32 log-likelihoods of float32 paths: its arg-max is the oracle's unless the two leading
        # scores are closer than that noise
        order = np.argsort(-score)
        assert got == w_best or abs(score[got] - score[w_best]) <= 2e-3 * abs(score[w_best]) + 1e-6, (got, w_best, score[order[:3]])
        np.testing.assert_allclose(best_path[k].cpu().numpy(), samples[k, got].cpu().numpy(), rtol=0, atol=0)
        # compute_uncertainty=True: variance over the samples of the last frame's origin (models/vgpmp.py:322-327)
        fr = orc.forward_kinematics(pb["scene"].robot, w_samples.reshape(-1, D)).reshape(S, N, D + 1, 4, 4)
        w_var = fr[:, :, -1, :3, 3].var(axis=0)
        np.testing.assert_allclose(ee[k].cpu().numpy(), w_var, rtol=2e-3, atol=1e-8)
    # the engine entry point = forward-only step with generated noise + the same extraction; shapes and the mean agree
    out = pl.sample_from_posterior(S, Xnew, step=3)
    assert out[1].shape == (2, N, D) and out[2].shape == (2, S, N, D) and out[3].dtype == torch.int32
    np.testing.assert_allclose(out[0].cpu().numpy(), mean.cpu().numpy(), rtol=0, atol=0)


def test_sampler_cache_follows_the_time_stamps_of_each_call():
    """Two calls with the same number of time stamps but different values: each is evaluated at ITS Xnew."""
    pb, sc, pl = _setup("franka", 150, 23, 6, 64)
    Xa = orc.init_trainset(17, 7)
    Xb = Xa ** 2                                        # same count, different stamps (still 0 .. 1)
    ma = pl.sample_from_posterior(30, Xa, step=1)[0].cpu().numpy()
    mb = pl.sample_from_posterior(30, Xb, step=1)[0].cpu().numpy()
    wa = orc.posterior_mean(pb["params"], pb["scene"].robot, Xa, pb["Zy"], pb["y"])
    wb = orc.posterior_mean(pb["params"], pb["scene"].robot, Xb, pb["Zy"], pb["y"])
    np.testing.assert_allclose(ma[0], wa, rtol=0, atol=1e-5)
    np.testing.assert_allclose(mb[0], wb, rtol=0, atol=1e-5)
    assert np.abs(wa - wb).max() > 1e-2


def test_planners_of_one_scene_share_the_sampler_view_but_not_its_variables():
    """solve_planning_problem builds one planner per start-goal query; their 150-path views have the same shape and hand one
    set of buffers on.  Two planners alive at once must still each sample from their OWN variables."""
    from vgpmp_amd import engine
    pb, sc, a = _setup("franka", 16, 9, 6, 64)
    b = engine.PlannerBatch(sc, np.stack([pb["y"][::-1], pb["y"]]), num_samples=4, num_inducing=6, num_data=7, num_bases=64,
                            lengthscales=[2.0] * 7, variance=0.2, alpha=pb["alpha"])
    b.q_mu.mul_(0.5)
    ma = a.sample_from_posterior(16, None, step=3)[0].clone()
    mb = b.sample_from_posterior(16, None, step=3)[0].clone()
    assert a.posterior_sampler(16) is b.posterior_sampler(16)            # one view ...
    ma2 = a.sample_from_posterior(16, None, step=3)[0].clone()
    mb2 = b.sample_from_posterior(16, None, step=3)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(ma, ma2) and torch.equal(mb, mb2) and not torch.equal(ma, mb)      # ... two sets of variables


def test_path_clearance_against_oracle():
    pb, sc, pl = _setup("franka", 150, 23, 6, 64)
    rng = np.random.default_rng(8)
    spec = pb["spec"]
    path = rng.uniform(spec.low, spec.high, (2, 19, spec.dof)).astype(np.float32)
    got = pl.path_clearance(torch.tensor(path, device=sc.device)).cpu().numpy()
    pos = orc.sphere_positions(pb["scene"].robot, path.reshape(-1, spec.dof).astype(np.float64))
    want = orc.sdf_distance(pb["scene"].sdf, pos.reshape(-1, 3) - pb["scene"].offset).reshape(2, 19, -1) - spec.sphere_radii
    ok = np.isclose(got, want, rtol=0, atol=1e-6)      # float32 sphere centres: a handful may sit in the neighbouring voxel
    assert ok.mean() > 0.995


def test_forward_only_steps_accept_dense_time_grids():
    """Posterior sampling on a time grid the reverse pass could not hold in LDS (N = 400 at M = 30): forward-only calls
    must not ask for the reverse kernel's resources; a training step on that grid reports the documented limit."""
    from vgpmp_amd import engine
    pb = small_problem(robot="franka", S=8, N=12, M=30, B=64, seed=3, n_grid=24)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    pl = engine.PlannerBatch(sc, pb["y"][None], num_samples=8, num_inducing=30, num_data=12, num_bases=64,
                             lengthscales=[2.0] * 7, variance=0.2)
    Xnew = orc.init_trainset(400, 7)
    mean, best, samples, idx = pl.sample_from_posterior(20, Xnew)
    assert mean.shape == (1, 400, 7) and bool(torch.isfinite(samples).all())
    dense = pl.posterior_sampler(20, Xnew)
    with pytest.raises(ValueError, match="VGPMP_E_SHAPE"):
        dense.step()
