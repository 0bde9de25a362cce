"""The oracle's forward pass and analytic reverse pass against torch.autograd (float64) on an
independent restatement (tests/torch_ref.py).  Parity with GPflow/TF itself is unpinned (no TF here)."""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
import torch_ref
from helpers import small_problem


@pytest.mark.parametrize("robot,problem", [("franka", "industrial"), ("wam", "industrial"), ("ur10", "industrial")])
def test_elbo_and_gradient_match_autograd(robot, problem):
    pb = small_problem(robot=robot, problem=problem, S=5, N=8, M=4, B=32, seed=3)
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], pb["noise"], pb["alpha"])
    grads, _ = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["noise"], pb["alpha"], fw)
    e, leaves, aux = torch_ref.elbo(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], pb["noise"], pb["alpha"])
    assert (fw["logp"] < 0).any(), "fixture must have active hinge terms"
    np.testing.assert_allclose(fw["g"], aux["g"].detach().numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fw["logp"], aux["logp"].detach().numpy(), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(fw["cv"]["kl"], float(aux["kl"].detach()), rtol=1e-9)
    np.testing.assert_allclose(fw["elbo"], float(e), rtol=1e-9)
    (-e).backward()
    for name in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
        want = leaves[name].grad.numpy()
        got = getattr(grads, name)
        if name == "q_sqrt":
            want = np.tril(want)
        scale = np.abs(want).max() + 1e-30
        assert np.abs(got - want).max() / scale < 2e-6, (name, np.abs(got - want).max(), scale)


def test_likelihood_constants_gradient_matches_autograd():
    """trainable_params.sigma_obs / alpha: loss = -(ELBO + log-det-Jacobians of the two positive() bijectors); the
    Normal priors the reference attaches are centred on the parameters themselves and drop out."""
    import dataclasses
    pb = small_problem(robot="franka", S=5, N=8, M=4, B=32, seed=3)
    rng = np.random.default_rng(0)
    P = pb["scene"].robot.num_spheres
    lp = orc.init_lik_params(7.5, 0.004 + 0.01 * rng.uniform(size=P))
    alpha, sigma = orc.lik_constrained(lp)
    np.testing.assert_allclose(alpha, 7.5, rtol=1e-12)
    sc = dataclasses.replace(pb["scene"], sigma_obs=sigma)
    fw = orc.elbo_forward(pb["params"], sc, pb["X"], pb["Zy"], pb["y"], pb["noise"], alpha)
    assert (fw["logp"] < 0).any()
    gl = orc.lik_backward(lp, sc, fw)
    e, leaves, _ = torch_ref.elbo(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], pb["noise"], None, lik=lp)
    np.testing.assert_allclose(fw["elbo"], float(e.detach()), rtol=1e-9)
    loss = -(e + torch.nn.functional.logsigmoid(leaves["raw_alpha"]) + torch.nn.functional.logsigmoid(leaves["raw_sigma"]).sum())
    loss.backward()
    np.testing.assert_allclose(gl.raw_alpha, leaves["raw_alpha"].grad.numpy(), rtol=1e-9)
    np.testing.assert_allclose(gl.raw_sigma, leaves["raw_sigma"].grad.numpy(), rtol=1e-8, atol=1e-12)
    assert np.abs(gl.raw_sigma).max() > 1e-3


def test_fk_backward_is_the_chain_derivative():
    pb = small_problem(robot="wam", S=3, N=4, M=3, B=8, seed=5)
    rb = pb["scene"].robot
    rng = np.random.default_rng(0)
    q = rng.uniform(rb.low, rb.high, (11, rb.dof))
    gpos = rng.standard_normal((11, rb.num_spheres, 3))
    frames = orc.forward_kinematics(rb, q)
    pos = orc.sphere_positions(rb, q, frames)
    got = orc.fk_backward(rb, frames, pos, gpos)
    h = 1e-6
    for j in range(rb.dof):
        dq = np.zeros(rb.dof); dq[j] = h
        num = ((orc.sphere_positions(rb, q + dq) - orc.sphere_positions(rb, q - dq)) / (2 * h) * gpos).sum((-1, -2))
        np.testing.assert_allclose(got[:, j], num, rtol=1e-6, atol=1e-7)


def test_adam_matches_torch_adam_modulo_epsilon_placement():
    """Keras Adam: x -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).  torch.optim.Adam puts eps
    after the bias correction; with eps -> 0 both coincide, which pins the moment recursions."""
    rng = np.random.default_rng(1)
    x0 = rng.standard_normal(7)
    p = orc.Params(q_mu=x0.reshape(7, 1).copy(), q_sqrt=np.eye(1)[None].repeat(1, 0), raw_ell=np.zeros(1), raw_var=np.zeros(1))
    st = orc.adam_init(p)
    xt = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([xt], lr=0.02, betas=(0.8, 0.95), eps=0.0)
    for t in range(5):
        g = rng.standard_normal(7)
        orc.adam_step(p, orc.Params(g.reshape(7, 1), np.zeros((1, 1, 1)), np.zeros(1), np.zeros(1)), st, 0.02,
                      dict(q_mu=True, q_sqrt=False, lengthscales=False, kernel_variance=False), eps=0.0)
        xt.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(p.q_mu[:, 0], xt.detach().numpy(), rtol=1e-10)


def test_philox_known_answer_and_moments():
    """Random123 KAT for philox4x32-10: counter=0,key=0 and the all-ones vector."""
    z = orc.philox4x32(np.zeros((1, 4), dtype=np.uint32), (0, 0))[0]
    assert [int(v) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    o = orc.philox4x32(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [int(v) for v in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    n = orc.philox_normals(200000, orc.philox_key(7, 3, 11), orc.STREAM_W)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01
    # the W stream: eight float16-valued weights per counter out of the 8192-entry table of half-normal bin means (a stratified
    # inverse-CDF draw, 16 384 equally likely values): moments, the bound, independence of the two halves of a word and of
    # neighbouring words, the distance from the normal distribution, and the half-word -> element map
    key = orc.philox_key(7, 3, 11)
    w = orc.philox_normals8(400000, key, orc.STREAM_W)
    assert abs(w.mean()) < 0.005 and abs(w.std() - 1.0) < 0.005
    assert abs((w ** 4).mean() - 3.0) < 0.06 and abs((w ** 3).mean()) < 0.03
    t = orc.w_table()
    assert t.dtype == np.float16 and t.shape == (8192,) and np.all(np.diff(t.astype(np.float64)) >= 0) and t[0] > 0
    assert np.abs(w).max() <= float(t[-1]) == 4.07421875
    assert np.array_equal(w, w.astype(np.float16).astype(np.float64))          # every weight IS a float16
    # the table by itself: exact moments of the 16 384-point distribution, and its Kolmogorov distance from N(0, 1)
    t64 = t.astype(np.float64)
    assert abs((t64 * t64).mean() - 1.0) < 1e-5 and abs(t64.mean() - np.sqrt(2.0 / np.pi)) < 1e-6
    assert abs((t64 ** 4).mean() - 3.0) < 2e-3
    from scipy import stats
    cdf_gap = np.abs(stats.norm.cdf(t64) - (0.5 + (np.arange(8192) + 0.5) / 16384.0)).max()
    assert cdf_gap < 2e-4       # (Kolmogorov distance: half a float16 step times the density, 4.9e-4 x 0.24 near |w| = 1; the bins are finer)
    assert abs(np.corrcoef(w[0::2], w[1::2])[0, 1]) < 0.01 and abs(np.corrcoef(w[:-2:2], w[2::2])[0, 1]) < 0.01
    assert stats.kstest(w[:100000], "norm").pvalue > 1e-3
    r = orc.philox4x32(np.array([[5, orc.STREAM_W, 0, 0]], dtype=np.uint32), key)[0]
    lo, hi = int(r[2]) & 0xFFFF, int(r[2]) >> 16
    want = [(-1.0 if h & 0x8000 else 1.0) * float(t[h & 0x1FFF]) for h in (lo, hi)]
    np.testing.assert_array_equal(w[8 * 5 + 4: 8 * 5 + 6], want)
    # the committed device table (csrc/gp_wtable.h, written by tools/make_w_table.py) is this table, bit for bit
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgpmp_amd", "csrc", "gp_wtable.h")).read()
    bits = np.array([int(v, 16) for v in re.findall(r"0x([0-9a-f]{4})", hdr)], dtype=np.uint16)
    assert np.array_equal(bits, t.view(np.uint16))
    nz = orc.philox_noise(7, 0, 0, S=64, L=3, D=3, B=256, Mz=6)
    np.testing.assert_array_equal(nz.w.reshape(-1), orc.philox_normals8(64 * 3 * 256, orc.philox_key(7, 0, 0), orc.STREAM_W))
    # Student-t(5) spectral draw: variance nu/(nu-2) = 5/3
    assert abs(nz.omega.var() - 5.0 / 3.0) < 0.25 and 0 <= nz.beta.min() and nz.beta.max() <= 2 * np.pi


def test_optimization_reduces_loss():
    pb = small_problem(S=8, N=12, M=5, B=64, seed=2)
    p = pb["params"].copy(); st = orc.adam_init(p)
    rng = np.random.default_rng(0)
    losses = []
    for step in range(12):
        noise = orc.draw_noise(rng, 8, 7, 7, 64, 7)
        losses.append(orc.optimization_step(p, st, pb["scene"], pb["X"], pb["Zy"], pb["y"], noise, pb["alpha"], 0.02))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < np.mean(losses[:3])


@pytest.mark.parametrize("robot", ["franka", "ur10"])
def test_inducing_location_gradient_matches_autograd(robot):
    """trainable_params.inducing_variable (utils/miscellaneous.py:338): Z = 0.09 + 0.82 sigmoid(raw_Z) [M, D]
    (models/vgpmp.py:29-42), every column with its own values.  The oracle's analytic d loss / d raw_Z -- through Kuu, Kuf,
    the Cholesky factor (q_sqrt assembly, exact update, KL) and through the random-feature prior evaluated at the rows of
    Zy -- against torch.autograd on the independent restatement, at perturbed (non-replicated) locations."""
    pb = small_problem(robot=robot, S=5, N=8, M=4, B=32, seed=3)
    D = pb["scene"].robot.dof
    rng = np.random.default_rng(1)
    raw_Z = orc.init_raw_Z(4, D) + 0.3 * rng.standard_normal((4, D))
    np.testing.assert_allclose(orc.zy_from_raw(orc.init_raw_Z(4, D)), pb["Zy"], rtol=0, atol=1e-15)
    Zy = orc.zy_from_raw(raw_Z)
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], Zy, pb["y"], pb["noise"], pb["alpha"])
    grads, _, g_zy = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], Zy, pb["noise"], pb["alpha"], fw, want_z=True)
    gz = orc.z_backward(raw_Z, g_zy)
    e, leaves, aux = torch_ref.elbo(pb["params"], pb["scene"], pb["X"], None, pb["y"], pb["noise"], pb["alpha"], raw_Z=raw_Z)
    assert (fw["logp"] < 0).any(), "fixture must have active hinge terms"
    np.testing.assert_allclose(fw["elbo"], float(e), rtol=1e-9)
    (-e).backward()
    want = leaves["raw_Z"].grad.numpy()
    assert np.abs(want).max() > 1e-3
    assert np.abs(gz - want).max() / np.abs(want).max() < 2e-6, np.abs(gz - want).max()
    # the other variables' gradients are unaffected by how Zy was produced
    for name in ("q_mu", "raw_ell"):
        w2 = leaves[name].grad.numpy()
        assert np.abs(getattr(grads, name) - w2).max() / (np.abs(w2).max() + 1e-30) < 2e-6
