"""The Gaussian-process half of the oracle against INDEPENDENT implementations that are importable here
(GPflow / TensorFlow are not): scikit-learn's Matern(nu=2.5) kernel and GP-regression mean, torch's closed-form
Gaussian KL, and the interpolation property of the Matheron update.  This does not replace a run of the
reference (parity with GPflow itself stays unpinned, see oracle/vgpmp_oracle.py), it pins the formulas the
restatement shares with third-party code: kernel (A2), conditional mean (A7 / A15), KL (A11), exact update (A6)."""
import numpy as np
import pytest
import torch
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import ConstantKernel, Matern

from oracle import vgpmp_oracle as orc
from helpers import small_problem


@pytest.mark.parametrize("ell,var", [(0.3, 0.25), (2.0, 0.1), (7.5, 3.0)])
def test_matern52_matches_scikit_learn(ell, var):
    rng = np.random.default_rng(0)
    t1, t2 = rng.uniform(0, 1, 17), rng.uniform(0, 1, 23)
    want = var * Matern(length_scale=ell, nu=2.5)(t1[:, None], t2[:, None])
    np.testing.assert_allclose(orc.matern52(t1, t2, ell, var), want, rtol=1e-12, atol=1e-15)
    # d/d ell by central differences of the independent kernel
    h = 1e-6 * ell
    fd = var * (Matern(length_scale=ell + h, nu=2.5)(t1[:, None], t2[:, None]) -
                Matern(length_scale=ell - h, nu=2.5)(t1[:, None], t2[:, None])) / (2 * h)
    np.testing.assert_allclose(orc.matern52_dell(t1, t2, ell, var), fd, rtol=2e-6, atol=1e-9)


def test_conditional_mean_matches_scikit_learn_gp_regression():
    """A = Kfu (Kuu + jitter I)^-1 applied to q_mu_full is the GP-regression mean with noise level `jitter`."""
    pb = small_problem(robot="franka", S=4, N=25, M=9, B=16, seed=2)
    p, X, Zy = pb["params"], pb["X"], pb["Zy"]
    y_u = orc.joint_sigmoid_inverse(pb["scene"].robot, pb["y"])
    cv = orc.cov_forward(p, X, Zy, y_u)
    ell, var = orc.constrained(p)
    for l in range(Zy.shape[1]):
        kern = ConstantKernel(var[l], constant_value_bounds="fixed") * Matern(ell[l], length_scale_bounds="fixed", nu=2.5)
        gpr = GaussianProcessRegressor(kernel=kern, alpha=orc.JITTER, optimizer=None).fit(Zy[:, l:l + 1], cv["m"][l])
        want = gpr.predict(X[:, l:l + 1])
        got = cv["A"][l] @ cv["m"][l]
        # the system has condition number ~1e7: two float64 solvers agree to ~1e-8 of the values' scale
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * (np.abs(want).max() + 1.0))


def test_kl_matches_torch_gaussian_kl():
    """prior_kl.py:16-35 in whitened form: the M free points are N(a, Q Q^T) against N(0, I)."""
    pb = small_problem(robot="wam", S=3, N=6, M=7, B=8, seed=4)
    p = pb["params"].copy()
    rng = np.random.default_rng(1)
    L, M = p.q_sqrt.shape[0], p.q_sqrt.shape[1]
    p.q_sqrt = np.tril(0.3 * rng.standard_normal((L, M, M))) + np.tile(np.eye(M), (L, 1, 1))
    p.q_mu = p.q_mu + 0.2 * rng.standard_normal(p.q_mu.shape)
    y_u = orc.joint_sigmoid_inverse(pb["scene"].robot, pb["y"])
    cv = orc.cov_forward(p, pb["X"], pb["Zy"], y_u)
    want = 0.0
    for l in range(L):
        a = torch.tensor(cv["a_full"][l, 2:])
        Q = torch.tensor(np.tril(p.q_sqrt[l]))
        Q = Q * torch.sign(torch.diagonal(Q))[None, :]            # scale_tril wants a positive diagonal; Q Q^T unchanged
        q = torch.distributions.MultivariateNormal(a, scale_tril=Q)
        pr = torch.distributions.MultivariateNormal(torch.zeros(M, dtype=torch.float64), torch.eye(M, dtype=torch.float64))
        want += float(torch.distributions.kl_divergence(q, pr))
    np.testing.assert_allclose(cv["kl"], want, rtol=1e-10)


def test_matheron_update_interpolates_the_inducing_values():
    """exact_update: a path evaluated AT the inducing points returns u - sqrt(jitter) eps' up to the jitter
    regularisation, whatever the prior draw was (Wilson et al. 2020, eq. 13)."""
    pb = small_problem(robot="franka", S=6, N=10, M=8, B=64, seed=6)
    p, Zy, noise = pb["params"], pb["Zy"], pb["noise"]
    fw = orc.elbo_forward(p, pb["scene"], Zy, Zy, pb["y"], noise, pb["alpha"])      # time points := inducing points
    cv = fw["cv"]
    L, Mz = Zy.shape[1], Zy.shape[0]
    for l in range(L):
        u = cv["m"][l][None, :] + noise.eps[:, :, l] @ cv["C"][l].T                   # [S, Mz]
        target = u - np.sqrt(orc.JITTER) * noise.eps2[:, :, l]
        f, f0 = fw["f"][:, l, :], fw["F0"][:, l, :Mz]                                  # [S, Mz] each
        # f = f0 + K (K + jI)^-1 (target - f0): the residual is jitter (K + jI)^-1 (target - f0)
        Kj = cv["K"][l] + orc.JITTER * np.eye(Mz)
        resid = orc.JITTER * np.linalg.solve(Kj, (target - f0).T).T
        np.testing.assert_allclose(f, target - resid, rtol=0, atol=1e-8 * (np.abs(target).max() + 1.0))
        assert np.abs(resid).max() < 0.2 * np.abs(target - f0).max()      # small: only directions with eigenvalue ~jitter remain
