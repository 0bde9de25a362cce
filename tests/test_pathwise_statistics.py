"""Statistical pin of the GP half (the reference holds no vector for it).

Independent of the restatement's code path: the first two moments of the pathwise samples
    f_s(X) = f0_s(X) + A (u_s - f0_s(Z) - sqrt(jitter) eps'_s),   A = K_XZ (K_ZZ + jitter I)^-1,
follow in closed form from the published algorithms alone (GPflowSampling's decoupled / Matheron update with a
random-Fourier-feature prior whose spectral draw is the Student-t of a Matern-5/2 kernel):

    E f      = A m
    Cov f    = Kt_XX - Kt_XZ A^T - A Kt_ZX + A (Kt_ZZ + C C^T + jitter I) A^T

with m, C = q_mu, q_sqrt (models/vgpmp.py:200-218), K the Matern-5/2 kernel on the 1-D times and Kt the kernel the
FEATURES realise in expectation: the same Matern-5/2 at distance sqrt(D) |t - t'| (the prior sees the D-vector
t 1_D, SURVEY A5's quirk).  A wrong spectral draw (degrees of freedom, scaling), feature normalisation, update
algebra or q_sqrt assembly shows up here; Monte-Carlo error sets the tolerance."""
import math

import numpy as np
import pytest

from oracle import vgpmp_oracle as orc

JIT = 1e-6


def m52(a, b, ell, var, scale=1.0):
    r = scale * np.abs(a[:, None] - b[None, :]) / ell
    return var * (1.0 + math.sqrt(5.0) * r + 5.0 / 3.0 * r * r) * np.exp(-math.sqrt(5.0) * r)


def closed_form(t, z, ell, var, m, C, D):
    Kzz = m52(z, z, ell, var) + JIT * np.eye(len(z))
    A = np.linalg.solve(Kzz, m52(z, t, ell, var)).T
    s = math.sqrt(D)
    Ktxx, Ktxz, Ktzz = m52(t, t, ell, var, s), m52(t, z, ell, var, s), m52(z, z, ell, var, s)
    cov = Ktxx - Ktxz @ A.T - A @ Ktxz.T + A @ (Ktzz + C @ C.T + JIT * np.eye(len(z))) @ A.T
    return A @ m, cov


def check_moments(samples, mean, cov, tag):
    """samples [n, N]; tolerances from the Monte-Carlo error of n draws (6 standard errors + 1 % model slack)."""
    n = samples.shape[0]
    emean, ecov = samples.mean(0), np.cov(samples.T)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(emean - mean) <= 6.0 * sd / math.sqrt(n) + 1e-3), (tag, np.abs(emean - mean).max())
    se = np.sqrt((np.outer(np.diag(cov), np.diag(cov)) + cov ** 2) / n)
    assert np.all(np.abs(ecov - cov) <= 6.0 * se + 0.01 * np.abs(cov).max()), (tag, np.abs(ecov - cov).max())


def test_oracle_pathwise_moments_match_closed_form():
    rng = np.random.default_rng(7)
    L = D = 3
    N, M, B, S, T = 7, 5, 256, 400, 60
    Mz = M + 2
    X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
    ell, var = np.array([1.3, 2.0, 0.7]), np.array([0.3, 0.25, 0.6])
    p = orc.Params(q_mu=rng.standard_normal((M, L)), q_sqrt=np.tril(0.4 * rng.standard_normal((L, M, M)) + np.eye(M)),
                   raw_ell=orc.softplus_inverse(ell), raw_var=orc.softplus_inverse(var - orc.VARIANCE_FLOOR))
    y_u = rng.standard_normal((2, L))
    cv = orc.cov_forward(p, X, Zy, y_u)
    pts = np.concatenate([X, Zy], axis=0)
    draws, priors = [], []
    for _ in range(T):
        nz = orc.draw_noise(rng, S, L, D, B, Mz)
        Phi = orc.rff_features(nz, pts, cv['ell'], cv['var'])
        F0 = np.matmul(nz.w.transpose(1, 0, 2), Phi.transpose(0, 2, 1)).transpose(1, 0, 2)
        u = cv['m'][None] + np.einsum('lmk,skl->slm', cv['C'], nz.eps)
        R = u - F0[:, :, N:] - math.sqrt(JIT) * nz.eps2.transpose(0, 2, 1)
        draws.append(F0[:, :, :N] + np.einsum('lnm,slm->sln', cv['A'], R))
        priors.append(F0[:, :, :N])
    f, f0 = np.concatenate(draws, axis=0), np.concatenate(priors, axis=0)       # [T*S, L, N]
    for l in range(L):
        mean, cov = closed_form(X[:, l], Zy[:, l], ell[l], var[l], cv['m'][l], cv['C'][l], D)
        check_moments(f[:, l, :], mean, cov, f"oracle latent {l}")
        # the prior draws alone: Matern-5/2 at distance sqrt(D) |t - t'| (the posterior moments above barely
        # depend on the prior's lengthscale; this check does)
        check_moments(f0[:, l, :], np.zeros(N), m52(X[:, l], X[:, l], ell[l], var[l], math.sqrt(D)), f"oracle prior {l}")


@pytest.mark.gpu
def test_device_pathwise_moments_match_closed_form():
    import torch
    from helpers import small_problem
    from vgpmp_amd import engine
    pb = small_problem(robot="franka", S=8, N=9, M=5, B=64, seed=4, n_grid=24)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    S, N, M, B, T = 256, 9, 5, 256, 80
    ell = [1.3, 2.0, 0.7, 1.0, 2.5, 1.6, 0.9]
    pl = engine.PlannerBatch(sc, pb["y"][None], num_samples=S, num_inducing=M, num_data=N, num_bases=B,
                             lengthscales=ell, variance=0.3, seed=11)
    rng = np.random.default_rng(3)
    Q = np.tril(0.4 * rng.standard_normal((1, 7, M, M)) + np.eye(M))
    pl.q_sqrt.copy_(torch.as_tensor(Q))
    pl.q_mu.add_(torch.as_tensor(0.3 * rng.standard_normal((1, 7, M))).to(pl.q_mu.device))
    draws, priors = [], []
    for k in range(T):
        pl.elbo(generate=True, step=k)                                  # device Philox noise of step k
        draws.append(pl.f[0].cpu().numpy().astype(np.float64))          # [S, L, N]
        F0 = pl.view("F0").reshape(pl.dims.split_k, S, 7, N + M + 2).sum(0)
        priors.append(F0[:, :, :N].cpu().numpy().astype(np.float64))
    f, f0 = np.concatenate(draws, axis=0), np.concatenate(priors, axis=0)
    X, Zy = pl.X.cpu().numpy(), pl.Zy.cpu().numpy()
    var = float(pl.variances()[0, 0])
    y_u = pl.y_u[0].cpu().numpy()
    for l in (0, 3, 6):
        m = np.concatenate([y_u[:, l], pl.q_mu[0, l].cpu().numpy()])
        Kzz = m52(Zy[:, l], Zy[:, l], ell[l], var) + JIT * np.eye(M + 2)
        Qp = np.zeros((M + 2, M + 2)); Qp[2:, 2:] = Q[0, l]
        C = np.linalg.cholesky(Kzz) @ Qp + np.diag([JIT, JIT] + [0.0] * M)           # vgpmp.py:208-218
        mean, cov = closed_form(X[:, l], Zy[:, l], ell[l], var, m, C, 7)
        check_moments(f[:, l, :], mean, cov, f"device latent {l}")
        check_moments(f0[:, l, :], np.zeros(N), m52(X[:, l], X[:, l], ell[l], var, math.sqrt(7.0)), f"device prior {l}")
