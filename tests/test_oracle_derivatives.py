"""The velocity-constrained kernel variant (SURVEY f-4): the oracle's derivative kernels against automatic
differentiation, the way the reference's own tests check them (tests/unit_test.py:8-54, TensorFlow tapes there,
torch.autograd here), plus the block layout of the constrained Kuu / Kuf."""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc


def _kern(x, y, ell, var, kind):
    d = (x[:, None] - y[None, :]) / ell
    if kind == orc.KIND_SE:
        return var * torch.exp(-0.5 * d * d)
    r = torch.sqrt(torch.clamp(d * d, min=1e-36))
    return var * (1 + 5 ** 0.5 * r + 5.0 / 3.0 * r * r) * torch.exp(-5 ** 0.5 * r)


@pytest.mark.parametrize("kind,ell,var", [(orc.KIND_MATERN52, 2.0, 0.8), (orc.KIND_SE, 1.5, 0.5), (orc.KIND_MATERN52, 0.4, 3.0)])
def test_first_order_matches_autograd(kind, ell, var):
    # unit_test.py:8-28: x = [1, 2, 3], y = [2, 3, 4], gradient wrt y
    x = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    y = torch.tensor([2.0, 3.0, 4.0], dtype=torch.float64, requires_grad=True)
    want = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            (g,) = torch.autograd.grad(_kern(x[i:i + 1], y[j:j + 1], ell, var, kind).sum(), y)
            want[i, j] = g.sum()
    np.testing.assert_allclose(orc.k_grad(x.numpy(), y.detach().numpy(), ell, var, kind), want, rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("kind,ell,var", [(orc.KIND_SE, 1.5, 0.5), (orc.KIND_MATERN52, 1.5, 0.5)])
def test_second_order_matches_autograd(kind, ell, var):
    # unit_test.py:31-54: y = x + 1e-5 (autodiff of Matern-5/2 is singular at r = 0)
    x = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, requires_grad=True)
    y = (torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64) + 1e-5).requires_grad_()
    want = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            k = _kern(x[i:i + 1], y[j:j + 1], ell, var, kind).sum()
            (gx,) = torch.autograd.grad(k, x, create_graph=True)
            (gxy,) = torch.autograd.grad(gx.sum(), y)
            want[i, j] = gxy.sum()
    got = orc.k_grad_grad(x.detach().numpy(), y.detach().numpy(), ell, var, kind)
    np.testing.assert_allclose(got, want, rtol=1e-5)


def test_second_order_zero_distance_quirk_and_blocks():
    ell, var = np.array([2.0, 0.7]), np.array([0.3, 1.4])
    # r == 0: the reference substitutes 5/3 / ell^2 (no variance), second_order.py:45
    d = orc.k_grad_grad(np.array([0.0, 1.0]), np.array([0.0, 1.0]), ell[0], var[0])
    np.testing.assert_allclose(np.diag(d), (5.0 / 3.0) / ell[0] ** 2)
    M, N, L = 5, 7, 2
    Zy, X = orc.inducing_Zy(M, L), orc.init_trainset(N, L)
    Kuu, Kuf = orc.velocity_kuu_kuf(Zy, X, ell, var, jitter=1e-6)
    assert Kuu.shape == (L, M + 4, M + 4) and Kuf.shape == (L, M + 4, N)
    for l in range(L):
        np.testing.assert_allclose(Kuu[l, 2:, 2:], orc.matern52(Zy[:, l], Zy[:, l], ell[l], var[l]) + 1e-6 * np.eye(M + 2), rtol=1e-14)
        np.testing.assert_allclose(Kuu[l, :2, 2:], -Kuu[l, 2:, :2].T, rtol=1e-14, atol=1e-300)      # dk(x, y) = -dk(y, x)
        np.testing.assert_allclose(Kuf[l, 2:], orc.matern52(Zy[:, l], X[:, l], ell[l], var[l]), rtol=1e-14)
        assert Kuu[l, 0, 0] == pytest.approx((5.0 / 3.0) / ell[l] ** 2 + 2e-6)
