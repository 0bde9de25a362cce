"""Velocity-constrained kernel variant (SURVEY f-4) on the device against the oracle (which tests/test_oracle_derivatives.py
pins on automatic differentiation, as the reference's tests/unit_test.py does)."""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [orc.KIND_MATERN52, orc.KIND_SE])
def test_pairwise_derivative_kernels_match_oracle(kind):
    from gpflow_vgpmp.derivatives.dispatch import K_grad, K_grad_grad
    from gpflow_vgpmp.kernels.kernels import Matern52, SquaredExponential
    rng = np.random.default_rng(0)
    x, y = rng.uniform(0, 1, 19), rng.uniform(0, 1, 23)
    y[:5] = x[:5]                                              # exact coincidences: the r == 0 branch
    ell, var = 0.7, 1.3
    kern = (Matern52 if kind == orc.KIND_MATERN52 else SquaredExponential)(ell, var)
    np.testing.assert_allclose(K_grad(x, y, kern).numpy(), orc.k_grad(x, y, ell, var, kind), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(K_grad_grad(x, y, kern).numpy(), orc.k_grad_grad(x, y, ell, var, kind), rtol=1e-13, atol=1e-15)
    # the reference's own checks (tests/unit_test.py:8-54) through the dispatchers: first order at x = [1, 2, 3], y = [2, 3, 4]
    a, b = np.array([1.0, 2.0, 3.0]), np.array([2.0, 3.0, 4.0])
    np.testing.assert_allclose(K_grad(a, b, kern).numpy(), orc.k_grad(a, b, ell, var, kind), rtol=1e-13)


def test_velocity_constrained_kuu_kuf_match_oracle():
    from gpflow_vgpmp.covariances import Kuu, Kuf
    from gpflow_vgpmp.inducing_variables.inducing_variables import (ConditionedVariableInducingPoints,
                                                                    SharedIndependentInducingVariables)
    from gpflow_vgpmp.kernels.kernels import FirstOrderKernelDerivativeSeparateIndependent, Matern52
    L, M, N = 3, 6, 11
    Z = np.tile(np.linspace(0.1, 0.9, M)[:, None], (1, L))
    iv = SharedIndependentInducingVariables(ConditionedVariableInducingPoints(Z, np.stack([np.zeros(L), np.ones(L)])))
    ell, var = [2.0, 3.0, 0.7], [0.3, 0.5, 1.1]
    kern = FirstOrderKernelDerivativeSeparateIndependent([Matern52(e, v) for e, v in zip(ell, var)])
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    wuu, wuf = orc.velocity_kuu_kuf(Zy, X, ell, var, jitter=1e-6)
    K = Kuu(iv, kern, jitter=1e-6)
    assert K.shape == (L, M + 4, M + 4)
    np.testing.assert_allclose(K.numpy(), wuu, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Kuf(iv, kern, X).numpy(), wuf, rtol=1e-13, atol=1e-15)
    # the constrained Kuu is what the model would factorise: symmetric up to the sign convention of the cross blocks
    np.testing.assert_allclose(K[:, 2:, 2:].numpy(), np.swapaxes(K[:, 2:, 2:].numpy(), 1, 2), rtol=1e-14)
    np.testing.assert_allclose(K[:, :2, 2:].numpy(), -np.swapaxes(K[:, 2:, :2].numpy(), 1, 2), rtol=1e-14, atol=1e-300)
