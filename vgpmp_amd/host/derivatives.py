"""Velocity-constrained kernel variant of the reference on the device: the K_grad / K_grad_grad dispatchers
(derivatives/dispatch.py, first_order.py:14-29, second_order.py:27-58, multioutput/*.py) and the constrained
Kuu / Kuf (covariances/multioutput/Kuus.py:17-39, Kufs.py:14-23).  Unreachable from VGPMP.initialize in the
reference; provided so that the plugin surface is whole.  Everything is computed by libvgpmp_hip
(vgpmp_kernel_derivative, vgpmp_velocity_kuu_kuf): no CPU fallback."""
import ctypes as C

import numpy as np
import torch

from .. import capi
from .model import Matern52, SeparateIndependent, SquaredExponential
from .shims import gpflow


def _kind(kernel) -> int:
    if isinstance(kernel, Matern52):
        return 0
    if isinstance(kernel, SquaredExponential):
        return 1
    raise NotImplementedError(f"derivative kernels exist for Matern52 and SquaredExponential, not {type(kernel).__name__}")


def _dev(a) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise capi.VgpmpError("the derivative kernels run on the device (no CPU fallback)")
    return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64).to("cuda").contiguous()


def _pairwise(order: int, x, y, kernel) -> torch.Tensor:
    lib = capi.load(require=True)
    xd, yd = _dev(np.ravel(x)), _dev(np.ravel(y))
    out = torch.empty((xd.numel(), yd.numel()), dtype=torch.float64, device=xd.device)
    capi.check(lib.vgpmp_kernel_derivative(_kind(kernel), order, capi.ptr(xd), xd.numel(), capi.ptr(yd), yd.numel(),
                                           float(kernel.lengthscales), float(kernel.variance), capi.ptr(out),
                                           capi.stream_ptr()), "vgpmp_kernel_derivative")
    return out.cpu()


def K_grad(x, y, kernel) -> torch.Tensor:
    """d k(x, y) / d y.  Multi-output kernels: column l of x / y with kernel l, stacked [L, len(x), len(y)]."""
    if isinstance(kernel, SeparateIndependent):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        return torch.stack([_pairwise(1, x[..., i], y[..., i], k) for i, k in enumerate(kernel.kernels)], 0)
    return _pairwise(1, x, y, kernel)


def K_grad_grad(x, y_or_kernel, kernel=None) -> torch.Tensor:
    """d^2 k / dx dy.  Two-argument form (Z, multi-output kernel) as in derivatives/multioutput/second_order.py:
    [L, len(Z), len(Z)] plus gpflow.default_jitter() on the diagonal."""
    if kernel is None:
        mk, Z = y_or_kernel, np.asarray(x, dtype=np.float64)
        K = torch.stack([_pairwise(2, Z[..., i], Z[..., i], k) for i, k in enumerate(mk.kernels)], 0)
        return K + gpflow.default_jitter() * torch.eye(K.shape[-1], dtype=K.dtype)
    return _pairwise(2, x, y_or_kernel, kernel)


def velocity_kuu_kuf(inducing_variable, kernel, Xnew, jitter: float):
    """(Kuu [L, Mz + 2, Mz + 2], Kuf [L, Mz + 2, N]) of FirstOrderKernelDerivativeSeparateIndependent."""
    lib = capi.load(require=True)
    kinds = {_kind(k) for k in kernel.kernels}
    if len(kinds) != 1:
        raise NotImplementedError("one kernel family per model")
    Zy = _dev(inducing_variable.Zy if hasattr(inducing_variable, "Zy") else inducing_variable)
    X = _dev(Xnew.Zy if hasattr(Xnew, "Zy") else Xnew)
    Mz, L = Zy.shape
    N = X.shape[0]
    assert X.shape[1] == L == len(kernel.kernels)
    ell = _dev([float(k.lengthscales) for k in kernel.kernels])
    var = _dev([float(k.variance) for k in kernel.kernels])
    Kuu = torch.empty((L, Mz + 2, Mz + 2), dtype=torch.float64, device=Zy.device)
    Kuf = torch.empty((L, Mz + 2, N), dtype=torch.float64, device=Zy.device)
    capi.check(lib.vgpmp_velocity_kuu_kuf(kinds.pop(), capi.ptr(Zy), capi.ptr(X), Mz, N, L, capi.ptr(ell), capi.ptr(var),
                                          float(jitter), capi.ptr(Kuu), capi.ptr(Kuf), capi.stream_ptr()),
               "vgpmp_velocity_kuu_kuf")
    return Kuu.cpu(), Kuf.cpu()
