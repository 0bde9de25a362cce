"""The covariance and KL dispatchers of the reference on the device:

  K_conditioned   kernel_conditioning/multioutput/cond_kernel.py:17-25, cond_kernel.py:19-22
  Kuu / Kuf / Kfu covariances/multioutput/Kuus.py:42-53, Kufs.py:26-34, covariances/Kfus.py:36-42
  prior_kl        kullback_leiblers/prior_kl.py:16-35

Every number comes out of libvgpmp_hip.so: the matrices from vgpmp_cov_matrices (float64 Matern-5/2 / squared exponential),
the KL from the covariance stage of the ELBO step itself (vgpmp_elbo_step with VGPMP_COV_ONLY: cov_a / cov_b, the code the
optimisation runs).  No CPU fallback; results are returned as CPU float64 tensors, as the callers of the reference's
dispatchers would print or plot them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import capi
from .. import engine


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise capi.VgpmpError("the covariance dispatchers run on the device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(a, dev) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64).to(dev).contiguous()


def _points(v) -> np.ndarray:
    """[n, L] time stamps of an inducing variable (its Zy: conditioned + inducing points) or of a plain array."""
    v = getattr(v, "inducing_variable", v)
    return np.asarray(v.Zy if hasattr(v, "Zy") else v, dtype=np.float64)


def kernel_kind(kernel) -> int:
    name = type(kernel).__name__
    if name == "Matern52":
        return 0
    if name == "SquaredExponential":
        return 1
    raise NotImplementedError(f"device kernels exist for Matern52 and SquaredExponential, not {name}")


def cov_matrices(kernels, Z: np.ndarray, X: np.ndarray, jitter: float = 0.0) -> torch.Tensor:
    """[L, |Z|, |X|]: latent l's kernel on column l of Z and X (vgpmp_cov_matrices)."""
    lib = capi.load(require=True)
    dev = _device()
    kinds = {kernel_kind(k) for k in kernels}
    if len(kinds) != 1:
        raise NotImplementedError("one kernel family per model")
    L = len(kernels)
    Z, X = np.asarray(Z, dtype=np.float64).reshape(-1, L), np.asarray(X, dtype=np.float64).reshape(-1, L)
    Zd, Xd = _dev(Z, dev), _dev(X, dev)
    ell = _dev([float(k.lengthscales) for k in kernels], dev)
    var = _dev([float(k.variance) for k in kernels], dev)
    out = torch.empty((L, Z.shape[0], X.shape[0]), dtype=torch.float64, device=dev)
    capi.check(lib.vgpmp_cov_matrices(kinds.pop(), capi.ptr(Zd), Z.shape[0], capi.ptr(Xd), X.shape[0], L, capi.ptr(ell),
                                      capi.ptr(var), float(jitter), capi.ptr(out), capi.stream_ptr()), "vgpmp_cov_matrices")
    return out.cpu()


def kernel_matrix(kernel, X, X2=None) -> torch.Tensor:
    """k(X, X2) of one single-output kernel on 1-D inputs: [|X|, |X2|]."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 1)
    X2 = X if X2 is None else np.asarray(X2, dtype=np.float64).reshape(-1, 1)
    return cov_matrices([kernel], X, X2)[0]


def K_conditioned(Z, X, kernel) -> torch.Tensor:
    kernels = kernel.kernels if hasattr(kernel, "kernels") else None
    if kernels is None:
        return kernel_matrix(kernel, _points(Z), _points(X))
    return cov_matrices(kernels, _points(Z), _points(X))


def Kuu(inducing_variable, kernel, *, jitter: float = 0.0) -> torch.Tensor:
    if type(kernel).__name__ == "FirstOrderKernelDerivativeSeparateIndependent":
        from . import derivatives
        iv = getattr(inducing_variable, "inducing_variable", inducing_variable)
        return derivatives.velocity_kuu_kuf(iv, kernel, iv.Zy, jitter)[0]
    Zy = _points(inducing_variable)
    return cov_matrices(kernel.kernels, Zy, Zy, jitter)


def Kuf(inducing_variable, kernel, Xnew) -> torch.Tensor:
    if type(kernel).__name__ == "FirstOrderKernelDerivativeSeparateIndependent":
        from . import derivatives
        iv = getattr(inducing_variable, "inducing_variable", inducing_variable)
        return derivatives.velocity_kuu_kuf(iv, kernel, Xnew, 0.0)[1]
    return cov_matrices(kernel.kernels, _points(inducing_variable), _points(Xnew))


def Kfu(inducing_variable, kernel, Xnew) -> torch.Tensor:
    return Kuf(inducing_variable, kernel, Xnew).transpose(-1, -2)


def prior_kl(inducing_variable, kernel, q_mu, q_sqrt, query_states, jitter: float = engine.JITTER) -> torch.Tensor:
    """KL(q || p) with the prior mean conditioned on the two end points (kullback_leiblers/prior_kl.py:16-35).
    q_mu [M, L], q_sqrt [L, M, M] (lower), query_states [2, L] in unconstrained space.  Runs the covariance stage of the
    ELBO step (cov_a / cov_b: Kuu, Cholesky, whitening, per-latent KL) for one problem and sums its per-latent result."""
    lib = capi.load(require=True)
    dev = _device()
    kernels = kernel.kernels
    if any(kernel_kind(k) != 0 for k in kernels):
        raise NotImplementedError("the ELBO path is built for Matern-5/2 (models/vgpmp.py:139)")
    Zy = _points(inducing_variable)
    Mz, L = Zy.shape
    M = Mz - 2
    f64 = torch.float64
    qm = _dev(np.asarray(q_mu, dtype=np.float64).T[None], dev)                          # [1, L, M]
    qs = _dev(np.tril(np.asarray(q_sqrt, dtype=np.float64))[None], dev)                 # [1, L, M, M]
    ell = np.array([float(k.lengthscales) for k in kernels])
    var = np.array([float(k.variance) for k in kernels])
    raw_ell = _dev(engine.softplus_inverse(ell)[None], dev)
    raw_var = _dev(engine.softplus_inverse(var - engine.VARIANCE_FLOOR)[None], dev)
    y_u = _dev(np.asarray(query_states, dtype=np.float64)[None], dev)                   # [1, 2, L]
    X = _dev(np.zeros((1, L)), dev)
    Zd = _dev(Zy, dev)
    dims = capi.Dims(1, 1, 1, 1, M, L, 16, 1, 0, 0)
    nbytes = C.c_size_t(0)
    capi.check(lib.vgpmp_workspace_bytes(C.byref(dims), C.byref(nbytes)), "vgpmp_workspace_bytes")
    ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
    params = capi.Params(capi.ptr(qm), capi.ptr(qs), capi.ptr(raw_ell), capi.ptr(raw_var))
    problem = capi.Problem(capi.ptr(X), capi.ptr(Zd), capi.ptr(y_u), 0.0, float(jitter), 1.0, None, None)
    noise, out = capi.Noise(), capi.Outputs()
    capi.check(lib.vgpmp_elbo_step(C.byref(dims), None, None, C.byref(problem), C.byref(params), None, None, C.byref(noise),
                                   C.byref(out), capi.ptr(ws), ws.numel(), capi.DO_FORWARD | capi.COV_ONLY, 0, 0.0, 1, 0, 0, 0,
                                   capi.stream_ptr()), "vgpmp_elbo_step(COV_ONLY)")
    p, n, dbl = C.c_void_p(), C.c_size_t(), C.c_int32()
    capi.check(lib.vgpmp_workspace_view(C.byref(dims), capi.ptr(ws), b"kl_l", C.byref(p), C.byref(n), C.byref(dbl)),
               "vgpmp_workspace_view")
    off = p.value - ws.data_ptr()
    kl_l = ws[off:off + 8 * n.value].view(f64).cpu()
    return kl_l.sum()
