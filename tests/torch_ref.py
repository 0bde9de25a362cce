"""Independent float64 PyTorch (CPU) restatement of the ELBO used ONLY to check the oracle's
analytic reverse pass: everything is differentiated by torch.autograd, the way the reference
lets TensorFlow differentiate it (utils/miscellaneous.py:77-80), with the SDF lookup wrapped in a
custom function that returns the central-difference gradient (likelihood.py:146-176)."""
import math

import torch

from oracle import vgpmp_oracle as orc

T = lambda a: torch.as_tensor(a, dtype=torch.float64)


class SDFLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rel_pos, table, origin, delta):
        q = (rel_pos - origin) / delta
        idx = torch.trunc(q).to(torch.int64)
        hi = torch.tensor(table.shape[:3]) - 1
        idx = torch.minimum(torch.maximum(idx, torch.zeros_like(idx)), hi)
        v = table[idx[..., 0], idx[..., 1], idx[..., 2]]
        ctx.save_for_backward(v[..., 1:])
        return v[..., 0]

    @staticmethod
    def backward(ctx, up):
        (g,) = ctx.saved_tensors
        return up[..., None] * g, None, None, None


def dh(theta, d, a, alpha, craig):
    ct, st = torch.cos(theta), torch.sin(theta)
    ca, sa = torch.cos(alpha), torch.sin(alpha)
    z, o = torch.zeros_like(ct), torch.ones_like(ct)
    if craig:
        rows = [ct, -st, z, a + z, st * ca, ct * ca, -sa + z, -d * sa + z,
                st * sa, ct * sa, ca + z, d * ca + z, z, z, z, o]
    else:
        rows = [ct, -st * ca, st * sa, a * ct, st, ct * ca, -ct * sa, a * st,
                z, sa + z, ca + z, d + z, z, z, z, o]
    return torch.stack(rows, -1).reshape(theta.shape + (4, 4))


def log_prob(scene: orc.Scene, g, sigma=None):
    rb = scene.robot
    A = dh(g + T(rb.twist), T(rb.dh[:, 0]), T(rb.dh[:, 1]), T(rb.dh[:, 2]), rb.craig)
    frames = [T(rb.base_pose).expand(g.shape[:-1] + (4, 4))]
    for i in range(rb.dof):
        frames.append(frames[-1] @ A[..., i, :, :])
    frames = torch.stack(frames, -3)
    Tp = frames[..., torch.as_tensor(rb.sphere_frame, dtype=torch.int64), :, :]
    off = torch.cat([T(rb.sphere_offsets), torch.ones(rb.num_spheres, 1, dtype=torch.float64)], -1)
    pos = (Tp @ off[..., None])[..., :3, 0]
    table = T(orc.sdf_gradient_table(scene.sdf))
    d = SDFLookup.apply(pos - T(scene.offset), table, T(scene.sdf.origin), scene.sdf.delta) - T(rb.radii)
    cost = torch.clamp(scene.epsilon - d, min=0.0)
    return -0.5 * (cost * cost / (T(scene.sigma_obs) if sigma is None else sigma)).sum(-1)


def matern52(t1, t2, ell, var):
    r2 = ((t1[:, None] - t2[None, :]) / ell) ** 2
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    return var * (1 + math.sqrt(5) * r + 5.0 / 3.0 * r2) * torch.exp(-math.sqrt(5) * r)


def elbo(params: orc.Params, scene: orc.Scene, X, Zy, y, noise: orc.Noise, alpha, jitter=orc.JITTER, lik=None, raw_Z=None):
    """Returns (elbo tensor, leaf tensors dict).  With `lik` (orc.LikParams) alpha and sigma_obs are functions of
    the extra leaves raw_alpha / raw_sigma (positive(lower) bijectors of GPflow: lower + softplus)."""
    rb = scene.robot
    leaves = dict(q_mu=T(params.q_mu).clone().requires_grad_(), q_sqrt=T(params.q_sqrt).clone().requires_grad_(),
                  raw_ell=T(params.raw_ell).clone().requires_grad_(), raw_var=T(params.raw_var).clone().requires_grad_())
    sigma = None
    if lik is not None:
        leaves["raw_alpha"] = T(lik.raw_alpha).clone().requires_grad_()
        leaves["raw_sigma"] = T(lik.raw_sigma).clone().requires_grad_()
        alpha = orc.ALPHA_FLOOR + torch.nn.functional.softplus(leaves["raw_alpha"])
        sigma = orc.SIGMA_FLOOR + torch.nn.functional.softplus(leaves["raw_sigma"])
    ell = torch.nn.functional.softplus(leaves["raw_ell"])
    var = orc.VARIANCE_FLOOR + torch.nn.functional.softplus(leaves["raw_var"])
    X = T(X)
    if raw_Z is not None:       # inducing locations as a leaf: Zy = [0; 1; 0.09 + 0.82 sigmoid(raw_Z)]  (models/vgpmp.py:29-42)
        leaves["raw_Z"] = T(raw_Z).clone().requires_grad_()
        Zv = orc.Z_LOW + (orc.Z_HIGH - orc.Z_LOW) * torch.sigmoid(leaves["raw_Z"])
        D = Zv.shape[1]
        Zy = torch.cat([torch.zeros(1, D, dtype=torch.float64), torch.ones(1, D, dtype=torch.float64), Zv], 0)
    else:
        Zy = T(Zy)
    low, high = T(rb.low), T(rb.high)
    y01 = (T(y) - low) / (high - low)
    y_u = torch.log(y01) - torch.log1p(-y01)
    L, M = params.q_sqrt.shape[0], params.q_sqrt.shape[1]
    Mz, N = Zy.shape[0], X.shape[0]
    B = noise.omega.shape[1]
    eye = torch.eye(Mz, dtype=torch.float64)
    fs, kl = [], 0.0
    pts = torch.cat([X, Zy], 0)
    for l in range(L):
        K = matern52(Zy[:, l], Zy[:, l], ell[l], var[l]) + jitter * eye
        Lk = torch.linalg.cholesky(K)
        Q = torch.tril(leaves["q_sqrt"][l])
        Qp = torch.nn.functional.pad(Q, (2, 0, 2, 0))
        jm = torch.zeros(Mz, dtype=torch.float64); jm[:2] = jitter
        C = Lk @ Qp + torch.diag(jm)
        m = torch.cat([y_u[:, l], leaves["q_mu"][:, l]])
        u = m[None] + T(noise.eps[:, :, l]) @ C.T                                   # [S, Mz]
        arg = (pts / ell[l]) @ T(noise.omega[l]).T + T(noise.beta[l])[None]
        Phi = torch.sqrt(2 * var[l] / B) * torch.cos(arg)                            # [J, B]
        F0 = T(noise.w[:, l, :]) @ Phi.T                                             # [S, J]
        K0 = matern52(Zy[:, l], Zy[:, l], ell[l], var[l])
        Lu = torch.linalg.cholesky(K0 + jitter * eye)
        err = u - F0[:, N:] - math.sqrt(jitter) * T(noise.eps2[:, :, l])
        v = torch.cholesky_solve(err.T, Lu).T                                        # [S, Mz]
        Kfu = matern52(X[:, l], Zy[:, l], ell[l], var[l])
        fs.append(F0[:, :N] + v @ Kfu.T)
        # prior_kl.py:16-35
        p_mu = K[:, :2] @ torch.cholesky_solve(y_u[:, l][:, None], Lk[:2, :2])
        wd = torch.linalg.solve_triangular(Lk, (m[:, None] - p_mu), upper=False)[2:, 0]
        kl = kl + 0.5 * ((wd ** 2).sum() - M - torch.log(torch.diagonal(Q) ** 2).sum() + (Q ** 2).sum())
    f = torch.stack(fs, -1)                                                          # [S, N, L]
    g = low + (high - low) * torch.sigmoid(f)
    logp = log_prob(scene, g, sigma)
    return alpha * logp.mean(0).sum() - kl, leaves, dict(f=f, g=g, logp=logp, kl=kl)
