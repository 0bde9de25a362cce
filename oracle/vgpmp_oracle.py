"""CPU oracle for the vGPMP ELBO hot path (float64 NumPy).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``vgpmp_amd``) never does and fails loudly without its HIP library.

This is a restatement, function by function, of the reference's ELBO inner loop
(paths below are relative to /root/reference).  The arithmetic that the
reference delegates to GPflow 2.2 / GPflowSampling / TensorFlow (not present in
the tree, not installable here) is restated from the published algorithms:
Matern-5/2, random Fourier features with a Student-t spectral draw, the
decoupled/Matheron exact update, ``gauss_kl`` with a white prior, Keras Adam.

PINNING STATUS
  * FK (A8): pinned against the reference's own numpy FK
    (gpflow_vgpmp/utils/robot_mixin.py:32-58, imported in the build container by
    tests/golden/make_golden.py) and the six UR10 DH matrices held by
    tests/test_robot.py:14-42.
  * SDF lookup / central-difference gradient (A9): pinned against the
    reference's numpy twins (gpflow_vgpmp/utils/sdf_utils.py:56-60,68-71,78-98)
    and its text parser (:195-210), executed in the build container.
  * Pathwise sampling, KL, ELBO, Adam (A2-A7, A11-A13): PARITY UNPINNED -- the
    reference holds no test or golden vector for them and GPflow/GPflowSampling/
    TF cannot run here.  Their gradients are pinned only against torch.autograd
    (float64) applied to an independent restatement in tests/, and the first two
    moments of the pathwise samples against the closed forms of the published
    algorithms (tests/test_pathwise_statistics.py): a statistical check, not a
    vector from the reference.  The formulas shared with third-party code that
    IS importable here are checked against it (tests/test_oracle_independent.py):
    Matern-5/2 and the conditional mean against scikit-learn, the KL against
    torch.distributions, the exact update against its interpolation identity.

Notation: S samples, N time points, D = L dof/latents, M inducing, Mz = M + 2,
P spheres, B Fourier bases.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

JITTER = 1e-6  # gpflow default_jitter(); benchmarking.py:11 builds a Config but never installs it
SQRT5 = math.sqrt(5.0)


# ----------------------------------------------------------------------------
# Robot description + forward kinematics (A8)
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class RobotTable:
    """What utils/sampler.py:28-56 keeps from the robot config and the Robot."""
    name: str
    dh: np.ndarray              # [D, 3] columns d, a, alpha  (config.yaml dh_parameters)
    twist: np.ndarray           # [D]
    craig: bool                 # craig_dh_convention
    base_pose: np.ndarray       # [4, 4]
    fk_slice: np.ndarray        # [F] frame index (0..D) of each sphere-carrying frame
    spheres_per_link: np.ndarray  # [F]
    sphere_offsets: np.ndarray  # [P, 3] translation of each sphere in its frame
    radii: np.ndarray           # [P]
    joint_limits: np.ndarray    # [D, 2] columns (high, low) as stored by the reference

    @property
    def dof(self) -> int:
        return self.dh.shape[0]

    @property
    def num_spheres(self) -> int:
        return self.sphere_offsets.shape[0]

    @property
    def sphere_frame(self) -> np.ndarray:
        """Frame index for each sphere: gather(fk_slice) + repeat (sampler.py:237-244)."""
        return np.repeat(self.fk_slice, self.spheres_per_link)

    @property
    def low(self) -> np.ndarray:
        return self.joint_limits[:, 1]

    @property
    def high(self) -> np.ndarray:
        return self.joint_limits[:, 0]


def dh_matrix_classic(theta, d, a, alpha):
    """utils/sampler.py:142-168 / robot_mixin.py:60-84 (Spong convention)."""
    ct, st, ca, sa = np.cos(theta), np.sin(theta), np.cos(alpha), np.sin(alpha)
    z, o = np.zeros_like(ct), np.ones_like(ct)
    h = np.stack([ct, -st * ca, st * sa, a * ct,
                  st, ct * ca, -ct * sa, a * st,
                  z, sa + z, ca + z, d + z,
                  z, z, z, o], axis=-1)
    return h.reshape(h.shape[:-1] + (4, 4))


def dh_matrix_craig(theta, d, a, alpha):
    """utils/sampler.py:190-214 / robot_mixin.py:114-131 (modified/Craig convention)."""
    ct, st, ca, sa = np.cos(theta), np.sin(theta), np.cos(alpha), np.sin(alpha)
    z, o = np.zeros_like(ct), np.ones_like(ct)
    h = np.stack([ct, -st, z, a + z,
                  st * ca, ct * ca, -sa + z, -d * sa + z,
                  st * sa, ct * sa, ca + z, d * ca + z,
                  z, z, z, o], axis=-1)
    return h.reshape(h.shape[:-1] + (4, 4))


def forward_kinematics(robot: RobotTable, q: np.ndarray) -> np.ndarray:
    """Cumulative frames T_0..T_D for a batch of joint vectors.

    utils/sampler.py:103-120: T_0 = base_pose, T_i = T_{i-1} @ A_i(q_i + twist_i).
    q: [..., D] -> [..., D+1, 4, 4]
    """
    q = np.asarray(q, dtype=np.float64)
    fn = dh_matrix_craig if robot.craig else dh_matrix_classic
    A = fn(q + robot.twist, robot.dh[:, 0], robot.dh[:, 1], robot.dh[:, 2])  # [..., D, 4, 4]
    out = np.empty(q.shape[:-1] + (robot.dof + 1, 4, 4))
    out[..., 0, :, :] = robot.base_pose
    for i in range(robot.dof):
        out[..., i + 1, :, :] = out[..., i, :, :] @ A[..., i, :, :]
    return out


def sphere_positions(robot: RobotTable, q: np.ndarray, frames: Optional[np.ndarray] = None) -> np.ndarray:
    """utils/sampler.py:216-244: (T_link(p) @ Trans(off_p))[:3, 3].  q [..., D] -> [..., P, 3]."""
    if frames is None:
        frames = forward_kinematics(robot, q)
    T = frames[..., robot.sphere_frame, :, :]                       # [..., P, 4, 4]
    return np.einsum('...pij,pj->...pi', T[..., :3, :3], robot.sphere_offsets) + T[..., :3, 3]


def fk_backward(robot: RobotTable, frames: np.ndarray, pos: np.ndarray, gpos: np.ndarray) -> np.ndarray:
    """d(sum gpos . pos)/dq via the geometric Jacobian (what TF autodiff of the matrix
    chain yields).  Joint i turns about z of frame i-1 (classic) or frame i (Craig)."""
    D = robot.dof
    sf = robot.sphere_frame
    gq = np.zeros(pos.shape[:-2] + (D,))
    mom = np.cross(pos, gpos)                                        # [..., P, 3]
    for i in range(1, D + 1):
        sel = sf >= i
        if not sel.any():
            continue
        F = gpos[..., sel, :].sum(-2)
        Mo = mom[..., sel, :].sum(-2)
        ax = frames[..., i if robot.craig else i - 1, :, :]
        z, o = ax[..., :3, 2], ax[..., :3, 3]
        gq[..., i - 1] = np.einsum('...i,...i->...', z, Mo - np.cross(o, F))
    return gq


# ----------------------------------------------------------------------------
# Signed distance field (A9)
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class SDFGrid:
    data: np.ndarray    # [nx, ny, nz] float64, data[x, y, z]  (sdf_utils.py:25-33)
    origin: np.ndarray  # [3]
    delta: float


def parse_sdf_text(path: str) -> SDFGrid:
    """utils/sdf_utils.py:195-210: header 'nx ny nz' / 'x0 y0 z0' / 'delta', then one value
    per line with x fastest, then y, then z."""
    with open(path, "r") as fh:
        nx, ny, nz = map(int, fh.readline().split())
        x0, y0, z0 = map(float, fh.readline().split())
        delta = float(fh.readline().strip())
        vals = np.loadtxt(fh, dtype=np.float64).reshape(-1)
    data = vals.reshape(nz, ny, nx).transpose(2, 1, 0).copy()
    return SDFGrid(data, np.array([x0, y0, z0]), delta)


def write_sdf_text(path: str, sdf: SDFGrid) -> None:
    nx, ny, nz = sdf.data.shape
    with open(path, "w") as fh:
        fh.write(f"{nx} {ny} {nz}\n")
        fh.write(" ".join(repr(float(v)) for v in sdf.origin) + "\n")
        fh.write(repr(float(sdf.delta)) + "\n")
        np.savetxt(fh, sdf.data.transpose(2, 1, 0).reshape(-1), fmt="%.17g")


def sdf_index(sdf: SDFGrid, rel_pos: np.ndarray) -> np.ndarray:
    """utils/sdf_utils.py:62-66: clip(int64_trunc((p - origin) / delta), 0, n-1)."""
    q = (np.asarray(rel_pos, dtype=np.float64) - sdf.origin) / sdf.delta
    idx = np.trunc(q).astype(np.int64)               # tf.cast float->int64 truncates toward zero
    hi = np.array(sdf.data.shape, dtype=np.int64) - 1
    return np.clip(idx, 0, hi)


def sdf_distance(sdf: SDFGrid, rel_pos: np.ndarray) -> np.ndarray:
    """utils/sdf_utils.py:73-76."""
    i = sdf_index(sdf, rel_pos)
    return sdf.data[i[..., 0], i[..., 1], i[..., 2]]


def sdf_gradient(sdf: SDFGrid, rel_pos: np.ndarray, replace_zero: bool = True) -> np.ndarray:
    """Clamped central difference.  replace_zero=True is the TF path used by the likelihood
    (sdf_utils.py:100-136: components exactly 0 become 0.1); False is the numpy twin (:78-98)."""
    i = sdf_index(sdf, rel_pos)
    hi = np.array(sdf.data.shape, dtype=np.int64) - 1
    n1 = np.clip(i + 1, 0, hi)
    n2 = np.clip(i - 1, 0, hi)
    d = sdf.data
    gx = (d[n1[..., 0], i[..., 1], i[..., 2]] - d[n2[..., 0], i[..., 1], i[..., 2]]) / (2 * sdf.delta)
    gy = (d[i[..., 0], n1[..., 1], i[..., 2]] - d[i[..., 0], n2[..., 1], i[..., 2]]) / (2 * sdf.delta)
    gz = (d[i[..., 0], i[..., 1], n1[..., 2]] - d[i[..., 0], i[..., 1], n2[..., 2]]) / (2 * sdf.delta)
    g = np.stack([gx, gy, gz], axis=-1)
    if replace_zero:
        g = np.where(g == 0, 0.1, g)
    return g


def sdf_gradient_table(sdf: SDFGrid) -> np.ndarray:
    """Per-voxel {d, gx, gy, gz} table [nx, ny, nz, 4]: what the likelihood reads for a
    query that lands in voxel (i,j,k).  Depends on the index only, so it is precomputable."""
    d = sdf.data
    out = np.empty(d.shape + (4,))
    out[..., 0] = d
    for ax in range(3):
        n = d.shape[ax]
        ip = np.minimum(np.arange(n) + 1, n - 1)
        im = np.maximum(np.arange(n) - 1, 0)
        g = (np.take(d, ip, axis=ax) - np.take(d, im, axis=ax)) / (2 * sdf.delta)
        out[..., 1 + ax] = np.where(g == 0, 0.1, g)
    return out


# ----------------------------------------------------------------------------
# Collision likelihood (A9, A10) and its VJP
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class Scene:
    robot: RobotTable
    sdf: SDFGrid
    offset: np.ndarray      # [3] scene position subtracted from sphere centres (likelihood.py:160)
    sigma_obs: np.ndarray   # [P]  (likelihood.py:37-41 -- used un-squared)
    epsilon: float


def joint_sigmoid(robot: RobotTable, f: np.ndarray) -> np.ndarray:
    """likelihoods/likelihood.py:49-52: tfb.Sigmoid(low=limits[:,1], high=limits[:,0])."""
    return robot.low + (robot.high - robot.low) / (1.0 + np.exp(-f))


def joint_sigmoid_inverse(robot: RobotTable, g: np.ndarray) -> np.ndarray:
    x = (np.asarray(g, dtype=np.float64) - robot.low) / (robot.high - robot.low)
    return np.log(x) - np.log1p(-x)


def log_prob(scene: Scene, g: np.ndarray, want_grad: bool = False, lookup_pos: Optional[np.ndarray] = None):
    """likelihood.py:57-176.  g [..., D] joint angles -> logp [...] (and dlogp/dg).

    lookup_pos [..., P, 3] (tests only): sphere centres at which the VOXELS are looked up instead of this function's own float64
    centres.  The nearest-voxel field is piecewise constant, so a float32 implementation whose centre lies within rounding of a
    cell boundary may read the neighbouring voxel; handing its centres in here makes both sides read the same voxels, and the
    comparison then measures arithmetic, not boundary luck.  Everything else (hinge, sums, the FK Jacobian of the reverse pass)
    stays this function's own float64."""
    rb = scene.robot
    frames = forward_kinematics(rb, g)
    pos = sphere_positions(rb, g, frames)
    rel = (pos if lookup_pos is None else np.asarray(lookup_pos, dtype=np.float64)) - scene.offset
    dist = sdf_distance(scene.sdf, rel) - rb.radii
    cost = np.maximum(scene.epsilon - dist, 0.0)                   # likelihood.py:131-143
    logp = -0.5 * np.sum(cost * cost / scene.sigma_obs, axis=-1)   # likelihood.py:99
    if not want_grad:
        return logp
    grad_d = sdf_gradient(scene.sdf, rel, replace_zero=True)       # custom VJP, likelihood.py:166-174
    # dlogp/dd = +cost/sigma (cost = eps - d where active)
    gpos = (cost / scene.sigma_obs)[..., None] * grad_d
    return logp, fk_backward(rb, frames, pos, gpos)


# ----------------------------------------------------------------------------
# GP algebra (A2-A7, A11)
# ----------------------------------------------------------------------------
def matern52(t1: np.ndarray, t2: np.ndarray, ell: float, var: float) -> np.ndarray:
    """gpflow.kernels.Matern52 on 1-D inputs (cond_kernel.py:19-22 feeds column l only)."""
    r = np.abs(t1[:, None] - t2[None, :]) / ell
    r = np.sqrt(np.maximum(r * r, 1e-36))
    return var * (1.0 + SQRT5 * r + 5.0 / 3.0 * r * r) * np.exp(-SQRT5 * r)


def matern52_dell(t1, t2, ell, var):
    r = np.abs(t1[:, None] - t2[None, :]) / ell
    return var * np.exp(-SQRT5 * r) * (5.0 * r * r / (3.0 * ell)) * (1.0 + SQRT5 * r)


def softplus(x):
    return np.logaddexp(0.0, x)


def softplus_inverse(y):
    return np.log(np.expm1(y))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


VARIANCE_FLOOR = 0.1      # models/vgpmp.py:139 positive(lower=1e-1)


@dataclasses.dataclass
class Params:
    """Unconstrained variables, as Adam sees them (utils/miscellaneous.py:77-82)."""
    q_mu: np.ndarray       # [M, L]   identity transform (vgpmp.py:256)
    q_sqrt: np.ndarray     # [L, M, M] lower triangle (FillTriangular is a permutation of it)
    raw_ell: np.ndarray    # [L] lengthscale = softplus(raw)
    raw_var: np.ndarray    # [L] variance = 0.1 + softplus(raw)

    def copy(self):
        return Params(*(np.array(a, dtype=np.float64, copy=True) for a in
                        (self.q_mu, self.q_sqrt, self.raw_ell, self.raw_var)))


def init_params(robot: RobotTable, y: np.ndarray, num_inducing: int, lengthscales: Sequence[float],
                variance: float) -> Params:
    """models/vgpmp.py:166-171,255-263: q_mu[i] = start + (end-start) i/M mapped through the
    inverse joint sigmoid; q_sqrt = I.  A variance sitting on the 0.1 floor (UR10/industrial)
    would map to -inf; it is lifted by 1e-6 (documented deviation)."""
    M, L = num_inducing, robot.dof
    q = np.stack([y[0] + (y[1] - y[0]) * i / M for i in range(M)])
    var = max(variance, VARIANCE_FLOOR + 1e-6)
    return Params(q_mu=joint_sigmoid_inverse(robot, q),
                  q_sqrt=np.tile(np.eye(M), (L, 1, 1)),
                  raw_ell=softplus_inverse(np.asarray(lengthscales, dtype=np.float64)),
                  raw_var=np.full(L, softplus_inverse(var - VARIANCE_FLOOR)))


def init_trainset(n: int, dof: int, end_time: float = 1.0) -> np.ndarray:
    """utils/miscellaneous.py:115-127 with scale=1: rows t * 1_D, t = linspace(0, 1, n)."""
    return np.tile(np.linspace(0.0, end_time, n)[:, None], (1, dof))


def inducing_Zy(num_inducing: int, dof: int) -> np.ndarray:
    """models/vgpmp.py:37-42 + inducing_variables.py:73-82: Zy = [0; 1; linspace(.1,.9,M)] * 1_D."""
    z = np.concatenate([[0.0, 1.0], np.linspace(0.1, 0.9, num_inducing)])
    return np.tile(z[:, None], (1, dof))


@dataclasses.dataclass
class Noise:
    """All randomness of one ELBO evaluation (redrawn every step, vgpmp.py:281)."""
    omega: np.ndarray   # [L, B, D] Student-t spectral draw (already scaled by rsqrt(gamma))
    beta: np.ndarray    # [L, B]    U(0, 2pi)
    w: np.ndarray       # [S, L, B] N(0,1) prior weights
    eps: np.ndarray     # [S, Mz, L] N(0,1) for u = q_mu + q_sqrt eps
    eps2: np.ndarray    # [S, Mz, L] N(0,1) jitter perturbation of the exact update


def draw_noise(rng: np.random.Generator, S, L, D, B, Mz) -> Noise:
    z = rng.standard_normal((L, B, D))
    chi = rng.standard_normal((L, B, 5))
    gam = (chi * chi).sum(-1) / 5.0                      # Gamma(5/2, rate 5/2) == chi^2_5 / 5
    return Noise(omega=z / np.sqrt(gam)[..., None],
                 beta=rng.uniform(0.0, 2.0 * math.pi, (L, B)),
                 w=rng.standard_normal((S, L, B)),
                 eps=rng.standard_normal((S, Mz, L)),
                 eps2=rng.standard_normal((S, Mz, L)))


def constrained(p: Params):
    return softplus(p.raw_ell), VARIANCE_FLOOR + softplus(p.raw_var)


def cov_forward(p: Params, X, Zy, y_u, jitter=JITTER) -> Dict[str, np.ndarray]:
    """Per-latent covariance path: Kuu/Kuf (A2), q_mu/q_sqrt assembly (A3), KL (A11)."""
    ell, var = constrained(p)
    L, M = p.q_sqrt.shape[0], p.q_sqrt.shape[1]
    Mz, N = Zy.shape[0], X.shape[0]
    K = np.empty((L, Mz, Mz)); Lk = np.empty_like(K); Kinv = np.empty_like(K)
    Kuf = np.empty((L, Mz, N)); A = np.empty((L, N, Mz)); C = np.empty((L, Mz, Mz))
    m = np.concatenate([y_u, p.q_mu], axis=0).T.copy()               # [L, Mz]  vgpmp.py:200-202
    jm = np.zeros(Mz); jm[:2] = jitter
    kl = 0.0
    a_full = np.empty((L, Mz)); cvec = np.empty((L, 2))
    for l in range(L):
        K[l] = matern52(Zy[:, l], Zy[:, l], ell[l], var[l])
        Kj = K[l] + jitter * np.eye(Mz)
        Lk[l] = np.linalg.cholesky(Kj)
        Kinv[l] = np.linalg.solve(Kj, np.eye(Mz))
        Kuf[l] = matern52(Zy[:, l], X[:, l], ell[l], var[l])
        A[l] = np.linalg.solve(Kj, Kuf[l]).T
        Qp = np.zeros((Mz, Mz)); Qp[2:, 2:] = np.tril(p.q_sqrt[l])
        C[l] = Lk[l] @ Qp + np.diag(jm)                              # vgpmp.py:208-218
        # prior_kl.py:16-35
        cvec[l] = np.linalg.solve(Kj[:2, :2], y_u[:, l])
        p_mu = Kj[:, :2] @ cvec[l]
        a_full[l] = np.linalg.solve(Lk[l], m[l] - p_mu)
        a = a_full[l, 2:]
        Q = np.tril(p.q_sqrt[l])
        kl += 0.5 * (a @ a - M - np.sum(np.log(np.diag(Q) ** 2)) + np.sum(Q * Q))
    return dict(ell=ell, var=var, K=K, Lk=Lk, Kinv=Kinv, Kuf=Kuf, A=A, C=C, m=m, kl=kl,
                a_full=a_full, cvec=cvec)


def rff_features(noise: Noise, pts: np.ndarray, ell, var, want_dell=False):
    """[3P] gpflow_sampling random_fourier: phi_l(x) = sqrt(2 var_l / B) cos((x/ell_l) . omega_l + beta_l).
    pts [J, D] -> Phi [L, J, B] (and dPhi/dell)."""
    B = noise.omega.shape[1]
    proj = np.einsum('jd,lbd->ljb', pts, noise.omega)
    arg = proj / ell[:, None, None] + noise.beta[:, None, :]
    c = np.sqrt(2.0 * var / B)[:, None, None]
    Phi = c * np.cos(arg)
    if not want_dell:
        return Phi
    dPhi = c * np.sin(arg) * proj / (ell ** 2)[:, None, None]
    return Phi, dPhi


def elbo_forward(p: Params, scene: Scene, X, Zy, y, noise: Noise, alpha: float,
                 jitter=JITTER, want_dell=True, lookup_pos: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """models/vgpmp.py:265-289 with injected randomness.  y [2, D] start/goal joints.
    lookup_pos [S, N, P, 3]: see log_prob (tests: the voxels a float32 implementation read)."""
    rb = scene.robot
    y_u = joint_sigmoid_inverse(rb, y)                                 # vgpmp.py:75-76
    cv = cov_forward(p, X, Zy, y_u, jitter)
    N, Mz = X.shape[0], Zy.shape[0]
    S = noise.w.shape[0]
    pts = np.concatenate([X, Zy], axis=0)                              # J = N + Mz
    if want_dell:
        Phi, dPhi = rff_features(noise, pts, cv['ell'], cv['var'], True)
        H = np.matmul(noise.w.transpose(1, 0, 2), dPhi.transpose(0, 2, 1)).transpose(1, 0, 2)
    else:
        Phi = rff_features(noise, pts, cv['ell'], cv['var']); H = None
    F0 = np.matmul(noise.w.transpose(1, 0, 2), Phi.transpose(0, 2, 1)).transpose(1, 0, 2)  # [S, L, J]
    u = cv['m'][None] + np.einsum('lmk,skl->slm', cv['C'], noise.eps)  # [S, L, Mz]
    R = u - F0[:, :, N:] - math.sqrt(jitter) * noise.eps2.transpose(0, 2, 1)
    f = F0[:, :, :N] + np.einsum('lnm,slm->sln', cv['A'], R)           # [S, L, N]
    fT = f.transpose(0, 2, 1)                                          # [S, N, L]
    g = joint_sigmoid(rb, fT)
    logp, dlogp_dg = log_prob(scene, g, want_grad=True, lookup_pos=lookup_pos)   # [S, N], [S, N, D]
    lik = alpha * logp.mean(0).sum()
    elbo = lik - cv['kl']
    return dict(cv=cv, y_u=y_u, F0=F0, H=H, R=R, f=f, g=g, logp=logp, dlogp_dg=dlogp_dg,
                lik=lik, elbo=elbo, N=N, Mz=Mz, S=S)


def chol_backward(Lc: np.ndarray, Lbar: np.ndarray) -> np.ndarray:
    """Reverse-mode Cholesky (Murray 2016): returns symmetric dK for K = Lc Lc^T."""
    Pm = np.tril(Lc.T @ np.tril(Lbar))
    Pm[np.diag_indices_from(Pm)] *= 0.5
    Sm = np.linalg.solve(Lc.T, np.linalg.solve(Lc.T, Pm.T).T)        # L^-T P L^-1
    return 0.5 * (Sm + Sm.T)


def matern52_dz(t1, t2, ell, var):
    """d k(t1_i, t2_j) / d t1_i for the Matern-5/2 kernel: -var (5/3) (d / ell^2) (1 + sqrt5 r) exp(-sqrt5 r), d = t1_i - t2_j."""
    d = t1[:, None] - t2[None, :]
    r = np.abs(d) / ell
    return -var * (5.0 / 3.0) * d / ell ** 2 * (1.0 + math.sqrt(5.0) * r) * np.exp(-math.sqrt(5.0) * r)


def elbo_backward(p: Params, scene: Scene, X, Zy, noise: Noise, alpha: float, fw, jitter=JITTER, want_z=False):
    """Gradient of loss = -ELBO wrt the unconstrained variables (analytic reverse pass).  With want_z also d loss / d Zy
    [Mz, D] (rows 0, 1 = the conditioned times, constants; rows 2.. = the inducing locations Z of models/vgpmp.py:37-42,
    trainable with trainable_params.inducing_variable): column l through latent l's Kuu / Kuf, every column through the
    random-feature prior of every latent at the rows of Zy."""
    rb = scene.robot
    cv = fw['cv']; N, Mz, S = fw['N'], fw['Mz'], fw['S']
    L, M = p.q_sqrt.shape[0], p.q_sqrt.shape[1]
    ell, var = cv['ell'], cv['var']
    sg = (fw['g'] - rb.low) / (rb.high - rb.low)
    dg_df = (rb.high - rb.low) * sg * (1.0 - sg)
    G = (-(alpha / S) * fw['dlogp_dg'] * dg_df).transpose(0, 2, 1)     # dloss/df [S, L, N]
    g_qmu = np.zeros_like(p.q_mu); g_qs = np.zeros_like(p.q_sqrt)
    g_ell = np.zeros(L); g_var = np.zeros(L)
    g_zy = np.zeros_like(np.asarray(Zy, dtype=np.float64))
    B = noise.omega.shape[1]
    for l in range(L):
        Gl, A, Rl = G[:, l, :], cv['A'][l], fw['R'][:, l, :]
        El = noise.eps[:, :, l]
        Kinv, Lk, K = cv['Kinv'][l], cv['Lk'][l], cv['K'][l]
        dR = Gl @ A                                                   # [S, Mz]
        dm = dR.sum(0)
        dC = dR.T @ El                                                # [Mz, Mz]
        dA = Gl.T @ Rl                                                # [N, Mz]
        F0X, F0Z = fw['F0'][:, l, :N], fw['F0'][:, l, N:]
        # RFF hyper-parameter gradients (prior draws are linear in sqrt(var); H = dF0/dell)
        g_var_l = (np.sum(Gl * F0X) - np.sum(dR * F0Z)) / (2.0 * var[l])
        g_ell_l = 0.0
        if fw['H'] is not None:
            g_ell_l = np.sum(Gl * fw['H'][:, l, :N]) - np.sum(dR * fw['H'][:, l, N:])
        # A = Kfu Kj^-1
        dKfu = dA @ Kinv                                              # [N, Mz]
        dKj = -(A.T @ dA) @ Kinv
        # C = Lk pad(Q) + jitter
        Qp = np.zeros((Mz, Mz)); Qp[2:, 2:] = np.tril(p.q_sqrt[l])
        dLk = dC @ Qp.T
        g_qs[l] = np.tril((Lk.T @ dC)[2:, 2:])
        # KL (+1 * KL in the loss)
        a_full = cv['a_full'][l]
        abar = a_full.copy(); abar[:2] = 0.0                         # d(0.5 a.a)/da_full, a = a_full[2:]
        ddelta = np.linalg.solve(Lk.T, abar)
        dLk += -np.outer(ddelta, a_full)
        dm_kl = ddelta
        dp_mu = -ddelta
        c = cv['cvec'][l]
        dKj[:, :2] += np.outer(dp_mu, c)
        Kj = K + jitter * np.eye(Mz)
        dc = Kj[:, :2].T @ dp_mu
        dKj[:2, :2] += -np.outer(np.linalg.solve(Kj[:2, :2].T, dc), c)
        Q = np.tril(p.q_sqrt[l])
        g_qs[l] += Q - np.diag(1.0 / np.diag(Q))
        dKj += chol_backward(Lk, dLk)
        g_qmu[:, l] = dm[2:] + dm_kl[2:]
        # map dKj, dKfu to hyper-parameters
        z, x = Zy[:, l], X[:, l]
        g_var_l += (np.sum(dKj * K) + np.sum(dKfu * cv['Kuf'][l].T)) / var[l]
        g_ell_l += np.sum(dKj * matern52_dell(z, z, ell[l], var[l])) \
            + np.sum(dKfu * matern52_dell(x, z, ell[l], var[l]))
        g_ell[l] = g_ell_l * sigmoid(p.raw_ell[l])
        g_var[l] = g_var_l * sigmoid(p.raw_var[l])
        if want_z:
            # Kuu[i, j] = k(z_i, z_j): z_i enters row i and column i;  Kfu[n, i] = k(x_n, z_i)
            dk = matern52_dz(z, z, ell[l], var[l])                    # d k(z_i, z_j) / d z_i
            g_zy[:, l] += np.sum((dKj + dKj.T) * dk, axis=1)
            g_zy[:, l] += np.sum(dKfu * matern52_dz(z, x, ell[l], var[l]).T, axis=0)
            # prior draw at the rows of Zy: F0Z[s, i] = sum_b w[s, b] c cos(Zy[i, :] . omega[b, :] / ell + beta[b]),
            # R = u - F0Z - ...  =>  d loss / d F0Z = -dR
            arg = (Zy @ noise.omega[l].T) / ell[l] + noise.beta[l][None]            # [Mz, B]
            Tm = dR.T @ noise.w[:, l, :]                                            # [Mz, B]
            coef = math.sqrt(2.0 * var[l] / B) / ell[l]
            g_zy += coef * (Tm * np.sin(arg)) @ noise.omega[l]                      # [Mz, D]
    if want_z:
        return Params(g_qmu, g_qs, g_ell, g_var), G, g_zy
    return Params(g_qmu, g_qs, g_ell, g_var), G


# ----------------------------------------------------------------------------
# Adam (A13) -- Keras / TF 2.12 semantics
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class AdamState:
    m: Params
    v: Params
    t: int = 0


def adam_init(p: Params) -> AdamState:
    z = lambda a: np.zeros_like(a)
    return AdamState(Params(z(p.q_mu), z(p.q_sqrt), z(p.raw_ell), z(p.raw_var)),
                     Params(z(p.q_mu), z(p.q_sqrt), z(p.raw_ell), z(p.raw_var)))


def adam_step(p: Params, g: Params, st: AdamState, lr: float, trainable: Dict[str, bool],
              beta1=0.8, beta2=0.95, eps=1e-7) -> None:
    """models/vgpmp.py:77 Adam(lr, 0.8, 0.95) on the unconstrained variables, in place."""
    st.t += 1
    lr_t = lr * math.sqrt(1.0 - beta2 ** st.t) / (1.0 - beta1 ** st.t)
    for name, flag in (("q_mu", "q_mu"), ("q_sqrt", "q_sqrt"), ("raw_ell", "lengthscales"),
                       ("raw_var", "kernel_variance")):
        if not trainable.get(flag, True):
            continue
        x, gr = getattr(p, name), getattr(g, name)
        m, v = getattr(st.m, name), getattr(st.v, name)
        m += (gr - m) * (1.0 - beta1)
        v += (gr * gr - v) * (1.0 - beta2)
        x -= lr_t * m / (np.sqrt(v) + eps)


DEFAULT_TRAINABLE = dict(q_mu=True, q_sqrt=True, lengthscales=True, kernel_variance=True)


def optimization_step(p, st, scene, X, Zy, y, noise, alpha, lr, trainable=DEFAULT_TRAINABLE, lookup_pos=None, want_grad=False):
    """utils/miscellaneous.py:68-84: loss = -ELBO, grads, Adam.apply_gradients.  Returns loss (with want_grad: loss, gradient).
    lookup_pos: see log_prob (tests: the voxels a float32 implementation read in this step)."""
    fw = elbo_forward(p, scene, X, Zy, y, noise, alpha, want_dell=trainable.get("lengthscales", True), lookup_pos=lookup_pos)
    g, _ = elbo_backward(p, scene, X, Zy, noise, alpha, fw)
    adam_step(p, g, st, lr, trainable)
    return (-fw['elbo'], g) if want_grad else -fw['elbo']


# ----------------------------------------------------------------------------
# Inducing locations as trainable variables (A14: trainable_params.inducing_variable)
# ----------------------------------------------------------------------------
Z_LOW, Z_HIGH = 0.09, 0.91     # models/vgpmp.py:41  bounded_Z(low=0.09, high=0.91): tfb.Sigmoid(low, high)


def z_constrained(raw_Z: np.ndarray) -> np.ndarray:
    return Z_LOW + (Z_HIGH - Z_LOW) * sigmoid(raw_Z)


def z_unconstrained(Z: np.ndarray) -> np.ndarray:
    x = (np.asarray(Z, dtype=np.float64) - Z_LOW) / (Z_HIGH - Z_LOW)
    return np.log(x) - np.log1p(-x)


def init_raw_Z(num_inducing: int, dof: int) -> np.ndarray:
    """models/vgpmp.py:37-42: Z[m, :] = linspace(0.1, 0.9, M)[m] * 1_D, held in unconstrained space."""
    return z_unconstrained(np.tile(np.linspace(0.1, 0.9, num_inducing)[:, None], (1, dof)))


def zy_from_raw(raw_Z: np.ndarray) -> np.ndarray:
    """inducing_variables.py:73-82: Zy = [conditioned times 0 and 1; Z]."""
    D = raw_Z.shape[1]
    return np.concatenate([np.zeros((1, D)), np.ones((1, D)), z_constrained(raw_Z)], axis=0)


def z_backward(raw_Z: np.ndarray, g_zy: np.ndarray) -> np.ndarray:
    """d loss / d raw_Z from d loss / d Zy (rows 0, 1 are constants)."""
    sg = sigmoid(raw_Z)
    return g_zy[2:] * (Z_HIGH - Z_LOW) * sg * (1.0 - sg)


def optimization_step_z(p, raw_Z, st, st_z, scene, X, y, noise, alpha, lr, trainable,
                        beta1=0.8, beta2=0.95, eps=1e-7):
    """One optimisation step with the inducing locations among the variables (same optimizer, shared step count).
    st_z = dict(m=..., v=...) like raw_Z.  Updates p, raw_Z in place; returns the loss."""
    Zy = zy_from_raw(raw_Z)
    fw = elbo_forward(p, scene, X, Zy, y, noise, alpha, want_dell=trainable.get("lengthscales", True))
    g, _, g_zy = elbo_backward(p, scene, X, Zy, noise, alpha, fw, want_z=True)
    gz = z_backward(raw_Z, g_zy)
    adam_step(p, g, st, lr, trainable)                  # advances st.t
    if trainable.get("inducing_variable", False):
        lr_t = lr * math.sqrt(1.0 - beta2 ** st.t) / (1.0 - beta1 ** st.t)
        st_z["m"] += (gz - st_z["m"]) * (1.0 - beta1)
        st_z["v"] += (gz * gz - st_z["v"]) * (1.0 - beta2)
        raw_Z -= lr_t * st_z["m"] / (np.sqrt(st_z["v"]) + eps)
    return -fw['elbo']


# ----------------------------------------------------------------------------
# Likelihood constants as trainable variables (A14: trainable_params.sigma_obs / alpha)
# ----------------------------------------------------------------------------
ALPHA_FLOOR = 1e-4     # models/vgpmp.py:82          Parameter(alpha, transform=positive(1e-4))
SIGMA_FLOOR = 1e-5     # likelihoods/likelihood.py:31,41  positive(DEFAULT_VARIANCE_LOWER_BOUND)


@dataclasses.dataclass
class LikParams:
    """Unconstrained likelihood variables: alpha = 1e-4 + softplus(raw_alpha), sigma_obs = 1e-5 + softplus(raw_sigma)
    (one entry per sphere, likelihood.py:37-41)."""
    raw_alpha: np.ndarray   # [] (0-d array)
    raw_sigma: np.ndarray   # [P]

    def copy(self):
        return LikParams(np.array(self.raw_alpha, dtype=np.float64, copy=True),
                         np.array(self.raw_sigma, dtype=np.float64, copy=True))


def init_lik_params(alpha: float, sigma_obs: np.ndarray) -> LikParams:
    return LikParams(np.asarray(softplus_inverse(alpha - ALPHA_FLOOR), dtype=np.float64),
                     softplus_inverse(np.asarray(sigma_obs, dtype=np.float64) - SIGMA_FLOOR))


def lik_constrained(lp: LikParams) -> Tuple[float, np.ndarray]:
    return float(ALPHA_FLOOR + softplus(lp.raw_alpha)), SIGMA_FLOOR + softplus(lp.raw_sigma)


def hinge_cost(scene: Scene, g: np.ndarray) -> np.ndarray:
    """likelihood.py:131-143: max(epsilon - d, 0) per sphere, [..., P]."""
    rb = scene.robot
    pos = sphere_positions(rb, g, forward_kinematics(rb, g))
    dist = sdf_distance(scene.sdf, pos - scene.offset) - rb.radii
    return np.maximum(scene.epsilon - dist, 0.0)


def lik_backward(lp: LikParams, scene: Scene, fw) -> LikParams:
    """Gradient of the training loss wrt (raw_alpha, raw_sigma).  utils/miscellaneous.py:324-343 attaches the priors
    Normal(loc=<the parameter itself>, scale) to both: their density is constant in the parameter (loc moves with it),
    so of GPflow's log_prior_density only the bijectors' log|d constrained / d raw| = log sigmoid(raw) remains:
        loss = -(ELBO + log sigmoid(raw_alpha) + sum_p log sigmoid(raw_sigma_p)).
    `scene.sigma_obs` and the alpha passed to elbo_forward must be lik_constrained(lp)."""
    alpha, sigma = lik_constrained(lp)
    S = fw['S']
    cost = hinge_cost(scene, fw['g'])                                   # [S, N, P]
    c2 = (cost * cost).reshape(-1, cost.shape[-1]).sum(0)
    dE_dalpha = fw['logp'].mean(0).sum()                                # ELBO = alpha * sum_n mean_s logp - KL
    dE_dsigma = (alpha / S) * 0.5 * c2 / (sigma * sigma)                # logp = -1/2 sum_p c^2 / sigma_p
    return LikParams(np.asarray(-(dE_dalpha * sigmoid(lp.raw_alpha) + sigmoid(-lp.raw_alpha))),
                     -(dE_dsigma * sigmoid(lp.raw_sigma) + sigmoid(-lp.raw_sigma)))


def optimization_step_lik(p, lp: LikParams, st, st_lik, scene: Scene, X, Zy, y, noise, lr,
                          trainable=DEFAULT_TRAINABLE, beta1=0.8, beta2=0.95, eps=1e-7):
    """optimization_step with sigma_obs / alpha among the variables.  st_lik = dict(m=LikParams, v=LikParams);
    the Adam step count is shared with `st` (one optimizer, models/vgpmp.py:77).  Returns the loss."""
    alpha, sigma = lik_constrained(lp)
    sc = dataclasses.replace(scene, sigma_obs=sigma)
    fw = elbo_forward(p, sc, X, Zy, y, noise, alpha, want_dell=trainable.get("lengthscales", True))
    g, _ = elbo_backward(p, sc, X, Zy, noise, alpha, fw)
    gl = lik_backward(lp, sc, fw)
    adam_step(p, g, st, lr, trainable)
    lr_t = lr * math.sqrt(1.0 - beta2 ** st.t) / (1.0 - beta1 ** st.t)
    for name, flag in (("raw_alpha", "alpha"), ("raw_sigma", "sigma_obs")):
        if not trainable.get(flag, False):
            continue
        x, gr = getattr(lp, name), getattr(gl, name)
        m, v = getattr(st_lik["m"], name), getattr(st_lik["v"], name)
        m += (gr - m) * (1.0 - beta1)
        v += (gr * gr - v) * (1.0 - beta2)
        x -= lr_t * m / (np.sqrt(v) + eps)
    return -(fw['elbo'] + np.log(sigmoid(lp.raw_alpha)) + np.sum(np.log(sigmoid(lp.raw_sigma))))


# ----------------------------------------------------------------------------
# Velocity-constrained kernel variant (SURVEY f-4; unreachable from VGPMP.initialize)
# ----------------------------------------------------------------------------
KIND_MATERN52, KIND_SE = 0, 1


def squared_exponential(t1, t2, ell, var):
    d = (np.asarray(t1, dtype=np.float64)[:, None] - np.asarray(t2, dtype=np.float64)[None, :]) / ell
    return var * np.exp(-0.5 * d * d)


def k_grad(x, y, ell: float, var: float, kind: int = KIND_MATERN52) -> np.ndarray:
    """derivatives/first_order.py:14-29: d k(x, y) / d y  ([len(x), len(y)]).
    Matern-5/2: var 5/3 (1 + sqrt5 r) exp(-sqrt5 r) (x - y) / ell^2;  SE: (x - y) / ell^2 k(x, y)."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    diff = x[:, None] - y[None, :]
    if kind == KIND_SE:
        return diff / ell ** 2 * squared_exponential(x, y, ell, var)
    s5r = SQRT5 * np.abs(diff) / ell
    return (5.0 / 3.0) * (1.0 + s5r) * np.exp(-s5r) * diff / ell ** 2 * var


def k_grad_grad(x, y, ell: float, var: float, kind: int = KIND_MATERN52) -> np.ndarray:
    """derivatives/second_order.py:27-58: d^2 k(x, y) / dx dy.
    Matern-5/2: -var 5/3 (5 r^2 - sqrt5 r - 1) exp(-sqrt5 r) / ell^2 for r != 0; entries that come out exactly 0
    (r == 0, where the reference's dr/dx is divide_no_nan -> 0) are REPLACED by 5/3 / ell^2 -- without the variance,
    as the reference does (second_order.py:45).  SE: (ell^2 - (x - y)^2) / ell^4 k(x, y)."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    diff = x[:, None] - y[None, :]
    if kind == KIND_SE:
        return (ell ** 2 - diff * diff) / ell ** 4 * squared_exponential(x, y, ell, var)
    r = np.abs(diff) / ell
    s5r = SQRT5 * r
    with np.errstate(divide='ignore', invalid='ignore'):
        dr_dx = np.where(r != 0.0, diff / (r * ell ** 2), 0.0)
    res = var * (5.0 / 3.0) * (5.0 * r * r - s5r - 1.0) * np.exp(-s5r) * dr_dx * (-dr_dx)
    return np.where(res == 0.0, (5.0 / 3.0) / ell ** 2, res)


def velocity_kuu_kuf(Zy, X, ell, var, jitter=JITTER, kind: int = KIND_MATERN52):
    """covariances/multioutput/Kuus.py:17-39 and Kufs.py:14-23 for FirstOrderKernelDerivativeSeparateIndependent:
    per latent l, with ny = the two conditioned times Zy[:2, l],
        Kuu = [[d2k(ny, ny) + 1e-6 I, dk(ny, Zy)], [dk(Zy, ny), k(Zy, Zy)]] + jitter I     [Mz + 2, Mz + 2]
        Kuf = [dk(ny, X); k(Zy, X)]                                                      [Mz + 2, N]
    (the 1e-6 on the 2 x 2 block is gpflow.default_jitter() added by the multi-output K_grad_grad,
    derivatives/multioutput/second_order.py:11-19)."""
    Zy = np.asarray(Zy, dtype=np.float64); X = np.asarray(X, dtype=np.float64)
    Mz, L = Zy.shape
    N = X.shape[0]
    kfun = squared_exponential if kind == KIND_SE else matern52
    Kuu = np.empty((L, Mz + 2, Mz + 2)); Kuf = np.empty((L, Mz + 2, N))
    for l in range(L):
        z, ny, x = Zy[:, l], Zy[:2, l], X[:, l]
        Kuu[l, :2, :2] = k_grad_grad(ny, ny, ell[l], var[l], kind) + JITTER * np.eye(2)
        Kuu[l, :2, 2:] = k_grad(ny, z, ell[l], var[l], kind)
        Kuu[l, 2:, :2] = k_grad(z, ny, ell[l], var[l], kind)
        Kuu[l, 2:, 2:] = kfun(z, z, ell[l], var[l])
        Kuu[l] += jitter * np.eye(Mz + 2)
        Kuf[l, :2] = k_grad(ny, x, ell[l], var[l], kind)
        Kuf[l, 2:] = kfun(z, x, ell[l], var[l])
    return Kuu, Kuf


# ----------------------------------------------------------------------------
# Plan extraction (A15)
# ----------------------------------------------------------------------------
def posterior_mean(p: Params, robot: RobotTable, Xnew, Zy, y, jitter=JITTER) -> np.ndarray:
    """models/vgpmp.py:316-317: gpflow conditional, whiten=False: Knm (Kmm + jitter I)^-1 q_mu."""
    y_u = joint_sigmoid_inverse(robot, y)
    cv = cov_forward(p, Xnew, Zy, y_u, jitter)
    mu = np.einsum('lnm,lm->nl', cv['A'], cv['m'])
    return joint_sigmoid(robot, mu)


def sample_from_posterior(p, scene, Xnew, Zy, y, noise: Noise, jitter=JITTER):
    """models/vgpmp.py:312-339: pathwise samples at Xnew, best = argmax_s sum_n logp."""
    fw = elbo_forward(p, scene, Xnew, Zy, y, noise, 1.0, jitter, want_dell=False)
    best = int(np.argmax(fw['logp'].sum(-1)))
    return posterior_mean(p, scene.robot, Xnew, Zy, y, jitter), fw['g'][best], fw['g'], best


# ----------------------------------------------------------------------------
# Philox-4x32-10 counter RNG, identical to the device generator (vgpmp_amd/csrc/philox.h)
# ----------------------------------------------------------------------------
_PM0, _PM1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PW0, _PW1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32 (10 rounds)."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = _PM0 * c[0]; p1 = _PM1 * c[2]
            hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
            c = [(hi1 ^ c[1] ^ np.uint64(k0)) & _MASK32, lo1,
                 (hi0 ^ c[3] ^ np.uint64(k1)) & _MASK32, lo0]
            k0 = np.uint32(k0 + _PW0); k1 = np.uint32(k1 + _PW1)
    return np.stack(c, axis=-1).astype(np.uint32)


def _u01(x: np.ndarray) -> np.ndarray:
    """uint32 -> (0, 1): ((x >> 8) + 0.5) * 2^-24 -- exact in float32 and float64 alike, so the
    device (float32) and this oracle start Box-Muller from identical uniforms."""
    return ((x >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24


def philox_normals(n: int, key: Tuple[int, int], stream: int) -> np.ndarray:
    """n standard normals: element i uses counter (i // 4, stream, 0, 0); lanes (0,1) and (2,3)
    feed two Box-Muller pairs -> 4 normals per counter."""
    nc = (n + 3) // 4
    ctr = np.zeros((nc, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nc, dtype=np.uint32); ctr[:, 1] = np.uint32(stream)
    r = philox4x32(ctr, key)
    u = _u01(r)
    rad0 = np.sqrt(-2.0 * np.log(u[:, 0])); rad1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    t0 = 2.0 * math.pi * u[:, 1]; t1 = 2.0 * math.pi * u[:, 3]
    out = np.stack([rad0 * np.cos(t0), rad0 * np.sin(t0), rad1 * np.cos(t1), rad1 * np.sin(t1)], axis=-1)
    return out.reshape(-1)[:n]


W_TABLE_SIZE = 8192
_w_table_cache = None


def w_table() -> np.ndarray:
    """The 8192 magnitudes of the W stream as float16 (restates tools/make_w_table.py -> csrc/gp_wtable.h): T[i] = mean of |z|,
    z ~ N(0, 1), over the i-th of 8192 equally probable bins of the half-normal distribution,
    16384 (phi(q_i) - phi(q_{i+1})), q_i = Phi^-1(1/2 + i / 16384), rounded to nearest even."""
    global _w_table_cache
    if _w_table_cache is None:
        from scipy.stats import norm
        q = norm.ppf(0.5 + np.arange(W_TABLE_SIZE + 1, dtype=np.float64) / (2 * W_TABLE_SIZE))
        pdf = norm.pdf(q)
        _w_table_cache = (2.0 * W_TABLE_SIZE * (pdf[:-1] - pdf[1:])).astype(np.float16)
    return _w_table_cache


def philox_normals8(n: int, key: Tuple[int, int], stream: int) -> np.ndarray:
    """n prior weights, EIGHT per counter (the W stream; csrc/vgpmp_device.h, "The W stream", csrc/gp_common.h::vg_w8): each 16-bit
    half of the four words of counter i // 8 is one weight -- element 8 i + 2 j + {0, 1} from the {low, high} half of word j --,
    w = +-T[h & 0x1fff], sign from bit 15.  A stratified inverse-CDF draw with 16 384 equally likely float16 values: E[w] = 0,
    Var[w] = 1 - 5e-6, |w| <= 4.074; exact integer / table arithmetic, so the device's weights are these bit for bit."""
    nc = (n + 7) // 8
    ctr = np.zeros((nc, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nc, dtype=np.uint32); ctr[:, 1] = np.uint32(stream)
    r = philox4x32(ctr, key)
    halves = np.stack([r & np.uint32(0xFFFF), r >> np.uint32(16)], axis=-1)                       # [nc, 4, 2]
    mag = w_table().astype(np.float64)[(halves & np.uint32(0x1FFF)).astype(np.int64)]
    out = np.where((halves & np.uint32(0x8000)) != 0, -mag, mag)
    return out.reshape(-1)[:n]


def philox_uniforms(n: int, key: Tuple[int, int], stream: int) -> np.ndarray:
    nc = (n + 3) // 4
    ctr = np.zeros((nc, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nc, dtype=np.uint32); ctr[:, 1] = np.uint32(stream)
    return _u01(philox4x32(ctr, key)).reshape(-1)[:n]


STREAM_OMEGA, STREAM_CHI, STREAM_BETA, STREAM_W, STREAM_EPS, STREAM_EPS2 = 0, 1, 2, 3, 4, 5


def philox_key(seed: int, problem: int, step: int) -> Tuple[int, int]:
    """Key schedule shared with the device: k0 = seed ^ (problem * 0x9E3779B1), k1 = step."""
    return ((seed ^ (problem * 0x9E3779B1)) & 0xFFFFFFFF, step & 0xFFFFFFFF)


def philox_noise(seed: int, problem: int, step: int, S, L, D, B, Mz) -> Noise:
    key = philox_key(seed, problem, step)
    z = philox_normals(L * B * D, key, STREAM_OMEGA).reshape(L, B, D)
    chi = philox_normals(L * B * 8, key, STREAM_CHI).reshape(L, B, 8)[..., :5]
    gam = (chi * chi).sum(-1) / 5.0
    return Noise(omega=z / np.sqrt(gam)[..., None],
                 beta=2.0 * math.pi * philox_uniforms(L * B, key, STREAM_BETA).reshape(L, B),
                 w=philox_normals8(S * L * B, key, STREAM_W).reshape(S, L, B),
                 eps=philox_normals(S * Mz * L, key, STREAM_EPS).reshape(S, Mz, L),
                 eps2=philox_normals(S * Mz * L, key, STREAM_EPS2).reshape(S, Mz, L))


# ----------------------------------------------------------------------------
# Mesh -> signed distance grid (replaces the external SDFGen binary of
# gpflow_vgpmp/utils/gen_sdf.py:16-43).  PARITY UNPINNED against SDFGen (not available): exact
# point-triangle distance, sign from the generalized winding number of each closed part.
# ----------------------------------------------------------------------------
def _closest_point_dist2(p, a, b, c):
    """Squared distance from points p [..., 3] to triangles (a, b, c) [T, 3] -> [..., T] (Ericson 5.1.5)."""
    p = p[..., None, :]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        v_ab = d1 / (d1 - d3)
        w_ac = d2 / (d2 - d6)
        w_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        den = 1.0 / (va + vb + vc)
    q = a + ab * (vb * den)[..., None] + ac * (vc * den)[..., None]             # interior
    q = np.where(((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0))[..., None], b + (c - b) * w_bc[..., None], q)
    q = np.where(((vb <= 0) & (d2 >= 0) & (d6 <= 0))[..., None], a + ac * w_ac[..., None], q)
    q = np.where(((d6 >= 0) & (d5 <= d6))[..., None], c + 0 * p, q)
    q = np.where(((vc <= 0) & (d1 >= 0) & (d3 <= 0))[..., None], a + ab * v_ab[..., None], q)
    q = np.where(((d3 >= 0) & (d4 <= d3))[..., None], b + 0 * p, q)
    q = np.where(((d1 <= 0) & (d2 <= 0))[..., None], a + 0 * p, q)
    return ((p - q) ** 2).sum(-1)


def mesh_signed_distance(vertices, faces, part, points):
    """Signed distance (negative inside) of `points` [..., 3] to the triangle mesh; a point is inside when
    the winding number of any closed part exceeds 1/2 in magnitude."""
    V = np.asarray(vertices, dtype=np.float64)
    F = np.asarray(faces, dtype=np.int64)
    part = np.asarray(part, dtype=np.int64)
    P = np.asarray(points, dtype=np.float64)
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    d2 = _closest_point_dist2(P, a, b, c).min(-1)
    pa, pb_, pc = a - P[..., None, :], b - P[..., None, :], c - P[..., None, :]
    la, lb, lc = (np.linalg.norm(x, axis=-1) for x in (pa, pb_, pc))
    num = (pa * np.cross(pb_, pc)).sum(-1)
    den = la * lb * lc + (pa * pb_).sum(-1) * lc + (pb_ * pc).sum(-1) * la + (pc * pa).sum(-1) * lb
    omega = 2.0 * np.arctan2(num, den)                                            # [..., T]
    inside = np.zeros(P.shape[:-1], dtype=bool)
    for k in range(int(part.max()) + 1):
        inside |= np.abs(omega[..., part == k].sum(-1)) > 2.0 * math.pi          # |w| > 1/2
    return np.where(inside, -1.0, 1.0) * np.sqrt(d2)
