"""The planner model and its likelihood behind the reference's names.

  VariationalMonteCarloLikelihood   likelihoods/likelihood.py:18-176
  VGPMP                             models/vgpmp.py:59-339
  kernels / inducing variables      kernels/kernels.py, inducing_variables/inducing_variables.py
  Kuu / Kuf / Kfu / K_conditioned / prior_kl   covariances/, kernel_conditioning/, kullback_leiblers/

One `VGPMP` owns a one-problem `PlannerBatch`; every evaluation (elbo, optimisation step, posterior
sampling, log_prob, the covariance / KL dispatchers) is a C-ABI call into libvgpmp_hip.so.
"""
from __future__ import annotations

import warnings
from typing import List, Optional

import numpy as np
import torch

from .. import engine
from .shims import Parameter, gpflow


# ---------------------------------------------------------------- plugin classes ---------------------------
class Matern52:
    def __init__(self, lengthscales=1.0, variance=1.0):
        self.lengthscales = lengthscales if isinstance(lengthscales, Parameter) else Parameter(lengthscales, name="lengthscales")
        self.variance = variance if isinstance(variance, Parameter) else Parameter(variance, name="variance")

    def __call__(self, X, X2=None):
        from .covariances import kernel_matrix
        return kernel_matrix(self, X, X2)               # vgpmp_cov_matrices


class SquaredExponential:
    def __init__(self, lengthscales=1.0, variance=1.0):
        self.lengthscales = lengthscales if isinstance(lengthscales, Parameter) else Parameter(lengthscales, name="lengthscales")
        self.variance = variance if isinstance(variance, Parameter) else Parameter(variance, name="variance")

    def __call__(self, X, X2=None):
        from .covariances import kernel_matrix
        return kernel_matrix(self, X, X2)               # vgpmp_cov_matrices


class SeparateIndependent:
    def __init__(self, kernels, name=None):
        self.kernels = list(kernels)
        self.name = name


class VanillaConditioningSeparateIndependent(SeparateIndependent):
    """Marker class the reference dispatches on (kernels/kernels.py:9)."""


class FirstOrderKernelDerivativeSeparateIndependent(SeparateIndependent):
    """Marker class of the velocity-constrained variant (kernels/kernels.py:4-6; unreachable from VGPMP.initialize in
    the reference).  Kuu / Kuf dispatch on it to the derivative blocks computed on the device (host/derivatives.py)."""


class VanillaConditioningSharedIndependent:
    def __init__(self, kernel, output_dim=None, name=None):
        self.kernel, self.output_dim, self.name = kernel, output_dim, name


class ConditionedVariableInducingPoints:
    """Zy = [conditioned time stamps; Z]  (inducing_variables/inducing_variables.py:73-82)."""

    def __init__(self, Z, conditioned_timesteps, name: Optional[str] = None):
        self._Z = Z if isinstance(Z, Parameter) else Parameter(Z, trainable=False, name="Z")
        self.conditioned_timesteps = np.asarray(conditioned_timesteps, dtype=np.float64)
        assert self._Z.numpy().shape[1] == self.conditioned_timesteps.shape[1]
        self.len_ny = 2

    @property
    def ny(self):
        return self.conditioned_timesteps

    @property
    def Zy(self):
        return np.concatenate([self.ny, self._Z.numpy()], axis=0)

    Z = Zy

    @property
    def num_inducing(self):
        return self._Z.numpy().shape[0]

    def __len__(self):
        return self.num_inducing


class SharedIndependentInducingVariables:
    def __init__(self, inducing_variable):
        self.inducing_variable = inducing_variable

    @property
    def num_inducing(self):
        return self.inducing_variable.num_inducing


# ---------------------------------------------------------------- likelihood --------------------------------
class _JointSigmoid:
    """tfb.Sigmoid(low, high) look-alike: callable forward + inverse (likelihood.py:49-52)."""

    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)

    def __call__(self, x):
        if isinstance(x, torch.Tensor):
            lo = torch.as_tensor(self.low, dtype=x.dtype, device=x.device)
            hi = torch.as_tensor(self.high, dtype=x.dtype, device=x.device)
            return lo + (hi - lo) * torch.sigmoid(x)
        x = np.asarray(x, dtype=np.float64)
        return self.low + (self.high - self.low) / (1.0 + np.exp(-x))

    forward = __call__

    def inverse(self, y):
        u = (np.asarray(y, dtype=np.float64) - self.low) / (self.high - self.low)
        return np.log(u) - np.log1p(-u)


class VariationalMonteCarloLikelihood:
    def __init__(self, sigma_obs: float, robot, sampler, sdf, offset: List[float], epsilon: float = 0.05,
                 DEFAULT_VARIANCE_LOWER_BOUND=1e-5, **kwargs):
        self.sdf, self.sampler, self.robot = sdf, sampler, robot
        self.variance = Parameter(np.full((1, robot.num_spheres), float(sigma_obs)), trainable=False, name="sigma_obs")
        self.offset = np.asarray(offset, dtype=np.float64).reshape(1, 3)
        self.sphere_radii = np.asarray(robot.sphere_radii, dtype=np.float64).reshape(1, -1)
        self.joint_constraints = np.asarray(robot.joint_limits, dtype=np.float64).reshape(-1, 2)
        self.velocity_constraints = np.asarray(robot.velocity_limits, dtype=np.float64).reshape(-1, 2)
        self.joint_sigmoid = _JointSigmoid(low=self.joint_constraints[:, 1], high=self.joint_constraints[:, 0])
        self.epsilon = float(epsilon)
        self.p = robot.num_spheres
        # device-resident scene: robot tables + packed voxel table.  The reference builds a new model per
        # start-goal query (utils/miscellaneous.py:162-169); the upload is cached per (sdf, robot, constants)
        key = (id(sdf), id(robot.spec), tuple(self.offset.reshape(3)), float(sigma_obs), self.epsilon)
        cache = sdf.__dict__.setdefault("_device_scenes", {})
        if key not in cache:
            cache[key] = engine.DeviceScene(robot.spec, sdf.grid, self.offset.reshape(3),
                                            sigma_obs=self.variance.numpy().reshape(-1), epsilon=self.epsilon)
        self.device_scene = cache[key]
        sdf.bind(self.device_scene)
        sampler.bind(self.device_scene)

    def log_prob(self, F):
        """log p(e | f) for joint configurations F [S, N, D] -> [S, N]  (likelihood.py:57-99)."""
        F = torch.as_tensor(np.asarray(F.detach().cpu() if isinstance(F, torch.Tensor) else F, dtype=np.float32))
        return self.device_scene.log_prob(F)

    _log_prob = log_prob

    def _signed_distance_grad(self, data):
        """Signed distance (minus sphere radius) of sphere centres [..., P, 3] and its gradient field."""
        rel = torch.as_tensor(np.asarray(data, dtype=np.float64)) - torch.as_tensor(self.offset)
        _, dist, grad = self.device_scene.sdf_query(rel.reshape(-1, 3))
        dist = dist.reshape(rel.shape[:-1]) - torch.as_tensor(self.sphere_radii, dtype=torch.float32, device=dist.device)
        return dist, grad.reshape(rel.shape)

    def _hinge_loss(self, data):
        d, _ = self._signed_distance_grad(data)
        return torch.clamp(self.epsilon - d, min=0.0)


# ---------------------------------------------------------------- the model ---------------------------------
class _Adam:
    def __init__(self, learning_rate, beta_1=0.8, beta_2=0.95, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations = 0


class VGPMP:
    def __init__(self, kernel, likelihood, inducing_variable, num_latent_gps, num_samples, num_bases, num_data,
                 query_states, num_inducing, learning_rate, alpha, q_mu, whiten=False, prior=None, seed: int = 0):
        self.kernel, self.likelihood, self.inducing_variable = kernel, likelihood, inducing_variable
        self.num_latent_gps, self.num_samples, self.num_bases = num_latent_gps, num_samples, num_bases
        self.num_inducing, self.num_data, self.prior = num_inducing, num_data, prior
        self.alpha = Parameter(alpha, trainable=False, name="alpha")
        self.optimizer = _Adam(learning_rate)
        self._y = np.asarray(query_states, dtype=np.float64).reshape(2, num_latent_gps)
        self._query_states = likelihood.joint_sigmoid.inverse(self._y)
        self._seed = seed
        self._planner: Optional[engine.PlannerBatch] = None
        self._init_q_mu = np.asarray(q_mu, dtype=np.float64)
        self._n_train = None
        self.trainable = dict(engine.DEFAULT_TRAINABLE)

    # -- construction (models/vgpmp.py:84-198) ----------------------------------------------------------------
    @classmethod
    def initialize(cls, sdf, robot, sampler, lengthscales, query_states, sigma_obs=0.05, alpha=1.0, variance=0.1,
                   learning_rate=0.1, num_inducing=14, num_samples=51, num_bases=1024, scene_offset=None,
                   num_data=None, num_output_dims=None, kernel=None, num_latent_gps=None, epsilon=0.05, q_mu=None,
                   interpolation_method: Optional[str] = "linear", **kwargs):
        query_states = np.asarray(query_states, dtype=np.float64)
        assert lengthscales is not None, "Lengthscales have not been set."
        assert query_states is not None and len(query_states) == 2, "Must pass a motion plan to initialize the model."
        if num_output_dims is None:
            num_output_dims = query_states[0].shape[-1]
        if num_latent_gps is None:
            num_latent_gps = num_output_dims
        if num_data is None:
            num_data = len(query_states)
        if scene_offset is None:
            scene_offset = [0, 0, 0]
            warnings.warn("Offset has not been set. Defaulting to [0, 0, 0].")
        assert len(lengthscales) == num_latent_gps and num_output_dims == num_latent_gps
        if kernel is None:
            kernel = VanillaConditioningSeparateIndependent(
                [Matern52(lengthscales=float(lengthscales[i]), variance=float(variance)) for i in range(num_latent_gps)])
        else:
            assert isinstance(kernel, SeparateIndependent), "Kernels must be a SeparateIndependent list of Matern52"
        cond = np.stack([np.zeros(num_output_dims), np.ones(num_output_dims)])
        Z = np.tile(np.linspace(0.1, 0.9, num_inducing)[:, None], (1, num_latent_gps))
        iv = SharedIndependentInducingVariables(ConditionedVariableInducingPoints(Z, cond))
        likelihood = VariationalMonteCarloLikelihood(sigma_obs=sigma_obs, robot=robot, sdf=sdf, sampler=sampler,
                                                     offset=scene_offset, epsilon=epsilon)
        q0, q1 = query_states[0].reshape(-1), query_states[1].reshape(-1)
        if q_mu is None:
            if interpolation_method is None:
                q_mu = likelihood.joint_sigmoid(np.zeros((num_inducing, num_latent_gps)))
            elif interpolation_method == "linear":
                q_mu = np.stack([q0 + (q1 - q0) * i / num_inducing for i in range(num_inducing)])
            else:
                raise NotImplementedError(interpolation_method)
        else:
            q_mu = np.asarray(q_mu, dtype=np.float64)
            assert q_mu.shape == (num_inducing, num_latent_gps)
        return cls(kernel=kernel, likelihood=likelihood, inducing_variable=iv, num_latent_gps=num_latent_gps,
                   num_samples=num_samples, num_bases=num_bases, num_data=num_data,
                   query_states=np.stack([q0, q1]), num_inducing=num_inducing, learning_rate=learning_rate,
                   alpha=alpha, q_mu=q_mu, whiten=False)

    # -- device state --------------------------------------------------------------------------------------
    def _ensure(self, n_time: int) -> engine.PlannerBatch:
        if self._planner is None or self._n_train != n_time:
            k = self.kernel.kernels
            pl = engine.PlannerBatch(
                self.likelihood.device_scene, self._y[None], num_samples=self.num_samples,
                num_inducing=self.num_inducing, num_data=n_time, num_bases=self.num_bases,
                lengthscales=[float(kk.lengthscales) for kk in k], variance=float(k[0].variance),
                alpha=float(self.alpha), learning_rate=float(self.optimizer.learning_rate), trainable=self.trainable,
                seed=self._seed)
            pl.raw_var.copy_(torch.tensor(engine.softplus_inverse(
                np.maximum([float(kk.variance) for kk in k], engine.VARIANCE_FLOOR + 1e-6) - engine.VARIANCE_FLOOR))[None])
            pl.q_mu.copy_(torch.tensor(self.likelihood.joint_sigmoid.inverse(self._init_q_mu).T[None]))
            old = self._planner
            if old is not None:
                # a different number of time stamps: the SAME variables evaluated on another X (the reference's variables do
                # not depend on the data) -- carry the parameters, the Adam state and the step count over
                for name in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
                    getattr(pl, name).copy_(getattr(old, name))
                for dst, src in zip(pl.adam_m + pl.adam_v, old.adam_m + old.adam_v):
                    dst.copy_(src)
                if pl.lik_variables and old.lik_variables:
                    pl.raw_alpha.copy_(old.raw_alpha); pl.raw_sigma.copy_(old.raw_sigma)
                    for dst, src in zip(pl.lik_adam_m + pl.lik_adam_v, old.lik_adam_m + old.lik_adam_v):
                        dst.copy_(src)
                if pl.z_variables and old.z_variables:          # trained inducing locations and their Adam moments
                    pl.raw_Z.copy_(old.raw_Z); pl.z_adam_m.copy_(old.z_adam_m); pl.z_adam_v.copy_(old.z_adam_v)
                pl.t = old.t
            self._planner, self._n_train = pl, n_time
        return self._planner

    def _check_data(self, data) -> int:
        data = np.asarray(data, dtype=np.float64)
        assert data.ndim == 2 and data.shape[1] == self.num_latent_gps, "data must be [N, D] time stamps"
        pl = self._planner
        if pl is not None and self._n_train == data.shape[0]:
            pl.set_time_stamps(data)
        return data.shape[0]

    # -- reference surface -----------------------------------------------------------------------------------
    @property
    def query_states(self):
        return self._query_states

    @property
    def _q_mu(self):
        """Unconstrained variational mean [M, L] (models/vgpmp.py:256)."""
        pl = self._planner
        return self.likelihood.joint_sigmoid.inverse(self._init_q_mu) if pl is None else pl.q_mu[0].T.cpu().numpy()

    @property
    def _q_sqrt(self):
        """Lower-triangular factor [L, M, M] (models/vgpmp.py:263)."""
        pl = self._planner
        return np.tile(np.eye(self.num_inducing), (self.num_latent_gps, 1, 1)) if pl is None else pl.q_sqrt[0].cpu().numpy()

    @property
    def q_mu(self):
        return np.concatenate([self.query_states, self._q_mu], axis=0)

    @property
    def q_sqrt(self):
        """Lk pad(_q_sqrt) + jitter diag(1,1,0..) per latent [L, Mz, Mz] (models/vgpmp.py:208-218);
        read back from the last device evaluation."""
        pl = self._ensure(self._n_train or self.num_data)
        pl.elbo(generate=True)
        return pl.view("C").reshape(self.num_latent_gps, pl.Mz, pl.Mz).cpu().numpy().astype(np.float64)

    @property
    def trainable_variables(self):
        pl = self._ensure(self._n_train or self.num_data)
        names = [("q_mu", pl.q_mu), ("q_sqrt", pl.q_sqrt), ("lengthscales", pl.raw_ell), ("kernel_variance", pl.raw_var)]
        out = [t for n, t in names if self.trainable.get(n, True)]
        if pl.lik_variables:                                    # disable_param_opt flags sigma_obs / alpha
            out += [t for n, t in (("sigma_obs", pl.raw_sigma), ("alpha", pl.raw_alpha)) if self.trainable.get(n, False)]
        if pl.z_variables:                                      # inducing_variable: raw_Z behind Sigmoid(0.09, 0.91)
            out.append(pl.raw_Z)
        return out

    def elbo(self, data) -> float:
        """models/vgpmp.py:265-289: alpha * sum_n mean_s log p(e | g) - KL, with freshly drawn paths."""
        pl = self._ensure(self._check_data(data))
        self._check_data(data)
        return float(pl.elbo(generate=True)[0])

    def training_loss_closure(self, data):
        pl = self._ensure(self._check_data(data))
        self._check_data(data)
        return lambda: float(-pl.elbo(generate=True)[0])

    def optimization_steps(self, data, num_steps: int, graph_unroll: int = 0) -> None:
        """num_steps x (ELBO, reverse pass, Adam) on the device in one vgpmp_elbo_steps call (plain launches measure
        2-3 % faster than hipGraph replay; graph_unroll > 0 captures and replays graphs of that many steps)."""
        pl = self._ensure(self._check_data(data))
        self._check_data(data)
        if graph_unroll and num_steps >= 2 * graph_unroll and pl._graph is None:
            pl.capture(graph_unroll)
            num_steps -= 1                      # capture ran one eager step
        pl.run_steps(num_steps)
        self.optimizer.iterations = pl.t
        self._sync_hyper()

    def _sync_hyper(self):
        pl = self._planner
        ell, var = pl.lengthscales()[0].cpu().numpy(), pl.variances()[0].cpu().numpy()
        for i, k in enumerate(self.kernel.kernels):
            k.lengthscales.assign(ell[i])
            k.variance.assign(var[i])
        if pl.lik_variables:
            self.alpha.assign(float(pl.alphas()[0]))
            self.likelihood.variance.assign(pl.sigma_obs()[0].cpu().numpy()[None])
        if pl.z_variables:
            iv = getattr(self.inducing_variable, "inducing_variable", self.inducing_variable)
            iv._Z.assign(pl.inducing_locations()[0].cpu().numpy())

    def sample_from_posterior(self, X, robot=None, compute_uncertainty=False):
        """models/vgpmp.py:312-331: (mean, best sample, first 7 samples, 2 sqrt(uncertainty)); uncertainty = the variance
        over the 150 samples of the end-effector position when compute_uncertainty, else 1."""
        X = np.asarray(X, dtype=np.float64)
        pl = self._ensure(self._n_train or X.shape[0])
        out = pl.sample_from_posterior(150, X, step=pl.t, compute_uncertainty=bool(compute_uncertainty))
        mu, best, samples = out[0], out[1], out[2]
        unc = 2.0 * np.sqrt(out[4][0].cpu().numpy().astype(np.float64)) if compute_uncertainty else 2.0
        return (mu[0].cpu().numpy().astype(np.float64), best[0].cpu().numpy().astype(np.float64),
                samples[0, :7].cpu().numpy().astype(np.float64), unc)

    def get_best_sample(self, samples):
        cost = self.likelihood.log_prob(samples).sum(-1)
        return int(torch.argmax(cost))

    def debug_likelihood(self, data):
        return float(self.likelihood.log_prob(data).mean(0).sum())


# ---------------------------------------------------------------- dispatchers ------------------------------
# Kuu / Kuf / Kfu / K_conditioned / prior_kl: vgpmp_amd/host/covariances.py (every number from the device)
from .covariances import K_conditioned, Kfu, Kuf, Kuu, prior_kl  # noqa: E402,F401
