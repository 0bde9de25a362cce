"""Batched planning engine over the C ABI: device-resident scene (robot + voxel table) and a batch
of independent start-goal problems with their variational parameters, Adam state, noise buffers
and workspace.  Host logic only (allocation, argument packing, step counter); all arithmetic of
the ELBO step runs in libvgpmp_hip.so.

Reference roles: `DeviceScene` ~ the (Sampler, SignedDistanceField, likelihood constants) triple that
VGPMP.initialize wires together (models/vgpmp.py:154-159); `PlannerBatch` ~ one VGPMP model per
problem plus its tf.optimizers.Adam (models/vgpmp.py:71-82), `step()` ~ utils/miscellaneous.py:68-84.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
import weakref
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import capi
from .robots import RobotSpec

JITTER = 1e-6
VARIANCE_FLOOR = 0.1
ALPHA_FLOOR, SIGMA_FLOOR = 1e-4, 1e-5      # models/vgpmp.py:82, likelihoods/likelihood.py:31,41
Z_LOW, Z_HIGH = 0.09, 0.91                 # models/vgpmp.py:41 bounded_Z: tfb.Sigmoid(0.09, 0.91)
DEFAULT_TRAINABLE = dict(q_mu=True, q_sqrt=True, lengthscales=True, kernel_variance=True)


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise capi.VgpmpError("no HIP device visible: the vGPMP hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def trainable_mask(flags: Dict[str, bool]) -> int:
    m = 0
    if flags.get("q_mu", True):
        m |= capi.TRAIN_Q_MU
    if flags.get("q_sqrt", True):
        m |= capi.TRAIN_Q_SQRT
    if flags.get("lengthscales", True):
        m |= capi.TRAIN_LENGTHSCALES
    if flags.get("kernel_variance", True):
        m |= capi.TRAIN_KERNEL_VARIANCE
    if flags.get("sigma_obs", False):
        m |= capi.TRAIN_SIGMA_OBS
    if flags.get("alpha", False):
        m |= capi.TRAIN_ALPHA
    if flags.get("inducing_variable", False):
        m |= capi.TRAIN_INDUCING
    return m


def softplus_inverse(y):
    y = np.asarray(y, dtype=np.float64)
    return np.log(np.expm1(y))


class DeviceScene:
    """Robot tables + signed distance field resident in HBM."""

    def __init__(self, spec: RobotSpec, grid, scene_offset: Sequence[float], sigma_obs=0.005,
                 epsilon: float = 0.05, device: Optional[torch.device] = None, layout: str = "brick",
                 free_space_summary: Optional[bool] = None, slab_bytes: int = 256 << 20,
                 free_space_mask: Optional[bool] = None, mask_budget_bytes: int = 32 << 10):
        """grid = (data, origin, delta): data[x, y, z] as a NumPy array / torch tensor, or an object with `.shape` and
        `.rows(x_lo, x_hi, device) -> float64 device tensor` (rows produced on demand, e.g. scenes.AnalyticSceneRows).
        layout: "brick" (4x4x4 Morton bricks, include/vgpmp.h VGPMP_SDF_BRICK4) or "linear".
        free_space_summary: hand the per-brick minimum distance to the batch likelihood kernel (exact: spheres whose
        brick lies beyond epsilon + radius of every obstacle skip the table access); None = only for tables larger than
        the Infinity Cache, where that access is an HBM gather.
        free_space_mask: bit masks of the blocks of voxels that clear epsilon + radius (one mask per radius class, at most four,
        `mask_budget_bytes` in all) for the batch likelihood kernel, which keeps them in LDS: the same exact test without a
        memory request per query; None = as for the summary."""
        self.lib = capi.load(require=True)
        self.device = _require_gpu() if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.spec = spec
        data, origin, delta = grid
        self.shape = tuple(int(v) for v in data.shape)
        self.origin = np.asarray(origin, dtype=np.float64).copy()
        self.delta = float(delta)
        self.scene_offset = np.asarray(scene_offset, dtype=np.float64).copy()
        self.sigma_obs = np.broadcast_to(np.asarray(sigma_obs, dtype=np.float64), (spec.num_spheres,)).copy()
        self.epsilon = float(epsilon)
        self.host_robot = capi.make_robot(spec, self.sigma_obs, epsilon, self.scene_offset)
        self.dev_robot = torch.empty(C.sizeof(capi.Robot), dtype=torch.uint8, device=self.device)
        capi.check(self.lib.vgpmp_robot_upload(C.byref(self.host_robot), capi.ptr(self.dev_robot), self._stream()),
                   "vgpmp_robot_upload")
        self._upload_grid(data, layout, free_space_summary, slab_bytes)
        self._build_free_masks(free_space_mask, mask_budget_bytes)
        # forward-only sampler views shared by the planners of this scene (PlannerBatch.posterior_sampler): re-pointing one at
        # a planner, packing its arguments and launching on it happen under this lock (planners may be driven from threads)
        self._sampler_lock = threading.RLock()
        self._sampler_views: Dict = {}

    def _stream(self) -> int:
        """torch's current stream on THIS scene's device; launches need that device current (LDS attributes and
        kernel modules are per device), so a caller driving several GPUs from one process is switched over."""
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def _upload_grid(self, data, layout: str, free_space_summary: Optional[bool], slab_bytes: int) -> None:
        nx, ny, nz = self.shape
        self.layout = {"linear": capi.SDF_LINEAR, "brick": capi.SDF_BRICK4}[layout]
        tb, bb = C.c_size_t(0), C.c_size_t(0)
        capi.check(self.lib.vgpmp_sdf_table_bytes(nx, ny, nz, self.layout, C.byref(tb), C.byref(bb)), "vgpmp_sdf_table_bytes")
        self.table = torch.empty(tb.value // 4, dtype=torch.float32, device=self.device)
        self.brick_min = torch.empty(bb.value // 4, dtype=torch.float32, device=self.device) if bb.value else None
        self.sdf = capi.Sdf()
        self.sdf.table = capi.ptr(self.table)
        self.sdf.nx, self.sdf.ny, self.sdf.nz, self.sdf.layout = nx, ny, nz, self.layout
        for k in range(3):
            self.sdf.origin[k] = float(self.origin[k])
        self.sdf.delta = self.delta
        self.sdf.brick_min = capi.ptr(self.brick_min)           # the pack kernel fills it
        # float64 rows go up slab by slab (with one halo row each side): a 512^3 source never needs a full device copy
        step = max(4, (slab_bytes // (8 * ny * nz)) // 4 * 4)
        lazy = hasattr(data, "rows")
        if not lazy and not torch.is_tensor(data):
            data = torch.from_numpy(np.ascontiguousarray(np.asarray(data, dtype=np.float64)))
        for x0 in range(0, nx, step):
            x1 = min(nx, x0 + step)
            lo, hi = max(x0 - 1, 0), min(x1 + 1, nx)
            rows = data.rows(lo, hi, self.device) if lazy else data[lo:hi].to(self.device, torch.float64).contiguous()
            capi.check(self.lib.vgpmp_sdf_pack(C.byref(self.sdf), capi.ptr(rows), lo, hi, x0, x1, self._stream()),
                       "vgpmp_sdf_pack")
            torch.cuda.current_stream(self.device).synchronize()
            del rows
        if free_space_summary is None:
            free_space_summary = tb.value > (256 << 20)
        self.free_space_summary = bool(free_space_summary and self.brick_min is not None)
        if not self.free_space_summary:
            self.sdf.brick_min = None

    def _build_free_masks(self, want: Optional[bool], budget_bytes: int) -> None:
        """Free-space masks (include/vgpmp.h, vgpmp_sdf_free_mask): radius classes -> clearances epsilon + r (float32, one ulp
        up: the kernel re-checks `eps - (clearance - r) <= 0` in the hinge's own arithmetic, a sphere a class does not cover
        falls to the next), block edge = the smallest power of two (>= one brick) whose masks fit the LDS budget."""
        self.free_mask = None
        self.free_space_mask = False
        if want is None:
            want = self.table.numel() * 4 > (256 << 20)
        if not want or self.brick_min is None:
            return
        radii = np.unique(np.asarray(self.spec.sphere_radii, dtype=np.float64))
        if radii.size > capi.MAX_MASKS:      # the largest radius and evenly spaced quantiles below it
            radii = np.unique(np.quantile(radii, np.linspace(0.0, 1.0, capi.MAX_MASKS), method="higher"))
        clr = [float(np.nextafter(np.float32(np.float32(self.epsilon) + np.float32(r)), np.float32(np.inf))) for r in radii]
        nx, ny, nz = self.shape
        words = C.c_size_t(0)
        shift = 2
        while True:
            capi.check(self.lib.vgpmp_sdf_mask_words(nx, ny, nz, shift, C.byref(words)), "vgpmp_sdf_mask_words")
            if words.value * 4 * len(clr) <= budget_bytes or shift >= 12:
                break
            shift += 1
        self.free_mask = torch.zeros(words.value * len(clr), dtype=torch.int32, device=self.device)
        full = capi.Sdf.from_buffer_copy(self.sdf)
        full.brick_min = capi.ptr(self.brick_min)               # the masks derive from the summary whether or not the kernels read it
        full.free_mask, full.mask_shift, full.mask_count, full.mask_words = capi.ptr(self.free_mask), shift, len(clr), int(words.value)
        for k, c in enumerate(clr):
            full.mask_clearance[k] = c
        capi.check(self.lib.vgpmp_sdf_free_mask(C.byref(full), self._stream()), "vgpmp_sdf_free_mask")
        self.sdf.free_mask, self.sdf.mask_shift, self.sdf.mask_count, self.sdf.mask_words = full.free_mask, shift, len(clr), int(words.value)
        for k, c in enumerate(clr):
            self.sdf.mask_clearance[k] = c
        self.mask_shift, self.mask_clearances = shift, clr
        self.free_space_mask = True

    # ---- stand-alone pieces -------------------------------------------------------------------
    def fk_spheres(self, q: torch.Tensor, want_frames: bool = False):
        """Sampler.forward_kinematics_cost: q [n, dof] float32 -> sphere centres [n, P, 3]."""
        q = q.to(self.device, torch.float32).contiguous()
        n = q.shape[0]
        pos = torch.empty((n, self.spec.num_spheres, 3), dtype=torch.float32, device=self.device)
        frames = torch.empty((n, self.spec.dof + 1, 3, 4), dtype=torch.float32, device=self.device) if want_frames else None
        capi.check(self.lib.vgpmp_fk_spheres(capi.ptr(self.dev_robot), capi.ptr(q), n, capi.ptr(pos), capi.ptr(frames),
                                             self._stream()), "vgpmp_fk_spheres")
        return (pos, frames) if want_frames else pos

    def sdf_query(self, rel_pos: torch.Tensor):
        """get_distance_tf / get_distance_grad_tf on float64 positions relative to the scene."""
        rel = rel_pos.to(self.device, torch.float64).contiguous().reshape(-1, 3)
        n = rel.shape[0]
        idx = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        dist = torch.empty(n, dtype=torch.float32, device=self.device)
        grad = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        capi.check(self.lib.vgpmp_sdf_query(C.byref(self.sdf), capi.ptr(rel), n, capi.ptr(idx), capi.ptr(dist),
                                            capi.ptr(grad), self._stream()), "vgpmp_sdf_query")
        return idx, dist, grad

    def sdf_index_f32(self, pos: torch.Tensor) -> torch.Tensor:
        """Voxel indices [n, 3] of float32 sphere centres in the robot frame by the ELBO kernels' own index path (tests)."""
        pos = pos.to(self.device, torch.float32).contiguous().reshape(-1, 3)
        idx = torch.empty((pos.shape[0], 3), dtype=torch.int32, device=self.device)
        off = (C.c_double * 3)(*[float(v) for v in self.scene_offset])
        capi.check(self.lib.vgpmp_sdf_index_float(C.byref(self.sdf), off, capi.ptr(pos), pos.shape[0], capi.ptr(idx), self._stream()),
                   "vgpmp_sdf_index_float")
        return idx

    def log_prob(self, g: torch.Tensor, want_grad: bool = False):
        """VariationalMonteCarloLikelihood.log_prob on joint angles g [..., dof]."""
        shape = g.shape[:-1]
        g2 = g.to(self.device, torch.float32).contiguous().reshape(-1, self.spec.dof)
        n = g2.shape[0]
        logp = torch.empty(n, dtype=torch.float32, device=self.device)
        dl = torch.empty_like(g2) if want_grad else None
        capi.check(self.lib.vgpmp_log_prob(capi.ptr(self.dev_robot), self.spec.dof, C.byref(self.sdf), capi.ptr(g2), n,
                                           capi.ptr(logp), capi.ptr(dl), self._stream()), "vgpmp_log_prob")
        logp = logp.reshape(shape)
        return (logp, dl.reshape(g.shape)) if want_grad else logp

    def joint_sigmoid(self, f: torch.Tensor) -> torch.Tensor:
        low = torch.as_tensor(self.spec.low, dtype=f.dtype, device=f.device)
        high = torch.as_tensor(self.spec.high, dtype=f.dtype, device=f.device)
        return low + (high - low) * torch.sigmoid(f)

    def joint_sigmoid_inverse(self, g) -> np.ndarray:
        x = (np.asarray(g, dtype=np.float64) - self.spec.low) / (self.spec.high - self.spec.low)
        return np.log(x) - np.log1p(-x)


class PlannerBatch:
    """`num_problems` independent VGPMP models sharing one scene, optimised in lock step."""

    def __init__(self, scene: DeviceScene, queries: np.ndarray, *, num_samples: int, num_inducing: int,
                 num_data: int, lengthscales: Sequence[float], variance: float, alpha: float = 100.0,
                 learning_rate: float = 0.02, num_bases: int = 1024, trainable: Optional[Dict[str, bool]] = None,
                 seed: int = 0, problem_base: int = 0, split_k: Optional[int] = None,
                 samples_total: Optional[int] = None, sample_offset: int = 0, kl_scale: float = 1.0,
                 X: Optional[np.ndarray] = None):
        self.scene, self.lib, self.device = scene, scene.lib, scene.device
        spec = scene.spec
        q = np.asarray(queries, dtype=np.float64).reshape(-1, 2, spec.dof)
        P, L, M, S, N, B = q.shape[0], spec.dof, int(num_inducing), int(num_samples), int(num_data), int(num_bases)
        self.P, self.L, self.M, self.S, self.N, self.B, self.Mz = P, L, M, S, N, B, M + 2
        self.alpha, self.lr = float(alpha), float(learning_rate)
        self.trainable = dict(DEFAULT_TRAINABLE if trainable is None else trainable)
        self.seed, self.problem_base, self.t = int(seed), int(problem_base), 0
        if split_k is None:   # K-slices of the prior GEMM: few problems -> slices to fill the chip (fused stage launches)
            # large batches: the LDS-tiled GEMM (no slices) -- its 64-sample tiles need S >= 48 to pay off
            # K-slices pay while the launch is small: measured on config 2 shapes, 2 problems 88 vs 99 us per step with 4
            # slices, 3 problems 130 vs 123, 4 problems 152 vs 148, 5 up: one launch per kernel with the tiled GEMM
            # (the measure is the number of 64-sample row tiles, not of problems: ONE problem with 1024 samples -- config 4 on
            #  one rank, 96 row tiles -- takes 134 us per step without slices, 151 with four)
            row_tiles = -(-S // 64) * P * L
            split_k = 4 if (row_tiles <= 32 or S < 48) else 1
            while (B // split_k) % 16:
                split_k //= 2
        self.dims = capi.Dims(P, S, int(samples_total or S), N, M, L, B, int(split_k), int(sample_offset), 0)
        dev, f64, f32 = self.device, torch.float64, torch.float32
        # ---- unconstrained variables (models/vgpmp.py:166-171, 255-263)
        y_u = scene.joint_sigmoid_inverse(q)                                       # [P, 2, L]
        lin = np.stack([q[:, 0] + (q[:, 1] - q[:, 0]) * i / M for i in range(M)], axis=1)   # [P, M, L]
        q_mu = scene.joint_sigmoid_inverse(lin).transpose(0, 2, 1)                  # [P, L, M]
        var = max(float(variance), VARIANCE_FLOOR + 1e-6)    # 0.1 sits on the positive(lower=0.1) floor
        self.q_mu = torch.tensor(q_mu, dtype=f64, device=dev).contiguous()
        self.q_sqrt = torch.eye(M, dtype=f64, device=dev).repeat(P, L, 1, 1).contiguous()
        self.raw_ell = torch.tensor(np.tile(softplus_inverse(lengthscales), (P, 1)), dtype=f64, device=dev)
        self.raw_var = torch.full((P, L), float(softplus_inverse(var - VARIANCE_FLOOR)), dtype=f64, device=dev)
        self.y_u = torch.tensor(y_u, dtype=f64, device=dev).contiguous()
        Xn = np.tile(np.linspace(0.0, 1.0, N)[:, None], (1, L)) if X is None else np.asarray(X, dtype=np.float64)
        Zy = np.tile(np.concatenate([[0.0, 1.0], np.linspace(0.1, 0.9, M)])[:, None], (1, L))
        self.X = torch.tensor(Xn, dtype=f64, device=dev).contiguous()
        self.Zy = torch.tensor(Zy, dtype=f64, device=dev).contiguous()
        z = lambda t: torch.zeros_like(t)
        self.adam_m = [z(self.q_mu), z(self.q_sqrt), z(self.raw_ell), z(self.raw_var)]
        self.adam_v = [z(self.q_mu), z(self.q_sqrt), z(self.raw_ell), z(self.raw_var)]
        # gradient + ELBO pieces as ONE contiguous float64 buffer [q_mu | q_sqrt | raw_ell | raw_var | lik | kl]: with a
        # sharded sample axis the per-step exchange is a single in-place all-reduce of it (vgpmp_amd/sharding.py)
        shapes = [(P, L, M), (P, L, M, M), (P, L), (P, L), (P,), (P,)]
        self.reduce_buf = torch.zeros(sum(int(np.prod(sh)) for sh in shapes), dtype=f64, device=dev)
        parts, o = [], 0
        for sh in shapes:
            n = int(np.prod(sh))
            parts.append(self.reduce_buf[o:o + n].view(sh))
            o += n
        self.grad = parts[:4]
        # ---- noise, outputs, workspace
        self.omega = torch.empty((P, L, B, L), dtype=f32, device=dev)
        self.beta = torch.empty((P, L, B), dtype=f32, device=dev)
        self.w = torch.empty((P, S, L, B), dtype=f32, device=dev)
        self.eps = torch.empty((P, S, self.Mz, L), dtype=f32, device=dev)
        self.eps2 = torch.empty((P, S, self.Mz, L), dtype=f32, device=dev)
        self.f = torch.empty((P, S, L, N), dtype=f32, device=dev)
        self.logp = torch.empty((P, S, N), dtype=f32, device=dev)
        self.lik, self.kl = parts[4], parts[5]
        nbytes = C.c_size_t(0)
        capi.check(self.lib.vgpmp_workspace_bytes(C.byref(self.dims), C.byref(nbytes)), "vgpmp_workspace_bytes")
        self.workspace = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
        self.kl_scale = float(kl_scale)
        # ---- sigma_obs / alpha as variables (trainable_params.sigma_obs / alpha; reference default: constants)
        self.lik_variables = bool(self.trainable.get("sigma_obs", False) or self.trainable.get("alpha", False))
        if self.lik_variables:
            if int(samples_total or S) != S:
                raise NotImplementedError("trainable sigma_obs / alpha with a sharded sample axis")
            sig = np.full(capi.MAX_SPHERES, 1.0)
            sig[:spec.num_spheres] = scene.sigma_obs
            self.raw_alpha = torch.full((P,), float(softplus_inverse(self.alpha - ALPHA_FLOOR)), dtype=f64, device=dev)
            self.raw_sigma = torch.tensor(np.tile(softplus_inverse(sig - SIGMA_FLOOR), (P, 1)), dtype=f64, device=dev)
            self.lik_adam_m = [z(self.raw_alpha), z(self.raw_sigma)]
            self.lik_adam_v = [z(self.raw_alpha), z(self.raw_sigma)]
            self.lik_grad = [z(self.raw_alpha), z(self.raw_sigma)]
            nb = C.c_size_t(0)
            capi.check(self.lib.vgpmp_lik_scratch_bytes(C.byref(self.dims), C.byref(nb)), "vgpmp_lik_scratch_bytes")
            self.lik_scratch = torch.zeros(int(nb.value), dtype=torch.uint8, device=dev)
        # ---- inducing locations as variables (trainable_params.inducing_variable; reference default: constants)
        self.z_variables = bool(self.trainable.get("inducing_variable", False))
        if self.z_variables:
            if int(samples_total or S) != S:
                raise NotImplementedError("trainable inducing locations with a sharded sample axis")
            if B % 64:
                raise ValueError("trainable inducing locations need num_bases to be a multiple of 64")
            z01 = (np.linspace(0.1, 0.9, M) - Z_LOW) / (Z_HIGH - Z_LOW)
            raw = np.tile((np.log(z01) - np.log1p(-z01))[None, :, None], (P, 1, L))           # models/vgpmp.py:37-42
            self.raw_Z = torch.tensor(raw, dtype=f64, device=dev).contiguous()
            self.z_adam_m, self.z_adam_v, self.z_grad = z(self.raw_Z), z(self.raw_Z), z(self.raw_Z)
            self.Zy_all = torch.zeros((P, self.Mz, L), dtype=f64, device=dev)
            nb = C.c_size_t(0)
            capi.check(self.lib.vgpmp_inducing_scratch_bytes(C.byref(self.dims), C.byref(nb)), "vgpmp_inducing_scratch_bytes")
            self.z_scratch = torch.zeros(int(nb.value), dtype=torch.uint8, device=dev)
        # device-resident step counter: lets a captured hipGraph of the step be replayed
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.fuse = True          # False: one launch per kernel even for small batches (measurement)
        self.extra_flags = 0      # e.g. capi.NO_SPLIT (measurement)
        # the f16-split prior kernel of large batches keeps x, omega and x . omega as f16 pairs: its range argument assumes time
        # stamps of order one (init_trainset: [0, 1]).  Stamps far outside that go to the float32-MFMA form instead.
        self._f32_by_stamps = False
        self._guard_time_stamps(Xn)
        self._fused_by_what = {}      # `what` -> did the library run the few-problem schedule (asked of it once: _run)
        self._graph = None
        self._graph_unroll = 0
        # the step whose omega / beta / w a call with VGPMP_NOISE_AHEAD has left in the noise buffers (None: nobody's); every
        # other writer of those buffers clears it, so VGPMP_NOISE_READY can never pair another step's draws with this one's eps
        self.noise_ahead_step = None
        self._initial = [(t, t.clone()) for t in self._variables()]
        self._pack()

    def _guard_time_stamps(self, Xn) -> None:
        """Whenever the time stamps change: |X| > 16 sends the prior draws of large batches to the float32-MFMA form (the
        f16-split kernel's range argument assumes stamps of order one); back below the bound the guard's own flag goes again."""
        far = float(np.abs(np.asarray(Xn, dtype=np.float64)).max()) > 16.0
        if far and not (self.extra_flags & capi.PRIOR_F32):
            self.extra_flags |= capi.PRIOR_F32
            self._f32_by_stamps = True
        elif not far and self._f32_by_stamps:
            self.extra_flags &= ~capi.PRIOR_F32
            self._f32_by_stamps = False

    def set_time_stamps(self, Xn) -> None:
        """Replace X [N, L] (same N) on the device; re-evaluates the range guard of the f16-split prior kernel."""
        Xn = np.asarray(Xn, dtype=np.float64).reshape(tuple(self.X.shape))
        self._guard_time_stamps(Xn)
        self.X.copy_(torch.as_tensor(Xn, dtype=torch.float64))

    def _variables(self):
        """Every unconstrained variable this planner may train (the optional ones only when they exist)."""
        v = [self.q_mu, self.q_sqrt, self.raw_ell, self.raw_var]
        if self.lik_variables:
            v += [self.raw_alpha, self.raw_sigma]
        if self.z_variables:
            v += [self.raw_Z]
        return v

    def _moments(self):
        m = self.adam_m + self.adam_v
        if self.lik_variables:
            m += self.lik_adam_m + self.lik_adam_v + self.lik_grad
        if self.z_variables:
            m += [self.z_adam_m, self.z_adam_v, self.z_grad]
        return m

    def reset(self) -> None:
        """Back to the freshly initialised model of every problem (EVERY variable incl. sigma_obs / alpha / inducing locations
        when they are variables, their Adam moments, step count): what the reference gets by building a new VGPMP per start-goal
        query (utils/miscellaneous.py:162-169).  The effective likelihood constants and Zy are derived from the raw variables at
        the start of every call (lik_consts_kernel, z_build_kernel), so nothing cached survives.  Device copies only, no sync."""
        for dst, src in self._initial:
            dst.copy_(src)
        for t in self._moments():
            t.zero_()
        self.t = 0
        self.noise_ahead_step = None

    def _params_struct(self, tensors) -> capi.Params:
        return capi.Params(*(capi.ptr(t) for t in tensors))

    def _pack(self) -> None:
        self._params = self._params_struct([self.q_mu, self.q_sqrt, self.raw_ell, self.raw_var])
        self._am = self._params_struct(self.adam_m)
        self._av = self._params_struct(self.adam_v)
        self._noise = capi.Noise(capi.ptr(self.omega), capi.ptr(self.beta), capi.ptr(self.w), capi.ptr(self.eps),
                                 capi.ptr(self.eps2))
        lik = None
        if self.lik_variables:
            self._lik = capi.LikParams(capi.ptr(self.raw_alpha), capi.ptr(self.raw_sigma), capi.ptr(self.lik_adam_m[0]),
                                       capi.ptr(self.lik_adam_v[0]), capi.ptr(self.lik_adam_m[1]), capi.ptr(self.lik_adam_v[1]),
                                       capi.ptr(self.lik_grad[0]), capi.ptr(self.lik_grad[1]), capi.ptr(self.lik_scratch))
            lik = C.pointer(self._lik)
        ind = None
        if getattr(self, "z_variables", False):
            self._ind = capi.InducingParams(capi.ptr(self.raw_Z), capi.ptr(self.z_adam_m), capi.ptr(self.z_adam_v),
                                            capi.ptr(self.z_grad), capi.ptr(self.Zy_all), capi.ptr(self.z_scratch))
            ind = C.pointer(self._ind)
        aux = None      # (vgpmp_problem.aux_stream: reserved)
        self._problem = capi.Problem(capi.ptr(self.X), capi.ptr(self.Zy), capi.ptr(self.y_u), self.alpha, JITTER,
                                     self.kl_scale, None, lik, ind, aux)
        self._problem_ctr = capi.Problem(capi.ptr(self.X), capi.ptr(self.Zy), capi.ptr(self.y_u), self.alpha, JITTER,
                                         self.kl_scale, capi.ptr(self.step_counter), lik, ind, aux)
        self._out = capi.Outputs(capi.ptr(self.f), capi.ptr(self.logp), capi.ptr(self.lik), capi.ptr(self.kl),
                                 self._params_struct(self.grad))

    # ---- randomness -----------------------------------------------------------------------------
    def set_noise(self, omega, beta, w, eps, eps2) -> None:
        """Inject the random tensors of one ELBO evaluation (parity tests)."""
        for dst, src in ((self.omega, omega), (self.beta, beta), (self.w, w), (self.eps, eps), (self.eps2, eps2)):
            dst.copy_(torch.as_tensor(np.asarray(src), dtype=torch.float32).reshape(dst.shape))
        self.noise_ahead_step = None

    def generate_noise(self, step: int) -> None:
        self.noise_ahead_step = None
        capi.check(self.lib.vgpmp_generate_noise(C.byref(self.dims), C.byref(self._noise), self.seed, self.problem_base,
                                                 int(step), self.scene._stream()), "vgpmp_generate_noise")

    # ---- the ELBO step --------------------------------------------------------------------------
    def _run(self, what: int, step: int) -> None:
        what |= (0 if self.fuse else capi.NO_FUSE) | self.extra_flags
        if (what & capi.NOISE_READY) and self.noise_ahead_step != int(step):
            what &= ~capi.NOISE_READY          # the buffers hold another step's draws (or none): this call draws its own
        ahead = bool(what & capi.NOISE_AHEAD) and bool(what & capi.GEN_NOISE)
        self.noise_ahead_step = None
        capi.check(self.lib.vgpmp_elbo_step(
            C.byref(self.dims), capi.ptr(self.scene.dev_robot), C.byref(self.scene.sdf), C.byref(self._problem),
            C.byref(self._params), C.byref(self._am), C.byref(self._av), C.byref(self._noise), C.byref(self._out),
            capi.ptr(self.workspace), self.workspace.numel(), what, trainable_mask(self.trainable), self.lr,
            max(self.t, 1), self.seed, self.problem_base, int(step), self.scene._stream()), "vgpmp_elbo_step")
        if ahead:
            # only the few-problem schedule draws ahead (the large-batch one ignores both noise flags): ask the library what it
            # ran for this `what` -- once, the answer depends on nothing else -- instead of mirroring its rule here
            key = what & ~capi.NOISE_READY
            fused = self._fused_by_what.get(key)
            if fused is None:
                fused = self._fused_by_what[key] = any(n.startswith("stage1_kernel") for n in capi.last_schedule(self.lib))
            if fused:
                self.noise_ahead_step = int(step) + 1

    def sphere_centres(self) -> torch.Tensor:
        """include/vgpmp_debug.h, vgpmp_debug_sphere_centres: the float32 sphere centres [P, S, N, spheres, 3] the likelihood launch
        of the last evaluation formed from self.f (same kernel form: same batch rule, same flags).  For parity tests."""
        pos = torch.empty((self.P, self.S, self.N, self.scene.spec.num_spheres, 3), dtype=torch.float32, device=self.device)
        what = (0 if self.fuse else capi.NO_FUSE) | self.extra_flags
        capi.check(self.lib.vgpmp_debug_sphere_centres(capi.ptr(self.scene.dev_robot), capi.ptr(self.f), self.P, self.S, self.L,
                                                       self.N, what, capi.ptr(pos), self.scene._stream()), "vgpmp_debug_sphere_centres")
        return pos

    def elbo(self, generate: bool = True, step: Optional[int] = None) -> torch.Tensor:
        """VGPMP.elbo (models/vgpmp.py:265-289) for every problem: alpha * sum_n mean_s logp - KL."""
        self._run(capi.DO_FORWARD | (capi.GEN_NOISE if generate else 0), self.t if step is None else step)
        return self.lik - self.kl

    def accumulate_grad(self, generate: bool = True, step: Optional[int] = None) -> None:
        """Forward and reverse pass into self.lik / self.kl / self.grad; no update and nothing else on the stream (the
        sample-sharded step calls this every iteration: forming the loss tensor as well cost two launches per step)."""
        self._run(capi.DO_FORWARD | capi.DO_BACKWARD | (capi.GEN_NOISE if generate else 0),
                  self.t if step is None else step)

    def loss_and_grad(self, generate: bool = True, step: Optional[int] = None):
        """loss = -ELBO and its gradient wrt the unconstrained variables (no update)."""
        self.accumulate_grad(generate, step)
        return -(self.lik - self.kl), self.grad

    def step(self, generate: bool = True) -> None:
        """optimization_step (utils/miscellaneous.py:68-84): forward, reverse, Adam; no host sync."""
        step = self.t
        self.t += 1
        self._run(capi.DO_FORWARD | capi.DO_BACKWARD | capi.DO_ADAM | (capi.GEN_NOISE if generate else 0), step)

    # ---- hipGraph replay of the training step ----------------------------------------------------
    def _run_counter(self, num_steps: int = 1, stage_ms=None) -> None:
        """`num_steps` training steps whose noise key / Adam step count come from the device counter."""
        what = capi.DO_FORWARD | capi.DO_BACKWARD | capi.DO_ADAM | capi.GEN_NOISE | (0 if self.fuse else capi.NO_FUSE) | self.extra_flags
        what &= ~(capi.NOISE_READY | capi.NOISE_AHEAD)       # counter-driven calls always draw their own noise
        self.noise_ahead_step = None
        args = (C.byref(self.dims), capi.ptr(self.scene.dev_robot), C.byref(self.scene.sdf),
                C.byref(self._problem_ctr), C.byref(self._params), C.byref(self._am), C.byref(self._av),
                C.byref(self._noise), C.byref(self._out), capi.ptr(self.workspace), self.workspace.numel(), what,
                trainable_mask(self.trainable), self.lr, 0, self.seed, self.problem_base, 0)
        if stage_ms is None:
            capi.check(self.lib.vgpmp_elbo_steps(*args, int(num_steps), self.scene._stream()), "vgpmp_elbo_steps")
        else:
            capi.check(self.lib.vgpmp_elbo_step_profiled(*args, self.scene._stream(), stage_ms), "vgpmp_elbo_step_profiled")

    def capture(self, unroll: int = 10) -> None:
        """Capture `unroll` consecutive training steps into one hipGraph (torch.cuda.CUDAGraph is the
        capture plumbing; every node is one of our kernels)."""
        self.step_counter.fill_(self.t)
        self._run_counter()                       # warm-up outside capture (module load, LDS attributes)
        self.t += 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run_counter(unroll)
        self._graph, self._graph_unroll = g, int(unroll)

    def run_steps(self, steps: int) -> None:
        """`steps` optimisation steps: whole graphs while they fit, eager launches for the rest."""
        if self._graph is not None:
            self.step_counter.fill_(self.t)          # step() / adam_only() between capture and replay advance only self.t
            while steps >= self._graph_unroll:
                self._graph.replay()
                self.t += self._graph_unroll
                steps -= self._graph_unroll
        if steps > 0:
            self.step_counter.fill_(self.t)
            self._run_counter(steps)
            self.t += steps

    def profile_steps(self, steps: int):
        """`steps` training steps with a HIP event around every kernel: mean milliseconds per stage."""
        ms = (C.c_float * capi.NUM_TIMES)()
        self.step_counter.fill_(self.t)
        for _ in range(steps):
            self._run_counter(stage_ms=ms)
            self.t += 1
        return {name: ms[i] / steps for i, name in enumerate(capi.STAGE_NAMES + capi.KERNEL_TIME_NAMES)}

    def adam_only(self) -> None:
        """Adam.apply_gradients on self.grad (after an external all-reduce)."""
        self.t += 1
        capi.check(self.lib.vgpmp_adam_step(C.byref(self.dims), C.byref(self._params), C.byref(self._out.grad),
                                            C.byref(self._am), C.byref(self._av), trainable_mask(self.trainable),
                                            self.lr, self.t, self.scene._stream()), "vgpmp_adam_step")

    def view(self, name: str) -> torch.Tensor:
        """Intermediate of the last evaluation, copied out of the workspace (tests only)."""
        p, n, dbl = C.c_void_p(), C.c_size_t(), C.c_int32()
        capi.check(self.lib.vgpmp_workspace_view(C.byref(self.dims), capi.ptr(self.workspace), name.encode(),
                                                 C.byref(p), C.byref(n), C.byref(dbl)), "vgpmp_workspace_view")
        off = p.value - self.workspace.data_ptr()
        nbytes = n.value * (8 if dbl.value else 4)
        raw = self.workspace[off:off + nbytes]
        return raw.view(torch.float64 if dbl.value else torch.float32).clone()

    # ---- plan extraction (models/vgpmp.py:312-339) -------------------------------------------------
    def posterior_sampler(self, num_samples: int = 150, Xnew: Optional[np.ndarray] = None) -> "PlannerBatch":
        """A forward-only view of the same models at other (S, N): shares the parameter tensors.  The view is cached per
        (S, N); its time stamps are those of THIS call (the reference passes X through on every call, models/vgpmp.py:312-331)."""
        n_new = self.N if Xnew is None else int(np.asarray(Xnew).shape[0])
        key = (int(num_samples), n_new)
        cache = self.__dict__.setdefault("_samplers", {})
        if key not in cache:
            # the view's own buffers (noise, paths, workspace) depend on shapes and constants only: planners of the same scene
            # and shape -- one per start-goal query in the reference's driver loop -- hand one set on (the variables it reads
            # are re-pointed at THIS planner's below)
            shared = self.scene._sampler_views
            skey = key + (self.P, self.M, self.B, self.alpha, self.lr, self.seed, self.problem_base,
                          tuple(sorted(self.trainable.items())))
            child = shared.get(skey)
            if child is None:
                mid = np.tile(0.5 * (self.scene.spec.low + self.scene.spec.high), (self.P, 2, 1))   # placeholder queries
                child = PlannerBatch(self.scene, mid, num_samples=num_samples, num_inducing=self.M, num_data=n_new,
                                     lengthscales=[1.0] * self.L, variance=1.0, alpha=self.alpha, learning_rate=self.lr,
                                     num_bases=self.B, trainable=self.trainable, seed=self.seed + 7919,
                                     problem_base=self.problem_base, X=Xnew)
                while len(shared) >= 8:                      # bounded: the oldest view (and its workspace) goes
                    shared.pop(next(iter(shared)))
                shared[skey] = child
            cache[key] = child
        child = cache[key]
        owner = getattr(child, "_view_of", None)
        if owner is None or owner() is not self:              # (another planner of this scene and shape used the view since)
            child.q_mu, child.q_sqrt, child.raw_ell, child.raw_var = self.q_mu, self.q_sqrt, self.raw_ell, self.raw_var
            child.y_u = self.y_u
            if self.lik_variables:      # the trained sigma_obs / alpha weigh the samples of get_best_sample
                child.raw_alpha, child.raw_sigma = self.raw_alpha, self.raw_sigma
            if self.z_variables:        # the trained inducing locations
                child.raw_Z = self.raw_Z
            child._pack()
            child._view_of = weakref.ref(self)                # (no strong reference: the view must not keep a planner alive)
        Xn = np.tile(np.linspace(0.0, 1.0, n_new)[:, None], (1, self.L)) if Xnew is None else np.asarray(Xnew, dtype=np.float64)
        child.set_time_stamps(Xn)
        return child

    def extract_plans(self, want_samples: bool = True, compute_uncertainty: bool = False):
        """vgpmp_sample_paths on the state the last forward pass left: (mean [P,N,L], best path [P,N,L], samples [P,S,N,L] or
        None, best index [P], end-effector variance [P,N,3] or None), all on the device, joint angles in float32."""
        dev, f32 = self.device, torch.float32
        mean = torch.empty((self.P, self.N, self.L), dtype=f32, device=dev)
        best_path = torch.empty_like(mean)
        best = torch.empty(self.P, dtype=torch.int32, device=dev)
        samples = torch.empty((self.P, self.S, self.N, self.L), dtype=f32, device=dev) if want_samples else None
        ee = torch.empty((self.P, self.N, 3), dtype=f32, device=dev) if compute_uncertainty else None
        capi.check(self.lib.vgpmp_sample_paths(C.byref(self.dims), capi.ptr(self.scene.dev_robot), capi.ptr(self.workspace),
                                               self.workspace.numel(), capi.ptr(self.f), capi.ptr(self.logp), capi.ptr(mean),
                                               capi.ptr(best), capi.ptr(best_path), capi.ptr(samples), capi.ptr(ee),
                                               self.scene._stream()), "vgpmp_sample_paths")
        return mean, best_path, samples, best, ee

    def posterior_mean(self) -> torch.Tensor:
        """Mean of q(f) at this batch's X after a forward pass: joint_sigmoid(Kfu (Kuu + jI)^-1 q_mu), [P, N, L]."""
        return self.extract_plans(want_samples=False)[0]

    def sample_from_posterior(self, num_samples: int = 150, Xnew: Optional[np.ndarray] = None, step: int = 0,
                              compute_uncertainty: bool = False):
        """(mean, best sample, samples, best index[, end-effector variance]) per problem; best = argmax_s sum_n log p
        (models/vgpmp.py:312-339).  One forward-only step at Xnew, then vgpmp_sample_paths: no torch arithmetic."""
        with self.scene._sampler_lock:       # the view is shared by the planners of this scene: re-point, launch, extract as one
            sp = self.posterior_sampler(num_samples, Xnew)
            sp.elbo(generate=True, step=step)
            mean, best_path, samples, best, ee = sp.extract_plans(True, compute_uncertainty)
        return (mean, best_path, samples, best, ee) if compute_uncertainty else (mean, best_path, samples, best)

    def path_clearance(self, path: torch.Tensor) -> torch.Tensor:
        """Signed clearance (SDF distance minus sphere radius) of every sphere along joint paths
        [..., N, L] -> [..., N, P].  Used by the headless success check that replaces the reference's
        simulated trajectory execution (utils/robot.py:416-480)."""
        sc = self.scene
        q = path.reshape(-1, self.L).to(torch.float32)
        pos = sc.fk_spheres(q).to(torch.float64)
        rel = pos - torch.as_tensor(sc.scene_offset, dtype=torch.float64, device=pos.device)
        _, dist, _ = sc.sdf_query(rel.reshape(-1, 3))
        radii = torch.as_tensor(sc.spec.sphere_radii, dtype=torch.float32, device=pos.device)
        return (dist.reshape(-1, sc.spec.num_spheres) - radii).reshape(path.shape[:-1] + (sc.spec.num_spheres,))

    def query_clearances(self, num_points: int = 100):
        """Signed clearance (minimum over spheres) of what a query starts from: its start state, its goal state and the
        straight line between them in joint space sampled at `num_points` (the path the variational mean is initialised on,
        models/vgpmp.py:166-171): three [P] tensors.  A query whose own end states touch the obstacles cannot pass the
        headless success check whatever the planner does; the report next to plans/sec says so per query."""
        dev = self.device
        q = self.scene.joint_sigmoid(self.y_u)                                   # [P, 2, L] the pinned states
        lam = torch.linspace(0.0, 1.0, int(num_points), device=dev, dtype=torch.float64)[None, :, None]
        line = q[:, :1] + (q[:, 1:] - q[:, :1]) * lam                            # [P, n, L]
        c = self.path_clearance(line.to(torch.float32)).amin(dim=2)              # [P, n]
        return c[:, 0], c[:, -1], c.amin(dim=1)

    # ---- results --------------------------------------------------------------------------------
    def samples(self) -> torch.Tensor:
        """Joint-space paths of the last evaluation: joint_sigmoid(f) as [P, S, N, L]."""
        return self.scene.joint_sigmoid(self.f.permute(0, 1, 3, 2))

    def lengthscales(self) -> torch.Tensor:
        return torch.nn.functional.softplus(self.raw_ell)

    def variances(self) -> torch.Tensor:
        return VARIANCE_FLOOR + torch.nn.functional.softplus(self.raw_var)

    def inducing_locations(self) -> torch.Tensor:
        """Z of every problem [P, M, L] (the constants linspace(0.1, 0.9, M) unless trainable_params.inducing_variable)."""
        if not self.z_variables:
            return self.Zy[2:].unsqueeze(0).repeat(self.P, 1, 1)
        return Z_LOW + (Z_HIGH - Z_LOW) * torch.sigmoid(self.raw_Z)

    def alphas(self) -> torch.Tensor:
        """alpha of every problem [P] (the constant unless trainable_params.alpha / sigma_obs made it a variable)."""
        if not self.lik_variables:
            return torch.full((self.P,), self.alpha, dtype=torch.float64, device=self.device)
        return ALPHA_FLOOR + torch.nn.functional.softplus(self.raw_alpha)

    def sigma_obs(self) -> torch.Tensor:
        """likelihood.variance of every problem [P, num_spheres]."""
        n = self.scene.spec.num_spheres
        if not self.lik_variables:
            return torch.as_tensor(self.scene.sigma_obs, dtype=torch.float64, device=self.device).repeat(self.P, 1)
        return SIGMA_FLOOR + torch.nn.functional.softplus(self.raw_sigma[:, :n])
