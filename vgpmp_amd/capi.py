"""ctypes binding of libvgpmp_hip.so (include/vgpmp.h).

PyTorch-ROCm tensors are only the memory container: every wrapper hands raw device pointers
and sizes to the C ABI on torch's current HIP stream.  There is NO fallback: if the library is
missing or cannot be loaded, importing the compute path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import numpy as np

MAX_DOF, MAX_SPHERES, MAX_MZ = 16, 64, 48
COMM_ID_BYTES = 128
TRAIN_Q_MU, TRAIN_Q_SQRT, TRAIN_LENGTHSCALES, TRAIN_KERNEL_VARIANCE = 1, 2, 4, 8
TRAIN_SIGMA_OBS, TRAIN_ALPHA = 16, 32      # need Problem.lik
TRAIN_INDUCING = 64                       # needs Problem.ind
# `what` of include/vgpmp.h: what a call computes
DO_FORWARD, DO_BACKWARD, DO_ADAM, GEN_NOISE, COV_ONLY = 1, 2, 4, 8, 2048
NOISE_AHEAD, NOISE_READY = 32768, 65536
# include/vgpmp_debug.h: measurement / test switches in the same argument (other schedules and kernel forms of the same numbers)
NO_FUSE, GEMM_DIRECT, NO_SPLIT, ELIM_BLOCK, LIK_LANES, LIK_LDS_STATE, COV_LDS_ROWS = 16, 32, 64, 128, 256, 512, 1024
NO_FUSE_PRIOR, PRIOR_F32, BWD_ONE_CHUNK = 4096, 8192, 16384

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libvgpmp_hip.so"

EXPORTS = ("vgpmp_version", "vgpmp_robot_upload", "vgpmp_sdf_table_bytes", "vgpmp_sdf_pack", "vgpmp_sdf_mask_words", "vgpmp_sdf_free_mask", "vgpmp_mesh_sdf", "vgpmp_fk_spheres", "vgpmp_sdf_query", "vgpmp_sdf_index_float",
           "vgpmp_log_prob", "vgpmp_cov_matrices", "vgpmp_kernel_derivative", "vgpmp_velocity_kuu_kuf", "vgpmp_workspace_bytes", "vgpmp_lik_scratch_bytes", "vgpmp_inducing_scratch_bytes", "vgpmp_generate_noise", "vgpmp_elbo_step",
           "vgpmp_elbo_steps", "vgpmp_elbo_step_profiled", "vgpmp_adam_step", "vgpmp_workspace_view", "vgpmp_sample_paths",
           "vgpmp_comm_unique_id", "vgpmp_comm_init", "vgpmp_allreduce_grads", "vgpmp_comm_destroy", "vgpmp_elbo_steps_reduced")
DEBUG_EXPORTS = ("vgpmp_debug_sphere_centres", "vgpmp_debug_last_schedule", "vgpmp_debug_mfma_load")      # include/vgpmp_debug.h
NUM_STAGES = 8
NUM_TIMES = 10
STAGE_NAMES = ("cov_fwd", "noise", "features", "prior_gemm", "paths_fwd", "loglik_fk_sdf", "paths_bwd", "final_adam")
KERNEL_TIME_NAMES = ("loglik_kernel", "prior_gemm_kernel")      # device start-to-end of those two kernels


class VgpmpError(RuntimeError):
    pass


class Robot(C.Structure):
    _fields_ = [("dof", C.c_int32), ("num_spheres", C.c_int32), ("craig", C.c_int32), ("reserved", C.c_int32),
                ("dh_d", C.c_float * MAX_DOF), ("dh_a", C.c_float * MAX_DOF),
                ("cos_alpha", C.c_float * MAX_DOF), ("sin_alpha", C.c_float * MAX_DOF),
                ("twist", C.c_float * MAX_DOF), ("low", C.c_float * MAX_DOF), ("high", C.c_float * MAX_DOF),
                ("base", C.c_float * 12), ("sphere_frame", C.c_int32 * MAX_SPHERES),
                ("sphere_off", (C.c_float * 3) * MAX_SPHERES), ("radius", C.c_float * MAX_SPHERES),
                ("sigma_obs", C.c_float * MAX_SPHERES), ("epsilon", C.c_float), ("reserved2", C.c_float),
                ("scene_offset", C.c_double * 3), ("inv_sigma_obs", C.c_float * MAX_SPHERES),
                ("joint_tab", (C.c_float * 8) * MAX_DOF), ("sphere_a", (C.c_float * 4) * MAX_SPHERES),
                ("sphere_b", (C.c_float * 2) * MAX_SPHERES),
                ("frame_first", C.c_int32 * 20)]


SDF_LINEAR, SDF_BRICK4 = 0, 1


class Sdf(C.Structure):
    _fields_ = [("table", C.c_void_p), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("layout", C.c_int32), ("origin", C.c_double * 3), ("delta", C.c_double),
                ("brick_min", C.c_void_p), ("free_mask", C.c_void_p), ("mask_shift", C.c_int32), ("mask_count", C.c_int32),
                ("mask_words", C.c_int32), ("reserved", C.c_int32), ("mask_clearance", C.c_float * 4)]


MAX_MASKS = 4


class Dims(C.Structure):
    _fields_ = [("num_problems", C.c_int32), ("S", C.c_int32), ("S_total", C.c_int32), ("N", C.c_int32),
                ("M", C.c_int32), ("L", C.c_int32), ("B", C.c_int32), ("split_k", C.c_int32),
                ("sample_offset", C.c_int32), ("reserved", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("q_mu", C.c_void_p), ("q_sqrt", C.c_void_p), ("raw_ell", C.c_void_p), ("raw_var", C.c_void_p)]


class Noise(C.Structure):
    _fields_ = [("omega", C.c_void_p), ("beta", C.c_void_p), ("w", C.c_void_p), ("eps", C.c_void_p),
                ("eps2", C.c_void_p)]


class LikParams(C.Structure):
    """vgpmp_lik_params: sigma_obs / alpha as trainable variables (device pointers)."""
    _fields_ = [("raw_alpha", C.c_void_p), ("raw_sigma", C.c_void_p), ("m_alpha", C.c_void_p), ("v_alpha", C.c_void_p),
                ("m_sigma", C.c_void_p), ("v_sigma", C.c_void_p), ("g_alpha", C.c_void_p), ("g_sigma", C.c_void_p),
                ("scratch", C.c_void_p)]


class InducingParams(C.Structure):
    """vgpmp_inducing_params: the inducing locations as trainable variables (device pointers)."""
    _fields_ = [("raw_Z", C.c_void_p), ("m_Z", C.c_void_p), ("v_Z", C.c_void_p), ("g_Z", C.c_void_p), ("Zy", C.c_void_p),
                ("scratch", C.c_void_p)]


class Problem(C.Structure):
    _fields_ = [("X", C.c_void_p), ("Zy", C.c_void_p), ("y_u", C.c_void_p), ("alpha", C.c_double),
                ("jitter", C.c_double), ("kl_scale", C.c_double), ("step_counter", C.c_void_p),
                ("lik", C.POINTER(LikParams)), ("ind", C.POINTER(InducingParams)), ("aux_stream", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("f", C.c_void_p), ("logp", C.c_void_p), ("lik", C.c_void_p), ("kl", C.c_void_p), ("grad", Params)]


_lib: Optional[C.CDLL] = None


def library_path() -> Path:
    return Path(os.environ.get("VGPMP_HIP_LIB", str(LIB_PATH)))


def load(require: bool = True) -> Optional[C.CDLL]:
    """Loads the shared library.  With require=True (the product path) a missing library is fatal."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        if require:
            raise VgpmpError(f"{path} not found: build it with `python -m vgpmp_amd.build` "
                             "(the HIP path has no CPU fallback)")
        return None
    # PyTorch bundles its own libamdhip64.so.7; it must be the process's HIP runtime BEFORE this
    # library is mapped, otherwise two runtimes coexist and torch's streams/pointers are foreign
    # to ours (hipErrorNoDevice on the first call).
    import torch  # noqa: F401
    lib = C.CDLL(str(path))
    lib.vgpmp_version.restype = C.c_char_p
    P = C.POINTER
    vp, i32, i64, u32, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double
    sigs = {
        "vgpmp_robot_upload": [P(Robot), vp, vp],
        "vgpmp_sdf_table_bytes": [i32, i32, i32, i32, P(C.c_size_t), P(C.c_size_t)],
        "vgpmp_sdf_pack": [P(Sdf), vp, i32, i32, i32, i32, vp],
        "vgpmp_sdf_mask_words": [i32, i32, i32, i32, P(C.c_size_t)],
        "vgpmp_sdf_free_mask": [P(Sdf), vp],
        "vgpmp_mesh_sdf": [vp, vp, i32, i32, i32, i32, P(C.c_double), dbl, vp, vp],
        "vgpmp_fk_spheres": [vp, vp, i64, vp, vp, vp],
        "vgpmp_sdf_query": [P(Sdf), vp, i64, vp, vp, vp, vp],
        "vgpmp_sdf_index_float": [P(Sdf), C.POINTER(C.c_double), vp, i64, vp, vp],
        "vgpmp_log_prob": [vp, i32, P(Sdf), vp, i64, vp, vp, vp],
        "vgpmp_cov_matrices": [i32, vp, i32, vp, i32, i32, vp, vp, dbl, vp, vp],
        "vgpmp_kernel_derivative": [i32, i32, vp, i32, vp, i32, dbl, dbl, vp, vp],
        "vgpmp_velocity_kuu_kuf": [i32, vp, vp, i32, i32, i32, vp, vp, dbl, vp, vp, vp],
        "vgpmp_workspace_bytes": [P(Dims), P(C.c_size_t)],
        "vgpmp_lik_scratch_bytes": [P(Dims), P(C.c_size_t)],
        "vgpmp_inducing_scratch_bytes": [P(Dims), P(C.c_size_t)],
        "vgpmp_generate_noise": [P(Dims), P(Noise), u32, u32, u32, vp],
        "vgpmp_elbo_step": [P(Dims), vp, P(Sdf), P(Problem), P(Params), P(Params), P(Params), P(Noise), P(Outputs),
                            vp, C.c_size_t, i32, i32, dbl, i32, u32, u32, u32, vp],
        "vgpmp_elbo_steps": [P(Dims), vp, P(Sdf), P(Problem), P(Params), P(Params), P(Params), P(Noise), P(Outputs),
                             vp, C.c_size_t, i32, i32, dbl, i32, u32, u32, u32, i32, vp],
        "vgpmp_elbo_step_profiled": [P(Dims), vp, P(Sdf), P(Problem), P(Params), P(Params), P(Params), P(Noise),
                                     P(Outputs), vp, C.c_size_t, i32, i32, dbl, i32, u32, u32, u32, vp,
                                     P(C.c_float)],
        "vgpmp_adam_step": [P(Dims), P(Params), P(Params), P(Params), P(Params), i32, dbl, i32, vp],
        "vgpmp_workspace_view": [P(Dims), vp, C.c_char_p, P(vp), P(C.c_size_t), P(i32)],
        "vgpmp_sample_paths": [P(Dims), vp, vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp],
        "vgpmp_comm_unique_id": [vp],
        "vgpmp_comm_init": [vp, i32, i32, P(vp)],
        "vgpmp_allreduce_grads": [vp, vp, C.c_size_t, vp],
        "vgpmp_comm_destroy": [vp],
        "vgpmp_elbo_steps_reduced": [P(Dims), vp, P(Sdf), P(Problem), P(Params), P(Params), P(Params), P(Noise), P(Outputs),
                                     vp, C.c_size_t, i32, i32, dbl, i32, u32, u32, u32, i32, vp, vp, C.c_size_t, vp],
    }
    sigs["vgpmp_debug_sphere_centres"] = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    sigs["vgpmp_debug_mfma_load"] = [vp, i32, i32, vp]
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.vgpmp_debug_last_schedule.argtypes = [C.c_char_p, C.c_size_t]
    lib.vgpmp_debug_last_schedule.restype = C.c_int64
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    names = {-1: "VGPMP_E_ARG", -2: "VGPMP_E_SHAPE", -3: "VGPMP_E_WORKSPACE", -4: "VGPMP_E_COMM"}
    if rc < 0:
        raise ValueError(f"{what}: {names.get(rc, rc)}")
    raise VgpmpError(f"{what}: hipError_t {rc}")


def make_robot(spec, sigma_obs, epsilon: float, scene_offset) -> Robot:
    """RobotSpec (+ likelihood constants) -> the POD of include/vgpmp.h."""
    D, P = spec.dof, spec.num_spheres
    if D > MAX_DOF or P > MAX_SPHERES:
        raise ValueError(f"robot {spec.name}: dof {D} / spheres {P} beyond {MAX_DOF}/{MAX_SPHERES}")
    r = Robot()
    r.dof, r.num_spheres, r.craig = D, P, int(bool(spec.craig))
    for i in range(D):
        r.dh_d[i], r.dh_a[i] = float(spec.dh[i, 0]), float(spec.dh[i, 1])
        r.cos_alpha[i], r.sin_alpha[i] = float(np.cos(spec.dh[i, 2])), float(np.sin(spec.dh[i, 2]))
        r.twist[i] = float(spec.twist[i])
        r.low[i], r.high[i] = float(spec.low[i]), float(spec.high[i])
    for i, v in enumerate(np.asarray(spec.base_pose)[:3, :].reshape(-1)):
        r.base[i] = float(v)
    frames = spec.sphere_frame
    sig = np.broadcast_to(np.asarray(sigma_obs, dtype=np.float64), (P,))
    for p in range(P):
        r.sphere_frame[p] = int(frames[p])
        for k in range(3):
            r.sphere_off[p][k] = float(spec.sphere_offsets[p, k])
        r.radius[p] = float(spec.sphere_radii[p])
        r.sigma_obs[p] = float(sig[p])
    r.epsilon = float(epsilon)
    for k in range(3):
        r.scene_offset[k] = float(scene_offset[k])
    return r


def stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def last_schedule(lib=None) -> list:
    """include/vgpmp_debug.h: the kernels the calling thread's last vgpmp_elbo_step* call enqueued for its last step."""
    lib = lib or load(require=True)
    need = int(lib.vgpmp_debug_last_schedule(None, 0))
    buf = C.create_string_buffer(need)
    lib.vgpmp_debug_last_schedule(buf, need)
    return [n for n in buf.value.decode().split("\n") if n]
