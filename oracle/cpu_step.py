"""ctypes loader of oracle/cpu_step.cpp -- TEST INFRASTRUCTURE and CPU baseline (see that file's header); never imported by the
product.  `step()` takes the oracle's own objects (vgpmp_oracle.Scene / Params / AdamState / Noise) and advances them in place."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _host_tag() -> str:
    """-march=native code must run where it was built: one library per distinct set of CPU flags (this container and the GPU
    box need not share a CPU; the snapshot that travels carries the other machine's file, which is then simply not used)."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as fh:
            flags = next((ln for ln in fh if ln.startswith("flags")), "")
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


LIB = os.path.join(HERE, "_build", f"libvgpmp_cpu_step.{_host_tag()}.so")


class _Scene(C.Structure):
    _fields_ = [("D", C.c_int32), ("P", C.c_int32), ("craig", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("dh", C.c_void_p), ("twist", C.c_void_p), ("base", C.c_void_p), ("sphere_frame", C.c_void_p),
                ("sphere_off", C.c_void_p), ("radii", C.c_void_p), ("low", C.c_void_p), ("high", C.c_void_p),
                ("sigma_obs", C.c_void_p), ("grid", C.c_void_p), ("origin", C.c_double * 3), ("offset", C.c_double * 3),
                ("delta", C.c_double), ("epsilon", C.c_double)]


class _Dims(C.Structure):
    _fields_ = [("S", C.c_int32), ("N", C.c_int32), ("M", C.c_int32), ("B", C.c_int32)]


class _State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_mu", "q_sqrt", "raw_ell", "raw_var", "m_q_mu", "m_q_sqrt", "m_raw_ell", "m_raw_var",
                                          "v_q_mu", "v_q_sqrt", "v_raw_ell", "v_raw_var")]


class _Noise(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("omega", "beta", "w", "eps", "eps2")]


def build(force: bool = False) -> str:
    """Builds the library for this host's CPU if it is missing or older than its source.  Concurrent builders (pytest, bench.py's
    OpenMP worker, __graft_entry__.build()) each write their own temporary file and rename it into place; a failed build raises
    with the compiler's own message."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "cpu_step.cpp")):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        tmp = f"{LIB}.{os.getpid()}.tmp"
        res = subprocess.run(["make", "-C", HERE, "-B", f"OUT={tmp}"], capture_output=True, text=True)
        if res.returncode != 0 or not os.path.exists(tmp):
            raise RuntimeError(f"oracle/cpu_step.cpp did not build (make rc {res.returncode}):\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}")
        os.replace(tmp, LIB)
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.vgo_step.restype = C.c_int
        _lib.vgo_step.argtypes = [C.POINTER(_Scene), C.POINTER(_Dims), C.POINTER(_State), C.POINTER(_Noise), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(C.c_double), C.c_void_p]
        _lib.vgo_max_threads.restype = C.c_int
        _lib.vgo_draw_noise.restype = C.c_int
        _lib.vgo_draw_noise.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    return _lib


class NoiseBuffers:
    """Preallocated arrays of one step's randomness, refilled by the compiled generator (timed baseline only)."""

    def __init__(self, S, L, D, B, Mz):
        self.dims = (S, L, D, B, Mz)
        self.omega, self.beta = np.empty((L, B, D)), np.empty((L, B))
        self.w, self.eps, self.eps2 = np.empty((S, L, B)), np.empty((S, Mz, L)), np.empty((S, Mz, L))

    def draw(self, seed: int, threads: int = 0):
        S, L, D, B, Mz = self.dims
        load().vgo_draw_noise(int(seed), S, L, D, B, Mz, self.omega.ctypes.data, self.beta.ctypes.data, self.w.ctypes.data,
                              self.eps.ctypes.data, self.eps2.ctypes.data, int(threads))
        return self


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Problem:
    """One start-goal problem bound to the compiled step: keeps contiguous float64 copies of the scene and the variables."""

    def __init__(self, scene, X, Zy, y, p, st=None):
        rb, g = scene.robot, scene.sdf
        self.keep = dict(dh=_f64(rb.dh), twist=_f64(rb.twist), base=_f64(rb.base_pose), off=_f64(rb.sphere_offsets),
                         radii=_f64(rb.radii), low=_f64(rb.low), high=_f64(rb.high), grid=_f64(g.data),
                         sigma=_f64(np.broadcast_to(scene.sigma_obs, (rb.num_spheres,))),
                         frame=np.ascontiguousarray(rb.sphere_frame, dtype=np.int32), X=_f64(X), Zy=_f64(Zy), y=_f64(y))
        k = self.keep
        ptr = lambda a: a.ctypes.data
        nx, ny, nz = g.data.shape
        self.scene = _Scene(rb.dof, rb.num_spheres, int(bool(rb.craig)), nx, ny, nz, ptr(k["dh"]), ptr(k["twist"]), ptr(k["base"]),
                            ptr(k["frame"]), ptr(k["off"]), ptr(k["radii"]), ptr(k["low"]), ptr(k["high"]), ptr(k["sigma"]),
                            ptr(k["grid"]), (C.c_double * 3)(*np.asarray(g.origin, dtype=float)),
                            (C.c_double * 3)(*np.asarray(scene.offset, dtype=float)), float(g.delta), float(scene.epsilon))
        self.p = type(p)(*(_f64(a).copy() for a in (p.q_mu, p.q_sqrt, p.raw_ell, p.raw_var)))
        z = lambda a: np.zeros_like(a)
        src_m = st.m if st is not None else None
        src_v = st.v if st is not None else None
        names = ("q_mu", "q_sqrt", "raw_ell", "raw_var")
        self.m = [_f64(getattr(src_m, n)).copy() if src_m is not None else z(getattr(self.p, n)) for n in names]
        self.v = [_f64(getattr(src_v, n)).copy() if src_v is not None else z(getattr(self.p, n)) for n in names]
        self.t = int(st.t) if st is not None else 0
        arrs = [getattr(self.p, n) for n in names] + self.m + self.v
        self.state = _State(*(ptr(a) for a in arrs))
        self.M, self.L = self.p.q_mu.shape
        self.N, self.Mz = k["X"].shape[0], k["Zy"].shape[0]

    def step(self, noise, alpha, lr, trainable=0xF, do_adam=True, threads=0, want_grad=False):
        lib = load()
        nz = [_f64(a) for a in (noise.omega, noise.beta, noise.w, noise.eps, noise.eps2)]
        S, B = nz[2].shape[0], nz[0].shape[1]
        dims = _Dims(S, self.N, self.M, B)
        cn = _Noise(*(a.ctypes.data for a in nz))
        loss = C.c_double(0.0)
        grad = np.zeros(self.M * self.L + self.L * self.M * self.M + 2 * self.L) if want_grad else None
        k = self.keep
        rc = lib.vgo_step(C.byref(self.scene), C.byref(dims), C.byref(self.state), C.byref(cn), k["X"].ctypes.data, k["Zy"].ctypes.data,
                          k["y"].ctypes.data, float(alpha), float(lr), self.t, int(trainable), int(bool(do_adam)), int(threads),
                          C.byref(loss), grad.ctypes.data if want_grad else None)
        assert rc == 0
        if do_adam:
            self.t += 1
        if want_grad:
            M, L = self.M, self.L
            o = [0, M * L, M * L + L * M * M, M * L + L * M * M + L]
            return loss.value, (grad[:o[1]].reshape(M, L), grad[o[1]:o[2]].reshape(L, M, M), grad[o[2]:o[3]], grad[o[3]:])
        return loss.value
