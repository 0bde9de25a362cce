"""Builds libvgpmp_hip.so (gfx950) in-tree with hipcc.  `python -m vgpmp_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libvgpmp_hip.so"
SOURCES = ["fk_sdf.hip", "gp_path.hip", "mesh_sdf.hip", "deriv_kernels.hip", "plan.hip", "inducing.hip", "comm.hip", "capi.hip"]
HEADERS = [CSRC / "vgpmp_device.h", CSRC / "gp_path.h", CSRC / "fk_chain.h", CSRC / "gp_math.h", ROOT / "include" / "vgpmp.h", ROOT / "include" / "vgpmp_debug.h"]
# private parts of gp_path.hip (one translation unit: its stage launches dispatch these bodies by role)
GP_PARTS = [CSRC / n for n in ("gp_common.h", "gp_rng.h", "gp_paths.h", "gp_update.h", "gp_cov.h", "gp_prior.h", "gp_prior_split.h", "gp_lik_consts.h", "gp_wtable.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


OBJ_DIR = PKG / "build"


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(p.stat().st_mtime > t for p in deps)


def needs_build() -> bool:
    return _stale(LIB, [CSRC / s for s in SOURCES] + HEADERS + GP_PARTS)


def build(force: bool = False, verbose: bool = True, measurement: bool = False) -> Path:
    """One object per translation unit (rebuilt only when it or a header changed), then the link.

    `measurement=True` builds tools/libvgpmp_bisect.so instead: the same sources with -DVGPMP_BISECT (in-kernel time
    stamps and role switches for tools/step_trace.py, tools/role_probe.sh); never loaded unless VGPMP_HIP_LIB names it."""
    lib = ROOT / "tools" / "libvgpmp_bisect.so" if measurement else LIB
    obj_dir = OBJ_DIR.with_name("build_bisect") if measurement else OBJ_DIR
    if not measurement and not force and not needs_build():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir.mkdir(exist_ok=True)
    cflags = FLAGS + (["-DVGPMP_BISECT"] if measurement else [])
    objs, procs = [], []
    for src in SOURCES:
        obj = obj_dir / (Path(src).stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [CSRC / src] + HEADERS + (GP_PARTS if src == "gp_path.hip" else [])):
            cmd = [hipcc(), *cflags, f"-I{ROOT / 'include'}", f"-I{CSRC}", "-c", str(CSRC / src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *[str(o) for o in objs], "-ldl", "-o", str(lib)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measurement="--measurement" in sys.argv))
