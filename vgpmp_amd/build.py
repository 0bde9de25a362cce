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
SOURCES = ["fk_sdf.hip", "gp_path.hip", "mesh_sdf.hip", "deriv_kernels.hip", "capi.hip"]
HEADERS = [CSRC / "vgpmp_device.h", CSRC / "gp_path.h", ROOT / "include" / "vgpmp.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc=0",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [CSRC / s for s in SOURCES] + HEADERS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    cmd = [hipcc(), *[f for f in FLAGS if f != "-fgpu-rdc=0"], f"-I{ROOT / 'include'}", f"-I{CSRC}",
           *[str(CSRC / s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
