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
HEADERS = [CSRC / "vgpmp_device.h", CSRC / "gp_path.h", CSRC / "fk_chain.h", CSRC / "gp_math.h", ROOT / "include" / "vgpmp.h", ROOT / "include" / "vgpmp_debug.h",
           Path(__file__).resolve()]      # (this file: a change of FLAGS rebuilds everything)
# private parts of gp_path.hip (one translation unit: its stage launches dispatch these bodies by role)
GP_PARTS = [CSRC / n for n in ("gp_common.h", "gp_rng.h", "gp_paths.h", "gp_update.h", "gp_cov.h", "gp_prior.h", "gp_prior_split.h", "gp_lik_consts.h", "gp_wtable.h")]
# -fno-slp-vectorize -fno-vectorize: NO packed-FP32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in the library.  On
# MI355X / ROCm 7.0.2 such an instruction, when its op_sel and op_sel_hi both select source 1's high register, reads 0.0 for that operand
# in lanes 48-63 while another wave of the same compute unit runs a wide f16 / bf16 matrix instruction (v_mfma_f32_16x16x32_f16 ...: this
# library's own prior draws, in another process or on another stream) -- tools/pk_probe.hip, tools/trigger_probe.py, profiles/r06/flake.md;
# the likelihood's wrong gradients of round 5.  The compiler's vectorisers are where all but three of the library's 8 687 packed
# instructions came from; they bought no time (config 2 50.2 / 50.2 us, config-5 share 735 / 732, 64 problems 362 / 355, config 3
# 121.6 / 122.0 us per step with / without).  packed_fp32_instructions() below is what tests/test_capi_load.py holds at zero;
# src1_high_half_instructions() lists the one form that misbehaves, for any library (tools/audit_packed.py): this one holds no packed
# instruction of any type.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-fno-vectorize"]


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


def _gfx950_disassemblies(lib: Path, objdump: str):
    """The disassembly (text) of every gfx950 code object embedded in `lib`: the clang offload bundles of its .hip_fatbin section."""
    import re
    import struct
    import tempfile
    blob = Path(lib).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    objects = 0
    for m in re.finditer(re.escape(magic), blob):
        p = m.start()
        n = struct.unpack_from("<Q", blob, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" not in triple or not size:
                continue
            objects += 1
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[p + o:p + o + size])
                f.flush()
                yield subprocess.run([objdump, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    if not objects:
        raise RuntimeError(f"no gfx950 code object found in {lib}")


def packed_fp32_instructions(lib: Path = LIB, objdump: str = "/opt/rocm/lib/llvm/bin/llvm-objdump") -> dict:
    """{instruction: count} of the packed-FP32 VALU instructions in the gfx950 code objects embedded in `lib`.  The library's policy is
    an empty dict (see FLAGS)."""
    import re
    found = {}
    for text in _gfx950_disassemblies(lib, objdump):
        for ins in re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", text):
            found[ins] = found.get(ins, 0) + 1
    return found


def src1_high_half_instructions(lib: Path = LIB, objdump: str = "/opt/rocm/lib/llvm/bin/llvm-objdump") -> list:
    """The packed (VOP3P, `v_pk_*`) instructions of `lib` whose op_sel AND op_sel_hi both select the high half of SOURCE 1 -- the one form
    that tools/pk_probe.hip shows reading 0.0 for that operand in lanes 48-63 beside a wide f16 matrix instruction (profiles/r06/flake.md, "The
    instruction", "What triggers it": v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[x,1,x] with op_sel_hi:[x,1,x], the default).  Policy: empty."""
    import re
    hits = []
    for text in _gfx950_disassemblies(lib, objdump):
        for line in text.split("\n"):
            m = re.search(r"\b(v_pk_\w+)\s+([^/]*)", line)
            if not m:
                continue
            sel = re.search(r"op_sel:\[([01,]+)\]", m.group(2))
            sel_hi = re.search(r"op_sel_hi:\[([01,]+)\]", m.group(2))
            lo = sel.group(1).split(",") if sel else ["0", "0", "0"]
            hi = sel_hi.group(1).split(",") if sel_hi else ["1", "1", "1"]
            if len(lo) > 1 and len(hi) > 1 and lo[1] == "1" and hi[1] == "1":
                hits.append((m.group(1) + " " + m.group(2)).strip())
    return hits


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measurement="--measurement" in sys.argv))
