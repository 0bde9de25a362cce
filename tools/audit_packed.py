#!/usr/bin/env python3
"""Audit a shared library (or executable) with embedded gfx950 code objects for the packed-FP32 instruction form that misbehaves beside
a wide f16 / bf16 matrix instruction on MI355X / ROCm 7.0.2 (profiles/r06/flake.md, "The instruction", "What triggers it"): `v_pk_*`
with op_sel AND op_sel_hi both selecting the high register of source 1.  Prints the count of packed-FP32 instructions, the count of wide
matrix instructions (the library as a NEIGHBOUR) and the kernels that hold instructions of that form; exit code 1 if any is found.

    python tools/audit_packed.py [--kernels] [path ...]          (default: the product library)

Handles the plain clang offload bundles `hipcc` writes by default and the compressed ones ("CCOB", zstd: what PyTorch's libraries carry;
through clang-offload-bundler of the ROCm toolchain)."""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FORM = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\s+([^/]*)")
WIDE = re.compile(r"\bv_mfma_f32_(?:16x16x32|32x32x16)_(?:f16|bf16)\b")


def code_objects(path):
    """Yields the gfx950 code objects (bytes) embedded in `path`."""
    blob = open(path, "rb").read()
    plain = b"__CLANG_OFFLOAD_BUNDLE__"
    for m in re.finditer(re.escape(plain), blob):
        p = m.start()
        n = struct.unpack_from("<Q", blob, p + 24)[0]
        off = p + 32
        if n > 64:
            continue
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple and size:
                yield blob[p + o:p + o + size]
    for m in re.finditer(b"CCOB", blob):
        o = m.start()
        size = struct.unpack_from("<I", blob, o + 8)[0]
        if size < 32 or o + size > len(blob):
            continue
        with tempfile.TemporaryDirectory() as d:
            piece, out = os.path.join(d, "piece.bin"), os.path.join(d, "obj.co")
            open(piece, "wb").write(blob[o:o + size])
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + piece], capture_output=True, text=True)
            tg = [t for t in r.stdout.split() if "gfx950" in t]
            if r.returncode or not tg:
                continue
            if subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece, "--targets=" + tg[0], "--output=" + out],
                              capture_output=True).returncode:
                continue
            yield open(out, "rb").read()


def audit(path):
    packed, hits, wide, objects = collections.Counter(), collections.Counter(), 0, 0
    for co in code_objects(path):
        objects += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            text = subprocess.run([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout
        cur = "?"
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if WIDE.search(line):
                wide += 1
            mm = FORM.search(line)
            if not mm:
                continue
            packed[mm.group(1)] += 1
            sel = re.search(r"op_sel:\[([01,]+)\]", mm.group(2))
            sel_hi = re.search(r"op_sel_hi:\[([01,]+)\]", mm.group(2))
            lo = sel.group(1).split(",") if sel else ["0", "0", "0"]
            hi = sel_hi.group(1).split(",") if sel_hi else ["1", "1", "1"]
            if len(lo) > 1 and len(hi) > 1 and lo[1] == "1" and hi[1] == "1":
                hits[cur] += 1
    return objects, packed, hits, wide


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kernels = "--kernels" in sys.argv
    if not args:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from vgpmp_amd import build
        args = [str(build.LIB)]
    bad = 0
    for p in args:
        objects, packed, hits, wide = audit(p)
        if not objects:
            print(f"{p}: no gfx950 code object found")
            continue
        n = sum(hits.values())
        print(f"{p}: {objects} gfx950 code objects; packed-FP32 instructions {dict(packed) or 0}; of the form 'both results from source 1's high register': "
              f"{n} in {len(hits)} kernels; wide f16 / bf16 matrix instructions (the trigger): {wide}")
        if kernels:
            for k, v in hits.most_common(40):
                d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print("   %4d  %s" % (v, d[:200]))
        bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
