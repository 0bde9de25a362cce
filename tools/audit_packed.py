#!/usr/bin/env python3
"""Audit a shared library (or executable) with embedded gfx950 code objects for the packed-FP32 instruction form that misbehaves under
preemption on MI355X / ROCm 7.0.2 (profiles/r06/flake.md, "The instruction"): `v_pk_*` with op_sel AND op_sel_hi both selecting the high
register of source 1.  Prints the count of packed-FP32 instructions and every instruction of that form; exit code 1 if any is found.

    python tools/audit_packed.py [path ...]          (default: the product library)

Uncompressed clang offload bundles only (what `hipcc` writes by default); a library with compressed bundles reports "no gfx950 code
object found"."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgpmp_amd import build      # noqa: E402


def main():
    paths = sys.argv[1:] or [str(build.LIB)]
    bad = 0
    for p in paths:
        try:
            packed = build.packed_fp32_instructions(p)
            hits = build.src1_high_half_instructions(p)
        except RuntimeError as e:
            print(f"{p}: {e}")
            continue
        print(f"{p}: packed-FP32 instructions {packed or 0}; of the form 'both results from source 1's high register': {len(hits)}")
        for h in hits[:20]:
            print("   ", h)
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
