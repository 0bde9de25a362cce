#!/usr/bin/env python3
"""Measurement aid (profiles/r06/flake.md, "What triggers it"): does the hazard reach code that is not ours?  PyTorch 2.10.0+rocm7.0's own
gfx950 kernels hold 338 packed-FP32 instructions of the form (all in complex<float> element-wise kernels: addcmul, addr, add, foreach
multiply, lerp -- a complex product IS "a pair times the second component of another pair").  This script runs some of those operators on
complex64 tensors over and over while vgpmp_debug_mfma_load (f16 matrix instructions and nothing else) runs on a second stream of the same
process, and compares every result, bit for bit, with the one taken on an idle device.

    python tools/torch_victim_probe.py [seconds per operator = 3]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgpmp_amd import capi      # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    lib = capi.load(require=True)
    torch.manual_seed(0)
    n = 1 << 22
    a = torch.randn(n, dtype=torch.complex64, device="cuda")
    b = torch.randn(n, dtype=torch.complex64, device="cuda")
    t = torch.randn(n, dtype=torch.complex64, device="cuda")
    s0 = torch.tensor(0.75 - 1.25j, dtype=torch.complex64, device="cuda")      # a 0-dim tensor: the "scalar tensor2" kernel of addcmul
    v1, v2 = torch.randn(2048, dtype=torch.complex64, device="cuda"), torch.randn(2048, dtype=torch.complex64, device="cuda")
    m = torch.randn(2048, 2048, dtype=torch.complex64, device="cuda")
    an, bn = a.view(2048, 2048).t(), b.view(2048, 2048).t()      # non-contiguous views
    ops = {
        "addcmul(t, a, 0-dim, value=2+3j)": lambda: torch.addcmul(t, a, s0, value=2 + 3j),
        "addcmul(t, a, b, value=2+3j)": lambda: torch.addcmul(t, a, b, value=2 + 3j),
        "a * b": lambda: a * b,
        "a * (0.75-1.25j)": lambda: a * (0.75 - 1.25j),
        "a.t() + b.t() (non-contiguous)": lambda: an + bn,
        "addr(m, v1, v2, beta=0.5+1j, alpha=2-1j)": lambda: torch.addr(m, v1, v2, beta=0.5 + 1j, alpha=2 - 1j),
        "lerp(a, b, 0.3+0.2j)": lambda: torch.lerp(a, b, 0.3 + 0.2j),
        "_foreach_mul([a, b], 1.5-0.5j)": lambda: torch.stack(torch._foreach_mul([a, b], 1.5 - 0.5j)),
        "float32 a.real * b.real + t.real (control)": lambda: a.real * b.real + t.real,
    }
    second = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    for name, op in ops.items():
        torch.cuda.synchronize()
        ref = op()
        torch.cuda.synchronize()
        end, reps, beside, bad, worst, lanes = None, 0, 0, 0, 0.0, set()
        t0 = time.time()
        while time.time() - t0 < seconds:
            if end is None or end.query():
                for _ in range(20):
                    capi.check(lib.vgpmp_debug_mfma_load(capi.ptr(sink), 1024, 20000, int(second.cuda_stream)), "vgpmp_debug_mfma_load")
                end = torch.cuda.Event(); end.record(second)
            out = op()
            torch.cuda.current_stream().synchronize()
            reps += 1
            beside += not end.query()
            ne = torch.view_as_real(out) != torch.view_as_real(ref) if out.is_complex() else (out != ref)
            if bool(ne.any()):
                bad += 1
                worst = max(worst, float((out - ref).abs().max()))
                idx = ne.reshape(ne.shape[0] if ne.dim() == 1 else -1, *([2] if out.is_complex() else [])).reshape(-1, 2 if out.is_complex() else 1).any(1).nonzero().flatten()[:4096]
                lanes.update((idx % 64).tolist())
        second.synchronize()
        print(f"{name:46s} {reps:6d} repetitions, {beside:6d} beside the matrix kernel, WRONG in {bad:6d}"
              + (f"; largest error {worst:.3g}; element index mod 64 of the wrong ones: {sorted(lanes)[:6]} .. {sorted(lanes)[-3:]}" if bad else ""), flush=True)


if __name__ == "__main__":
    main()
