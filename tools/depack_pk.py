#!/usr/bin/env python3
"""Measurement aid (profiles/r06/flake.md): rewrite chosen packed-FP32 instructions of a gfx950 assembly file as two plain ones.

The hand-reduced reproducer (tools/sweep_probe.hip) fails under preemption when the compiler's SLP vectoriser has packed its arithmetic
and never when it has not; packed instructions written by hand (modes 4-8) did not reproduce it.  This tool bisects the FAILING build
itself: every `v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32` the selection names becomes the two VOP3 instructions that
compute the same two halves (same operands, same rounding), everything else -- schedule, registers, waits -- stays as the compiler
wrote it.

    hipcc -O3 --offload-arch=gfx950 -DMODE=1 -S --cuda-device-only -o m1.s tools/sweep_probe.hip
    tools/depack_pk.py m1.s out.s --ops fma,mul --where sgpr [--range 10:40] [--kernel _Z12sweep_kernelPK5RobotiPj]
    clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c out.s -o out.o && ld.lld -shared out.o -o out.hsaco
    SWEEP_HSACO=out.hsaco tools/sweep_probe_mod 6

--where (conditions joined by commas must all hold): all | sgpr (a scalar-register pair among the sources) | nosgpr | opsel (any op_sel / op_sel_hi written) | opsel_lo | opsel_hi | plain (none written)
         | neg (neg_lo / neg_hi written) | const (an inline constant among the sources)
--range a:b keeps only the a-th .. (b-1)-th of the selected instructions (in file order) -- for bisecting down to one instruction.
--invert   rewrites every packed instruction EXCEPT the selected ones.
"""
import argparse
import re
import sys

PK = re.compile(r"^(\s*)(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32|v_pk_mov_b32)\s+(.*?)\s*(;.*)?$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")
NSRC = {"v_pk_fma_f32": 3, "v_pk_mul_f32": 2, "v_pk_add_f32": 2, "v_pk_mov_b32": 2}
PLAIN = {"v_pk_fma_f32": "v_fma_f32", "v_pk_mul_f32": "v_mul_f32_e64", "v_pk_add_f32": "v_add_f32_e64"}


def half(op, sel):
    """Register (or constant) text of half `sel` of a 64-bit operand."""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if m:
        return "%s%d" % (m.group(1), int(m.group(2)) + sel)
    if re.fullmatch(r"-?\d+(\.\d+)?|0x[0-9a-f]+", op):      # inline constant: its low half is the value, its high half zero
        return op if sel == 0 else "0"
    raise ValueError("operand %r" % op)


def parse(rest):
    mods = {"op_sel": None, "op_sel_hi": None, "neg_lo": None, "neg_hi": None}
    for k, v in MOD.findall(rest):
        mods[k] = [int(x) for x in v.split(",")]
    ops = [o.strip() for o in MOD.sub("", rest).strip().rstrip(",").split(",")]
    return ops, mods


def depack(mn, rest, tmp):
    ops, mods = parse(rest)
    n = NSRC[mn]
    dst, src = ops[0], ops[1:1 + n]
    assert len(src) == n, (mn, rest)
    sel_lo = mods["op_sel"] or [0] * n
    sel_hi = mods["op_sel_hi"] or [1] * n
    neg_lo = mods["neg_lo"] or [0] * n
    neg_hi = mods["neg_hi"] or [0] * n
    dlo, dhi = half(dst, 0), half(dst, 1)
    if mn == "v_pk_mov_b32":      # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
        lo_src, hi_src = [half(src[0], sel_lo[0])], [half(src[1], sel_lo[1])]
        fmt = lambda d, s: "v_mov_b32_e32 %s, %s" % (d, s[0])
    else:
        sg = lambda s, neg: ("-" + s) if neg else s
        lo_src = [sg(half(src[i], sel_lo[i]), neg_lo[i]) for i in range(n)]
        hi_src = [sg(half(src[i], sel_hi[i]), neg_hi[i]) for i in range(n)]
        fmt = lambda d, s: "%s %s, %s" % (PLAIN[mn], d, ", ".join(s))
    reads = lambda srcs, reg: any(s.lstrip("-") == reg for s in srcs)
    if not reads(hi_src, dlo):
        return [fmt(dlo, lo_src), fmt(dhi, hi_src)]
    if not reads(lo_src, dhi):
        return [fmt(dhi, hi_src), fmt(dlo, lo_src)]
    return [fmt(tmp, lo_src), fmt(dhi, hi_src), "v_mov_b32_e32 %s, %s" % (dlo, tmp)]


def wanted(mn, rest, where):
    return all(wanted1(mn, rest, w) for w in where.split(","))      # "nosgpr,plain": every condition holds


def wanted1(mn, rest, where):
    ops, mods = parse(rest)
    src = ops[1:]
    if where == "all":
        return True
    if where == "sgpr":
        return any(s.startswith("s[") for s in src)
    if where == "nosgpr":
        return not any(s.startswith("s[") for s in src)
    if where == "opsel":
        return mods["op_sel"] is not None or mods["op_sel_hi"] is not None
    if where == "opsel_lo":
        return mods["op_sel"] is not None
    if where == "opsel_hi":
        return mods["op_sel_hi"] is not None
    if where == "plain":
        return all(v is None for v in mods.values())
    if where == "neg":
        return mods["neg_lo"] is not None or mods["neg_hi"] is not None
    if where == "const":
        return any(not s.startswith(("v[", "s[")) for s in src)
    raise SystemExit("unknown --where " + where)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--ops", default="fma,mul,add,mov")
    ap.add_argument("--where", default="all")
    ap.add_argument("--range", default=None)
    ap.add_argument("--invert", action="store_true")
    ap.add_argument("--kernel", default="_Z12sweep_kernelPK5RobotiPj")
    a = ap.parse_args()
    ops = {"v_pk_%s_%s" % (o, "b32" if o == "mov" else "f32") for o in a.ops.split(",") if o}
    lines = open(a.src).read().split("\n")
    # the kernel's body and its register budget
    start = next(i for i, l in enumerate(lines) if l.startswith(a.kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    nfree = next(i for i in range(end, len(lines)) if ".amdhsa_next_free_vgpr" in lines[i])
    nv = int(lines[nfree].split()[-1])
    tmp = "v%d" % nv
    sel = [i for i in range(start, end) if (m := PK.match(lines[i])) and m.group(2) in ops and wanted(m.group(2), m.group(3), a.where)]
    if a.range:
        lo, hi = (int(x) for x in a.range.split(":"))
        sel = sel[lo:hi]
    chosen = set(sel)
    if a.invert:
        chosen = {i for i in range(start, end) if PK.match(lines[i])} - chosen
    out, used_tmp, n = [], False, 0
    for i, l in enumerate(lines):
        if i in chosen:
            m = PK.match(l)
            new = depack(m.group(2), m.group(3), tmp)
            used_tmp |= len(new) == 3
            out += ["%s%s" % (m.group(1), x) for x in new]
            n += 1
        else:
            out.append(l)
    if used_tmp:      # one more register for the swaps (the accumulation offset stays a multiple of four)
        new_nv = (nv + 1 + 3) // 4 * 4
        text = "\n".join(out)
        k = text.index(a.kernel + ":")
        head, body = text[:k], text[k:]
        body = re.sub(r"(\.amdhsa_next_free_vgpr)\s+%d\b" % nv, r"\1 %d" % new_nv, body, count=1)
        body = re.sub(r"(\.amdhsa_accum_offset)\s+%d\b" % nv, r"\1 %d" % new_nv, body, count=1)
        body = re.sub(r"(\.vgpr_count:\s+)%d\b" % nv, r"\g<1>%d" % new_nv, body, count=1)
        out = (head + body).split("\n")
    open(a.dst, "w").write("\n".join(out))
    left = sum(1 for l in out if PK.match(l))
    print("%s: %d packed instructions rewritten (%s, %s%s%s), %d left%s" % (a.dst, n, a.ops, a.where, " range " + a.range if a.range else "",
          " inverted" if a.invert else "", left, "; one swap register added" if used_tmp else ""), file=sys.stderr)


if __name__ == "__main__":
    main()
