"""Measurement aid (profiles/r06/flake.md): in a kernel's ISA (hipcc -S), find LDS / scalar-memory instructions whose ADDRESS registers are
overwritten while the instruction may still be in flight -- between its issue and the first `s_waitcnt lgkmcnt(n)` that covers it.  Legal
(the hardware reads addresses at issue), but if an in-flight instruction were ever re-issued from the register file (a replay after a
context restore), it would fetch from the wrong place.  Linear scan per kernel, labels ignored (approximate at loop back-edges).

    python tools/isa_addr_reuse.py file.s [kernel-name-substring ...]
"""
import re
import sys


def regs(tok):
    """'v12' -> {('v', 12)}; 'v[12:15]' -> four; 's[8:9]'; others -> empty."""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    if tok == "vcc":
        return {("s", 106), ("s", 107)}
    return set()


def scan(name, lines):
    inflight = []      # (kind, text, addr regs, line no)
    hits = []
    for no, ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                lds = [x for x in inflight if x[0] == "lds"]
                keep = lds[len(lds) - n:] if n else []
                inflight = keep + ([x for x in inflight if x[0] == "smem"] if n else [])
            continue
        # writes of this instruction
        written = set()
        if op.startswith(("ds_read", "s_load", "s_buffer_load")):
            written = regs(ops[0]) if ops else set()
        elif op.startswith(("ds_write", "global_store", "buffer_store", "s_cbranch", "s_branch", "s_endpgm", "s_nop", "s_barrier", "global_load", "buffer_load")):
            written = regs(ops[0]) if op.startswith(("global_load", "buffer_load")) and ops else set()
        elif ops:
            written = regs(ops[0])
            if op.startswith("v_cmp") or op.startswith("v_cmpx"):
                written = regs(ops[0])
        for x in inflight:
            if written & x[2]:
                hits.append((x[3], x[1], no, ln))
        if op.startswith("ds_read") or op.startswith("ds_write") or op.startswith("ds_bpermute"):
            addr = regs(ops[1]) if op.startswith("ds_read") and len(ops) > 1 else (regs(ops[0]) if ops else set())
            inflight.append(("lds", ln, addr, no))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memrealtime") or op.startswith("s_memtime"):
            addr = regs(ops[1]) if len(ops) > 1 else set()
            inflight.append(("smem", ln, addr, no))
        if op == "s_endpgm":
            inflight = []
    return hits


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2:]
    starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    for k, (i, nm) in enumerate(starts):
        if want and not any(w in nm for w in want):
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(text)
        hits = scan(nm, [(j + 1, text[j]) for j in range(i, end)])
        print(f"{nm[:110]}: {len(hits)} address registers overwritten under an in-flight LDS / scalar load")
        for h in hits[:6]:
            print(f"     line {h[0]}: {h[1]}   <- overwritten at line {h[2]}: {h[3]}")


if __name__ == "__main__":
    main()
