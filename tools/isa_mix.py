"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py <file.s> <substring of the mangled name>"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = Counter(ins)


def group(k):
    if "dpp" in k:
        return "dpp"
    if k.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
        return "_".join(k.split("_")[:2])
    if k.startswith("s_"):
        return "s_branch" if "branch" in k else "s_wait/barrier" if ("waitcnt" in k or "barrier" in k) else "scalar"
    if "f64" in k:
        return "v_f64"
    if any(t in k for t in ("exp", "log", "rcp", "rsq", "sqrt", "sin_", "cos_")):
        return "v_trans"
    return "v_other"


g = Counter()
for k, v in c.items():
    g[group(k)] += v
print(lines[start][:90], "total", len(ins))
print(sorted(g.items(), key=lambda x: -x[1]))
print(c.most_common(45))
