"""Measurement aid: loads, waits and vector instructions of kernels in a `hipcc -S` listing (full waits inside a gather loop
serialise it: lesson 60).   python tools/isa_waits.py <file.s> <substring of a mangled name> [...]"""
import sys

src = open(sys.argv[1]).read().split("\n")
for key in sys.argv[2:]:
    st = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    en = next(i for i in range(st + 1, len(src)) if src[i].startswith(".Lfunc_end"))
    ins = [l.strip() for l in src[st + 1:en] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = lambda pre: sum(1 for l in ins if l.startswith(pre))
    w0 = sum(1 for l in ins if l.startswith("s_waitcnt") and "vmcnt(0)" in l)
    wn = sum(1 for l in ins if l.startswith("s_waitcnt") and "vmcnt" in l)
    print(f"{key}: {len(ins)} instructions, {c('v_')} vector, {c('s_')} scalar, global_load_dwordx4 {c('global_load_dwordx4')}, "
          f"global_load_dword {c('global_load_dword ')}, ds_read {c('ds_read')}, ds_write {c('ds_write')}, s_waitcnt vmcnt(0) {w0} / any vmcnt {wn}, "
          f"scratch {c('scratch_')}")
