"""Per-kernel average of one PMC counter from a `rocprofv3 --pmc X --kernel-trace --output-format csv` run.

    python tools/pmc_aggregate.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> <traffic.json>

Writes the averages (KB, as the counters count) and the bytes-per-launch table `bench.py` reads.  Calibration
of the unit is read off sdf_pack_kernel in the same run (it reads 8 B and writes 16 B per voxel, streaming)."""
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict


def averages(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = name.split("(")[0]
        acc[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return {k: acc[k] / cnt[k] for k in acc}, dict(cnt)


def main():
    fdir, wdir, out, traffic = sys.argv[1:5]
    tag = sys.argv[5] if len(sys.argv) > 5 else os.path.basename(out)       # workload: config2 / config3 / config5_mask / ...
    fetch, n = averages(fdir, "FETCH_SIZE")
    write, _ = averages(wdir, "WRITE_SIZE")
    table = {k: {"launches": n[k], "fetch_size_kb": round(fetch[k], 1), "write_size_kb": round(write.get(k, 0.0), 1)}
             for k in sorted(fetch)}
    json.dump(table, open(out, "w"), indent=1)
    stamp = ""
    sp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "COLLECT_STAMP")
    if os.path.exists(sp):
        stamp = open(sp).read().strip()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_bytes
    w = kernel_bytes.workload_of(tag)
    t = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, plain launches) of the same bench.py command, "
                   "aggregated per kernel by tools/pmc_aggregate.py.  hbm_bytes_per_launch = FETCH_SIZE*1024 x fetch_multiplier + "
                   "WRITE_SIZE*1024.  The multiplier is CALIBRATED PER KERNEL against the kernel's algorithmic read volume "
                   "(tools/kernel_bytes.py, every tensor once): gfx950 tallies a wide coalesced stream (16 B per lane, LDS-DMA rows) at "
                   "half its bytes (MI355X_MICROARCH.md, HBM) -- a kernel of that access pattern gets x2, and so does any kernel whose raw "
                   "FETCH_SIZE is below 0.75 of what it cannot avoid reading; scattered 16-byte gathers are tallied ~1:1 (65 B per "
                   "gather that misses: profiles/r02/final/gather_probe_calib.txt) and get x1; kernels without a stated volume get x1 "
                   "and are marked uncalibrated.  traffic_over_algorithmic = (reads x multiplier + writes) / (algorithmic reads + "
                   "writes): well above 1 is re-read or partial-sector traffic",
         "collected_at": stamp, "workload": tag}
    for k, v in table.items():
        rec = kernel_bytes.lookup(k, w)
        f_bytes, w_bytes = v["fetch_size_kb"] * 1024, v["write_size_kb"] * 1024
        if rec is None:
            mult, why = 1, "uncalibrated (no algorithmic volume stated for this kernel)"
            extra = {}
        else:
            raw = f_bytes / max(rec["read"], 1.0)
            if rec["access"] == "gather":
                mult, why = 1, "scattered 16-byte gathers"
            elif rec["access"] == "wide" or raw < 0.75:
                mult, why = 2, ("wide coalesced streams" if rec["access"] == "wide" else "raw FETCH_SIZE %.2f of the algorithmic reads: under-tallied" % raw)
            else:
                mult, why = 1, "raw FETCH_SIZE %.2f of the algorithmic reads" % raw
            extra = {"algorithmic_read_bytes": int(rec["read"]), "algorithmic_write_bytes": int(rec["write"]),
                     "fetch_over_algorithmic_reads_raw": round(raw, 3),
                     "write_over_algorithmic_writes": round(w_bytes / max(rec["write"], 1.0), 3),
                     "traffic_over_algorithmic": round((mult * f_bytes + w_bytes) / max(rec["read"] + rec["write"], 1.0), 3),
                     "algorithmic_flops": rec.get("flops"), "what": rec["note"]}
        t[k] = dict(v, hbm_bytes_per_launch=int(mult * f_bytes + w_bytes), fetch_multiplier=mult, multiplier_basis=why, **extra)
    json.dump(t, open(traffic, "w"), indent=1)
    for k, v in t.items():
        if k not in ("source", "collected_at", "workload"):
            print(k, v)


if __name__ == "__main__":
    main()
