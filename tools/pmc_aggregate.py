"""Per-kernel average of one PMC counter from a `rocprofv3 --pmc X --kernel-trace --output-format csv` run.

    python tools/pmc_aggregate.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> <traffic.json>

Writes the averages (KB, as the counters count) and the bytes-per-launch table `bench.py` reads.  Calibration
of the unit is read off sdf_pack_kernel in the same run (it reads 8 B and writes 16 B per voxel, streaming)."""
import csv
import glob
import json
import re
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
    fetch, n = averages(fdir, "FETCH_SIZE")
    write, _ = averages(wdir, "WRITE_SIZE")
    table = {k: {"launches": n[k], "fetch_size_kb": round(fetch[k], 1), "write_size_kb": round(write.get(k, 0.0), 1)}
             for k in sorted(fetch)}
    json.dump(table, open(out, "w"), indent=1)
    import os
    stamp = ""
    sp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "COLLECT_STAMP")
    if os.path.exists(sp):
        stamp = open(sp).read().strip()
    t = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, plain launches) of the same bench.py command, "
                   "aggregated per kernel by tools/pmc_aggregate.py; bytes = FETCH_SIZE*1024 x multiplier + WRITE_SIZE*1024, multiplier 2 "
                   "for kernels whose reads are wide coalesced streams (the gfx950 correction of MI355X_MICROARCH.md), 1 for the "
                   "likelihood's 16-byte gathers (FETCH_SIZE tallies 65 B per random 16-byte gather: profiles/r02/final/gather_probe_calib.txt)",
         "collected_at": stamp}
    wide16 = ("prior_gemm_kernel<0>", "prior_gemm_lds_kernel", "prior_gemm_tiled_kernel", "stage2_kernel<true, 0>", "stage2_kernel<true, 8>",
              "paths_bwd_sc8", "stage3_kernel")
    for k, v in table.items():
        mult = 2 if any(k.startswith(w.split("<")[0]) for w in wide16) else 1
        t[k] = dict(v, hbm_bytes_per_launch=int(mult * v["fetch_size_kb"] * 1024 + v["write_size_kb"] * 1024),
                    fetch_multiplier=mult)
    json.dump(t, open(traffic, "w"), indent=1)
    for k, v in t.items():
        if k not in ("source", "collected_at"):
            print(k, v)


if __name__ == "__main__":
    main()
