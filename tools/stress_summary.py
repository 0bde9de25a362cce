"""One line per table form from the files tools/collect_stress.sh leaves: python tools/stress_summary.py <dir>"""
import glob
import json
import os
import sys

d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "stress_*_bench.json"))):
    try:
        b = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e)
        continue
    r = b["roofline"]
    tag = os.path.basename(f)[7:-11]
    kb = json.load(open(f.replace("_bench.json", "_pmc_fetch_write_kb.json")))
    k = next((v for n, v in kb.items() if n.startswith(r["kernel"].split("<")[0]) and ("rows" in n) == ("rows" in r["kernel"])), {})
    q = b["roofline"]["algorithmic_bytes_per_launch"] / 1e6
    print(f"{tag:28s} {b['value']:8.0f} problem-steps/s  step {b['ms_per_step']:.3f} ms  {r['kernel']:44s} {1e3 * r['avg_launch_ms']:7.1f} us  "
          f"frac28 {r['frac']:.3f}  frac16 {r['by_16B_per_query']['frac']:.3f}  FETCH {k.get('fetch_size_kb', 0) / 1e6:.3f} GB  WRITE {k.get('write_size_kb', 0) / 1e6:.3f} GB")
    print("    ", {k2: round(v * 1e3) for k2, v in b["stage_ms"].items()})
