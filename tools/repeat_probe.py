"""Measurement aid: per-dispatch durations of paths_fwd launched 1+4 times back to back inside each step."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "paths_fwd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
by_pos = collections.defaultdict(list)
for i, d in enumerate(dur[50:]):
    by_pos[i % 5].append(d)
for k in sorted(by_pos): print("position", k, "n", len(by_pos[k]), "avg us", round(sum(by_pos[k]) / len(by_pos[k]), 2))
