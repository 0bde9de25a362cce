#!/bin/bash
# Measurement aid: per-kernel average durations of the eager single-problem bench (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_fused; rm -rf $out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --scene synthetic --no-cpu-baseline --no-solve --unroll ${UNROLL:-0} --profile-steps 1 "$@" > gpurun_out/prof_fused.log 2>&1
echo rc=$?
f=$(ls $out/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us  {r['Percentage']}%")
PY
