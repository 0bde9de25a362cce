#!/bin/bash
# Measurement aid: per-kernel averages of one bench.py run under rocprofv3 (top 14 kernels).
#   tools/kstats.sh [bench.py arguments]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --no-cpu-baseline --no-solve --min-seconds 0.5 "$@" > /dev/null 2> gpurun_out/ks.err
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/ks/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("  %-70s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf gpurun_out/ks
