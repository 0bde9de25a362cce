#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in product s3; do
  if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
  rm -rf gpurun_out/ks_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$v -- python3 bench.py --no-cpu-baseline --no-solve --min-seconds 0.3 --profile-steps 2 > /dev/null 2> gpurun_out/ks_$v.err
  echo "== $v"
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/ks_$v/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("  %-60s calls %6s avg %7.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/ks_$v
done
