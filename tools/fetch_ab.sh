#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/xcc_probe
for v in product xcdlik; do
  if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
  rm -rf gpurun_out/fa_$v
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fa_$v -- python3 bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 2 --min-seconds 0 > /dev/null 2> gpurun_out/fa_$v.err
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/fa_$v/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "loglik_paths_kernel" in r["Kernel_Name"]]
print("$v loglik FETCH_SIZE per launch (MB):", round(sum(v) / len(v) * 1024 / 1e6, 1), "launches", len(v))
PY
  rm -rf gpurun_out/fa_$v
done
