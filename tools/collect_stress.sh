#!/bin/bash
# BASELINE config 5 (one GPU's share: 14-DoF arm, 512^3 voxels = 2 GiB table, 64 problems) on the GPU box, from the repo
# root: bench line + rocprofv3 kernel statistics for each table form, FETCH_SIZE / WRITE_SIZE passes for the likelihood
# kernel, and the 16-byte-gather ceiling of the memory system (tools/gather_probe).  Output: gpurun_out/r02/stress_*.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${OUT:-gpurun_out/r02}; mkdir -p $out
B="--workload stress --problems 64 --grid 512 --steps 10 --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10 $EXTRA"
forms=${FORMS:-"linear:off brick:off brick:on"}
for f in $forms; do
  lay=${f%%:*}; sm=${f##*:}; tag=stress_${lay}_summary_${sm}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag.d -- python3 bench.py $B --layout $lay --summary $sm > $out/${tag}_bench_under_rocprof.json 2> $out/$tag.err
  cp $(ls $out/$tag.d/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
  rm -rf $out/$tag.d
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -- python3 bench.py $B --layout $lay --summary $sm --min-seconds 0 --profile-steps 2 > /dev/null 2> $out/${tag}_$c.err
  done
  python tools/pmc_aggregate.py $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE $out/${tag}_pmc_fetch_write_kb.json $out/${tag}_pmc_traffic.json > /dev/null
  rm -rf $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE
  timeout 600 python bench.py $B --layout $lay --summary $sm --traffic-file $out/${tag}_pmc_traffic.json > $out/${tag}_bench.json 2>> $out/$tag.err
  tail -c 1500 $out/${tag}_bench.json; echo
done
if [ -x tools/gather_probe ]; then
  timeout 300 tools/gather_probe > $out/gather_probe.txt 2>&1; cat $out/gather_probe.txt
  # what FETCH_SIZE tallies per uniformly random 16-byte gather from a 2 GiB table (known count per launch)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/gp_fetch -- tools/gather_probe calib > $out/gather_probe_calib.txt 2>&1
  python - <<PY >> $out/gather_probe_calib.txt
import csv, glob
f = glob.glob("$out/gp_fetch/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "gather" in r["Kernel_Name"]]
n = 256 * 8 * 8 * 64 * 64 * 4
print("FETCH_SIZE per gather launch (KB):", v, "-> bytes tallied per random 16-byte gather:", [x * 1024 / n for x in v])
PY
  rm -rf $out/gp_fetch; cat $out/gather_probe_calib.txt
fi
ls -la $out
