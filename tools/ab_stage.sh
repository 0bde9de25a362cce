#!/bin/bash
# Measurement aid: one stage of the large-batch step (stage_ms key) under variant libraries tools/libvgpmp_<name>.so.
#   tools/ab_stage.sh paths_bwd "product v1 v2" [bench args; default: the config-5 share]
cd "$(dirname "$0")/.."
key=$1; names=$2; shift 2
args=${@:---workload stress}
for v in $names; do
  if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
  python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10 --min-seconds 0.5 --allow-nan 2>/dev/null | tail -1 | V=$v K=$key python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print(os.environ['V'], '|', os.environ['K'], 'us', round(1e3 * d['stage_ms'][os.environ['K']], 1), '| step ms', round(d['ms_per_step'], 4))"
done
