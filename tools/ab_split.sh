#!/bin/bash
# Measurement aid: the fused prior kernel of variant libraries tools/libvgpmp_<name>.so ("product" = the built one) on the
# config-5 share and on 64 Franka problems: its own event time (roofline_secondary.avg_launch_ms) and the step.
#   tools/ab_split.sh "product v1 v2"
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in $1; do
    if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
    for w in "--workload stress" "--problems 64 --scene synthetic"; do
      python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10 --min-seconds 0.5 --allow-nan 2>/dev/null | tail -1 | V=$v W="$w" python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print(os.environ['V'], '|', os.environ['W'], '| prior kernel us', round(1e3 * d['roofline_secondary']['avg_launch_ms'], 1), '| step ms', round(d['ms_per_step'], 4))"
    done
  done
done
