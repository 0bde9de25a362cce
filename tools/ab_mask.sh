#!/bin/bash
# Measurement aid: the SDF pass of the config-5 share by free-space test (brick summary in memory, masks in LDS) and by
# library variant (tools/libvgpmp_<name>.so; "product" = the built one), interleaved on one box.
#   tools/ab_mask.sh "product nopipe" "on:off off:on off:off" [steps list, default "10 200"]      (forms are mask:summary)
cd "$(dirname "$0")/.."
libs=${1:-product}; forms=${2:-"on:off off:on"}; steps=${3:-"10 200"}
for rep in 1 2; do for st in $steps; do for v in $libs; do for f in $forms; do
  if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
  m=${f%%:*}; s=${f##*:}
  python bench.py --workload stress --mask $m --summary $s --steps $st --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10 --min-seconds 0.5 --allow-nan 2>/dev/null | tail -1 | V=$v M=$m S=$s ST=$st python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print(os.environ['V'], 'mask', os.environ['M'], 'summary', os.environ['S'], 'steps', os.environ['ST'], '| loglik us', round(1e3 * d['stage_ms']['loglik_fk_sdf'], 1), '| step ms', round(d['ms_per_step'], 4))"
done; done; done; done
