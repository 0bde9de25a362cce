#!/bin/bash
# Measurement aid: interleaved bench.py runs over several variant libraries tools/libvgpmp_<name>.so ("product" = the built one).
#   tools/ab_multi.sh "product L0 L1" [bench.py arguments]
cd "$(dirname "$0")/.."
names=$1; shift
for i in 1 2 3; do
  for v in $names; do
    if [ $v = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$v.so; fi
    python bench.py --no-cpu-baseline --no-solve --min-seconds 1 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(1e3 * d['ms_per_step'], 2), 'us/step')"
  done
done
