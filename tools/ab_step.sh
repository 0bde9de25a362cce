#!/bin/bash
# Measurement aid: A/B of the product library against tools/libvgpmp_ab.so (a variant build) on bench.py, interleaved.
#   tools/ab_step.sh [bench.py arguments]
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for v in product variant; do
    if [ $v = variant ]; then export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_ab.so; else unset VGPMP_HIP_LIB; fi
    python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-solve "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(1e3 * d['ms_per_step'], 2), 'us/step')"
  done
done
