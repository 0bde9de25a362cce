#!/bin/bash
# Measurement aid: interleaved product-build runs of the working tree's library against tools/libvgpmp_<name>.so (tools/build_variant.sh).
#   bash tools/ab_var.sh <name> [reps]
name=$1; reps=${2:-2}
for rep in $(seq $reps); do
for lib in base $name; do
  if [ $lib = base ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_$name.so; fi
  for wl in "--workload config3 --steps 130" "--workload stress --steps 200" "--problems 64 --steps 200"; do
    python bench.py $wl --no-cpu-baseline --no-solve --warmup 3 --min-seconds 0.5 --also-stress off --also-config3 off 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readlines()[-1]); print('$lib', '$wl', 'ms_per_step', round(l['ms_per_step'], 4))"
  done
done
done
