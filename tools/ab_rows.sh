for rep in 1 2; do
for lib in head new; do
  if [ $lib = head ]; then export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_head.so; else unset VGPMP_HIP_LIB; fi
  for wl in "--workload config3 --steps 130" "--workload stress --steps 200" "--problems 64 --steps 200"; do
    python bench.py $wl --no-cpu-baseline --no-solve --warmup 3 --min-seconds 0.5 --also-stress off --also-config3 off 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readlines()[-1]); print('$lib', '$wl', 'ms_per_step', round(l['ms_per_step'], 4))"
  done
done
done
