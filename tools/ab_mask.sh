#!/bin/bash
# Measurement aid: the SDF pass of the config-5 share by free-space test: brick summary (global), masks in LDS, both.
#   tools/ab_mask.sh [bench args]
cd "$(dirname "$0")/.."
for rep in 1 2; do for st in 10 200; do for form in "off on" "on off" "on on" "off off"; do
  set -- $form
  python bench.py --workload stress --mask $1 --summary $2 --steps $st --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10 --min-seconds 0.5 --allow-nan 2>/dev/null | tail -1 | M=$1 S=$2 ST=$st python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print('mask', os.environ['M'], 'summary', os.environ['S'], 'steps', os.environ['ST'], '| loglik us', round(1e3 * d['stage_ms']['loglik_fk_sdf'], 1), '| step ms', round(d['ms_per_step'], 4))"
done; done; done
