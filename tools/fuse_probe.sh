#!/bin/bash
# Measurement aid: single-problem bench, fused stage launches vs one launch per kernel, graph and eager.
for nf in "" "--no-fuse"; do
  for un in 10 0; do
    python bench.py --scene synthetic --no-cpu-baseline --no-solve $nf --unroll $un "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse=${nf:-yes} unroll $un', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us')"
  done
done
