#!/bin/bash
# stage-time scaling with problems per GPU (measurement aid)
for p in "$@"; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-solve --problems $p 2>/dev/null | tail -1 | P=$p python -c "
import sys,json,os
d=json.loads(sys.stdin.read())
print('problems', os.environ['P'], 'it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done
