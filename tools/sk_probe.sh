#!/bin/bash
for sk in 1 2 4 8; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-solve --split-k $sk --scene synthetic 2>/dev/null | tail -1 | SK=$sk python -c "
import sys,json,os
d=json.loads(sys.stdin.read())
print('split_k', os.environ['SK'], 'it/s', round(d['value']), {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done
