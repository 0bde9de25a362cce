#!/bin/bash
# Run tools/collect_profiles.sh on a GPU box and file the results under profiles/r02/final, stamped with the commit they were
# taken at (the GPU box has no .git).  From the repo root, in the build container.
set -e
head=$(git rev-parse --short HEAD); dirty=$(git status --porcelain | grep -v '^??' | wc -l)
stamp="commit $head"; if [ "$dirty" != 0 ]; then stamp="$stamp + $dirty uncommitted file(s)"; fi
echo "$stamp" > profiles/COLLECT_STAMP
/usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -5 gpurun_out/collect.log'
dst=profiles/r02/final; rm -rf $dst; mkdir -p $dst
cp gpurun_out/r02final/*.json gpurun_out/r02final/*.csv gpurun_out/r02final/*.txt $dst/ 2>/dev/null || true
for d in gpurun_out/r02final/sq_*; do [ -f $d/summary.txt ] && cp $d/summary.txt $dst/$(basename $d).txt; done
echo "$stamp" > $dst/COLLECTED_AT
