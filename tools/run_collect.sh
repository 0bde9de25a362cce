#!/bin/bash
# Run tools/collect_profiles.sh on a GPU box and file the results under profiles/<round>/final, stamped with the commit they
# were taken at (the GPU box has no .git).  From the repo root, in the build container.  ROUND=r06 by default.
set -euo pipefail
round=${ROUND:-r06}
head=$(git rev-parse --short HEAD); dirty=$(git status --porcelain | grep -v '^??' | wc -l || true)
stamp="commit $head"; if [ "$dirty" != 0 ]; then stamp="$stamp + $dirty uncommitted file(s)"; fi
echo "$stamp" > profiles/COLLECT_STAMP
python -m vgpmp_amd.build > /dev/null && python -m vgpmp_amd.build --measurement > /dev/null      # product and measurement libraries of THIS tree
src=gpurun_out/${round}final
rm -rf "$src"
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-3000} -- "ROUND=$round bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -5 gpurun_out/collect.log" || true
[ -f "$src/config2_bench.json" ] || { echo "collection did not produce $src/config2_bench.json" >&2; exit 1; }
dst=profiles/$round/final; rm -rf "$dst"; mkdir -p "$dst"
cp "$src"/*.json "$src"/*.csv "$src"/*.txt "$dst"/ 2>/dev/null || true
for d in "$src"/sq_*; do if [ -f "$d/summary.txt" ]; then cp "$d/summary.txt" "$dst/$(basename "$d").txt"; fi; done
echo "$stamp" > "$dst/COLLECTED_AT"
# the table bench.py reads for roofline.traffic is THIS collection's (round 2 left the round-1 file in place)
cp "$dst/config2_pmc_traffic.json" profiles/pmc_traffic.json
[ -f "$dst/config5_mask_pmc_traffic.json" ] && cp "$dst/config5_mask_pmc_traffic.json" profiles/pmc_traffic_config5.json
[ -f "$dst/config3_pmc_traffic.json" ] && cp "$dst/config3_pmc_traffic.json" profiles/pmc_traffic_config3.json
[ -f "$dst/franka64_pmc_traffic.json" ] && cp "$dst/franka64_pmc_traffic.json" profiles/pmc_traffic_franka64.json
python - <<PY
import json
t = json.load(open("profiles/pmc_traffic.json"))
assert t["collected_at"] == "$stamp", (t["collected_at"], "$stamp")
print("profiles/pmc_traffic.json:", t["collected_at"])
PY
