#!/bin/bash
# Measurement aid: N sessions of tools/sweep_probe_mod per code object of tools/sweep_hsaco/ (tools/build_hsaco.sh).
#   tools/flake_session/run_hsaco.sh <out file> <sessions> <name> ...
out=$1; n=$2; shift; shift
mkdir -p "$(dirname "$out")"; : > "$out"
export SWEEP_PROBE=sweep_probe_mod FLAKE_SECONDS=${FLAKE_SECONDS:-6}
for name in "$@"; do
  fails=0
  for i in $(seq 1 "$n"); do
    log=$(SWEEP_HSACO=tools/sweep_hsaco/$name.hsaco timeout 100 python -m pytest tools/flake_session/test_sweep_probe.py -q -s -p no:cacheprovider 2>&1)
    if echo "$log" | grep -q "SWEEP DIFFERS"; then fails=$((fails + 1)); fi
    echo "[$name session $i] $(echo "$log" | grep -m1 "reference launch checksum") | $(echo "$log" | grep -c "SWEEP DIFFERS") launches differ | $(echo "$log" | grep -m1 "SWEEP DIFFERS" | cut -c1-150)" >> "$out"
    echo "$log" | grep -E "HIP error|Error|error" | head -2 >> "$out"
  done
  echo "== $name: $fails of $n sessions differ" >> "$out"
done
