#!/bin/bash
# Measurement aid: N sessions of the one-kernel reproducer (test_lik_only.py) per library variant.
#   tools/flake_session/run_variants.sh <out file> <sessions> <name>=<lib path or "product"> ...
out=$1; n=$2; shift; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  fails=0
  for i in $(seq 1 "$n"); do
    if [ "$lib" = product ]; then unset VGPMP_HIP_LIB; else export VGPMP_HIP_LIB=$lib; fi
    log=$(timeout 300 python -m pytest tools/flake_session/${FLAKE_TEST:-test_lik_only.py} -q -s -p no:cacheprovider 2>&1)
    if echo "$log" | grep -qE "LIKELIHOOD ALONE DIFFERS|FIRST DIFFERENCE|failed"; then
      fails=$((fails + 1))
      echo "[$name session $i] $(echo "$log" | grep -m1 -E "LIKELIHOOD ALONE DIFFERS|FIRST DIFFERENCE|failed")" >> "$out"
      echo "$log" | grep -m1 'wrong entries per joint' >> "$out"
    else
      echo "[$name session $i] $(echo "$log" | grep -m1 -E "likelihood alone:|passed" || echo "$log" | tail -3)" >> "$out"
    fi
  done
  echo "== $name: $fails of $n sessions differ" >> "$out"
done
