#!/bin/bash
# Measurement aid: duration of each stage launch with roles / parts switched off (results are wrong on purpose).
export VGPMP_HIP_LIB=$PWD/tools/libvgpmp_bisect.so
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg tools/prof_fused.sh --allow-nan | grep -E "stage[123]_kernel"
done
