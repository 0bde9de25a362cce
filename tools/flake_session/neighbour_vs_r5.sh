#!/bin/bash
# usage: nb.sh "<label>" <env assignments and command for the neighbour ...>
# Measurement aid (profiles/r06/flake.md, "What triggers it"): the 12-problem ELBO test of tests/test_gpu_attach.py on VICTIM_LIB (default: the
# round-5 sources built with vgpmp_debug_mfma_load, tools/libvgpmp_r5hook.so) while a neighbour process of choice runs.
label=$1; shift
env "$@" > /dev/null 2>&1 & bg=$!
sleep 2
r=$(VGPMP_HIP_LIB=${VICTIM_LIB:-tools/libvgpmp_r5hook.so} timeout 200 python -m pytest tests/test_gpu_attach.py -q -s -k "franka_x12" 2>&1 | grep -E "PARITY" | cut -c1-230)
kill $bg 2>/dev/null; wait $bg 2>/dev/null
echo "[$label] $r"
