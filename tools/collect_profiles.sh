#!/bin/bash
# Evidence for profiles/<round> (run on the GPU box from the repo root; tools/run_collect.sh wraps it with the commit stamp):
# bench lines, rocprofv3 kernel statistics of the same commands, FETCH_SIZE / WRITE_SIZE passes, SQ / MFMA counter passes.
# Every profiler run sits under `timeout`; the program stands directly behind `--`.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
round=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${round}final; rm -rf "$out"; mkdir -p "$out"
stats() {   # stats <tag> <bench args...>: bench line under the kernel trace + the kernel statistics table
  local tag=$1; shift
  if timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag.d -- python3 bench.py "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/$tag.err; then
    local f; f=$(ls $out/$tag.d/*/*kernel_stats.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $out/${tag}_kernel_stats.csv; else echo "no kernel_stats for $tag" >&2; fi
  else echo "rocprofv3 stats run failed for $tag" >&2; fi
  rm -rf $out/$tag.d
}
pmc() {     # pmc <tag> <bench args...>: FETCH_SIZE and WRITE_SIZE in separate passes -> per-kernel KB and bytes per launch
  local tag=$1; shift
  local ok=1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -- python3 bench.py "$@" --min-seconds 0 --profile-steps 2 > /dev/null 2> $out/${tag}_$c.err || ok=0
  done
  if [ $ok = 1 ]; then python tools/pmc_aggregate.py $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE $out/${tag}_pmc_fetch_write_kb.json $out/${tag}_pmc_traffic.json $tag > /dev/null
  else echo "pmc pass failed for $tag" >&2; fi
  rm -rf $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE
}
Q="--no-cpu-baseline --no-solve --also-stress off --also-config3 off"      # the line alone (no sub-records)
# ---- config 2 (the benchmark line): one problem per GPU
pmc config2 --steps 40 --warmup 5 $Q
if [ -f $out/config2_pmc_traffic.json ]; then cp $out/config2_pmc_traffic.json profiles/pmc_traffic.json; fi
stats config2 $Q
# ---- the few-problem schedules: microseconds per step by the number of problems on the GPU
for n in 1 2 3 4 6 8 13 16 24 32 48; do
  python bench.py --problems $n $Q --min-seconds 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('problems', $n, 'us_per_step', round(1e3 * d['ms_per_step'], 2), 'problem_steps_per_s', round(d['value']))" >> $out/problems_sweep.txt
done
# ... the same for the 14-joint arm of config 5 (L = 14: twice the latent pairs per problem, a cache-resident 128^3 table so that the
#     sweep is about the schedules, not the table) and for config 3's shape (S = 7, M = 24, T = 70)
for n in 1 2 3 4 6 8 16 32 64; do
  python bench.py --workload stress --grid 128 --problems $n $Q --min-seconds 0.5 --steps 100 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('14-joint arm, problems', $n, 'us_per_step', round(1e3 * d['ms_per_step'], 2), 'problem_steps_per_s', round(d['value']))" >> $out/problems_sweep_L14.txt
done
for n in 1 2 4 8 16 32 55; do
  python bench.py --workload config3 --problems $n $Q --min-seconds 0.5 --steps 130 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 3 shape (S=7), problems', $n, 'us_per_step', round(1e3 * d['ms_per_step'], 2), 'problem_steps_per_s', round(d['value']))" >> $out/problems_sweep_S7.txt
done
# ---- the metric's plans/sec as wall time of solve_planning_problem() calls
timeout 600 python tools/solve_timing.py > $out/solve_timing_config2.txt 2>&1
# ---- config 3: Franka / bookshelves, the full C(11,2) = 55 start-goal batch, S=7 M=24 T=70
python bench.py --workload config3 --steps 130 --warmup 10 $Q > $out/config3_bench.json 2> $out/config3.err
stats config3 --workload config3 --steps 130 --warmup 10 $Q --min-seconds 0.5
pmc config3 --workload config3 --steps 130 --warmup 10 $Q
if [ -f $out/config3_pmc_traffic.json ]; then cp $out/config3_pmc_traffic.json profiles/pmc_traffic_config3.json; fi
# ---- 64 Franka problems per GPU (batch regime, cache-resident table)
stats franka64 $Q --problems 64 --scene synthetic --min-seconds 0.5 --steps 200
pmc franka64 $Q --problems 64 --steps 200
if [ -f $out/franka64_pmc_traffic.json ]; then cp $out/franka64_pmc_traffic.json profiles/pmc_traffic_franka64.json; fi
python bench.py $Q --problems 64 --scene synthetic --steps 200 > $out/franka64_bench.json 2>> $out/franka64.err
# ---- config 4: UR10, S = 1024 samples; one rank, and two ranks on this one GPU over gloo (rehearsal of the N > 1 path,
#      started by bench.py itself: no external launcher)
python bench.py --shard samples --steps 100 --warmup 10 > $out/config4_1rank_bench.json 2> $out/config4.err
stats config4_1rank --shard samples --steps 100 --warmup 10 --min-seconds 0.5
timeout 600 python bench.py --gpus 2 --shard samples --steps 100 --warmup 10 2>> $out/config4.err | tail -1 > $out/config4_2ranks_gloo_one_gpu_bench.json
# ---- the default N > 1 line (config 2 per GPU + the config-5 share as batch_512), two ranks on this one GPU: rehearsal only
timeout 900 python bench.py --gpus 2 --steps 100 --warmup 10 $Q 2> $out/gpus2.err | tail -1 > $out/gpus2_rehearsal_one_gpu_bench.json
# ---- config 5 share: 14-DoF arm, 512^3 voxels (2 GiB table), 64 problems; by free-space test of the SDF pass:
#      mask = bit masks in LDS (the default), summary = the brick summary in memory, none = every sphere gathers
B5="--workload stress --steps 200 --warmup 3 $Q --profile-steps 200 --min-seconds 0.5"      # a timed block = a whole plan from fresh models
for f in ${FORMS:-mask:on:off summary:off:on none:off:off}; do
  IFS=: read name mk sm <<< "$f"; tag=config5_$name
  stats $tag $B5 --mask $mk --summary $sm
  pmc $tag $B5 --mask $mk --summary $sm
  timeout 600 python bench.py $B5 --mask $mk --summary $sm --traffic-file $out/${tag}_pmc_traffic.json > $out/${tag}_bench.json 2>> $out/$tag.err
done
if [ -f $out/config5_mask_pmc_traffic.json ]; then cp $out/config5_mask_pmc_traffic.json profiles/pmc_traffic_config5.json; fi
# the SDF pass over a plan: first steps / whole plan, by form (tools/ab_mask.sh), and what a coarse free-space level could skip
timeout 1200 bash tools/ab_mask.sh "product" "on:off off:on off:off" > $out/sdf_pass_by_form.txt 2>&1
timeout 900 python tools/sdf_frames.py > $out/sdf_frames_config5.txt 2>&1
# ---- are the plans plans: clearance per query at the reference's own planner parameters, oracle beside the device
timeout 900 python tests/plan_report.py franka industrial --json $out/plan_report_franka_industrial.json > $out/plan_report_franka_industrial.txt 2>&1
# ---- SQ / MFMA counters: the fused prior kernel and the batch likelihood at config 5, the prior GEMM role at config 2
B5S="--workload stress --steps 20 --warmup 3 $Q --profile-steps 2"
# (the prior kernel ALONE: the two event-timed steps of --profile-steps run one launch per kernel; the timed steps run it with stage B behind its tiles)
tools/pmc_sq.sh "::prior_fused_split_kernel" $out/sq_prior_fused_config5 $B5S > /dev/null 2>&1
tools/pmc_sq.sh "::prior_split_cov_b_kernel" $out/sq_prior_cov_b_config5 $B5S > /dev/null 2>&1
tools/pmc_sq.sh "loglik_paths_mask_kernel<" $out/sq_loglik_config5 $B5S > /dev/null 2>&1
tools/pmc_sq.sh paths_bwd_regs $out/sq_paths_bwd_config5 $B5S > /dev/null 2>&1
tools/pmc_sq.sh "::cov_b_kernel<" $out/sq_cov_b_config5 $B5S > /dev/null 2>&1
tools/pmc_sq.sh stage2_kernel $out/sq_stage2_config2 --steps 40 --warmup 5 $Q > /dev/null 2>&1
# ---- the memory system's ceiling for 16-byte gathers
if [ -x tools/gather_probe ]; then
  timeout 300 tools/gather_probe > $out/gather_probe.txt 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/gp_fetch -- tools/gather_probe calib > $out/gather_probe_calib.txt 2>&1
  python - <<PY >> $out/gather_probe_calib.txt
import csv, glob
f = glob.glob("$out/gp_fetch/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "gather" in r["Kernel_Name"]]
n = 256 * 8 * 8 * 64 * 64 * 4
print("FETCH_SIZE per gather launch (KB):", v, "-> bytes tallied per random 16-byte gather:", [x * 1024 / n for x in v])
PY
  rm -rf $out/gp_fetch
fi
# ---- in-kernel time stamps of one config-2 step (measurement build: the stamps perturb the step by ~1-2 us; rocprof's kernel
#      durations above are the authority for totals, this shows where inside the launches the time goes)
if [ -f tools/libvgpmp_bisect.so ]; then
  VGPMP_HIP_LIB=$PWD/tools/libvgpmp_bisect.so timeout 300 python tools/step_trace.py 1 > $out/step_trace_config2.txt 2>&1
  # ... and the phases of one workgroup of the large-batch prior kernel at the config-5 share (draw | features | barrier | products | barrier)
  VGPMP_HIP_LIB=$PWD/tools/libvgpmp_bisect.so timeout 300 python tools/prior_trace.py > $out/prior_trace_config5.txt 2>&1
fi
# ---- the driver's line: config 2 + sub-records (batch_512, config3, batch_64) + plan quality + CPU baselines, with this
#      collection's traffic tables in place
python bench.py > $out/config2_bench.json 2> $out/config2_bench.err; tail -c 600 $out/config2_bench.json; echo
cp bench_detail.json $out/config2_bench_detail.json 2>/dev/null      # (the full record behind the compact line)
find $out -name "*.err" -size 0 -delete
ls -la $out | head -80
