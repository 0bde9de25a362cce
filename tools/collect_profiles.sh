#!/bin/bash
# Evidence for profiles/r01 (run on the GPU box from the repo root): bench line, rocprofv3 kernel statistics of the
# same command, FETCH_SIZE / WRITE_SIZE passes.  Every profiler run sits under `timeout`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r01; rm -rf $out; mkdir -p $out
python bench.py > $out/r01_bench.json 2> $out/bench.err; tail -c 600 $out/r01_bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/r01_bench_under_rocprof.json 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/r01_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats64 -- python3 bench.py --no-cpu-baseline --problems 64 --scene synthetic > $out/r01_bench_64problems_under_rocprof.json 2> $out/stats64.err
cp $(ls $out/stats64/*/*kernel_stats.csv | head -1) $out/r01_kernel_stats_64problems.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 2 > $out/pmc_$c.json 2> $out/pmc_$c.err
done
python tools/pmc_aggregate.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/r01_pmc_fetch_write_kb.json $out/pmc_traffic.json | head -30
rm -rf $out/stats $out/stats64 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
