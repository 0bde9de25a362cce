"""bench.py's own rank launcher (vgpmp_amd/launch.py) and the evidence files bench.py reads -- CPU only."""
import json
import os
import subprocess
import sys
import textwrap

from vgpmp_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_spawn_ranks_sets_the_rendezvous_environment_and_relays_rank0(tmp_path):
    out = tmp_path / "out"
    out.mkdir()
    script = _script(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "VGPMP_DIST_BACKEND")
        rec = {k: os.environ.get(k) for k in keys}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(sys.argv[1], "rank%s.json" % os.environ["RANK"]), "w").write(json.dumps(rec))
        print("line from rank", os.environ["RANK"])
    """)
    code = ("import sys; sys.path.insert(0, %r); from vgpmp_amd import launch; "
            "sys.exit(launch.spawn_ranks(3, [%r, '--x', '1'], script=%r, devices=1))" % (ROOT, str(out), script))
    env = {k: v for k, v in os.environ.items() if k != "VGPMP_DIST_BACKEND"}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    assert res.stdout.strip() == "line from rank 0"                      # only rank 0 reaches the parent's stdout
    assert "line from rank 1" in res.stderr and "line from rank 2" in res.stderr
    recs = [json.load(open(out / f"rank{r}.json")) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and all(r["WORLD_SIZE"] == "3" for r in recs)
    assert all(r["MASTER_ADDR"] == "127.0.0.1" and r["MASTER_PORT"] == recs[0]["MASTER_PORT"] for r in recs)
    assert all(r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for r in recs)
    assert all(r["VGPMP_DIST_BACKEND"] == "gloo" for r in recs)          # fewer devices than ranks: the rehearsal backend
    assert recs[1]["argv"] == [str(out), "--x", "1"]


def test_spawn_ranks_reports_a_failed_rank_and_ends_the_others(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)
    """)
    rc = launch.spawn_ranks(2, [], script=script, devices=8, timeout_s=50)
    assert rc == 7
    env = launch.rank_env(0, 2, 1234, base={}, devices=8)
    assert "VGPMP_DIST_BACKEND" not in env                               # enough devices: RCCL ("nccl")


def test_bench_gloo_world2_rendezvous_of_the_timed_region(tmp_path):
    """The barrier / repeat-count broadcast / MAX-over-ranks plumbing of bench.py's timed region with two gloo ranks on
    the CPU (the kernels replaced by a sleep): every rank runs the same number of blocks, the slowest rank sets the time."""
    script = _script(tmp_path, """
        import argparse, json, os, sys, time
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        torch.cuda.synchronize = lambda *a, **k: None
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        calls = []
        def run_steps(k):
            calls.append(k)
            time.sleep(0.01 * (1 + 2 * rank))
        args = argparse.Namespace(steps=5, min_seconds=0.1)
        elapsed, reps = bench.timed_region(run_steps, args, dist, "gloo")
        json.dump({"elapsed": elapsed, "reps": reps, "calls": len(calls)}, open(os.path.join(sys.argv[1], "r%%d.json" %% rank), "w"))
        dist.barrier(); dist.destroy_process_group()
    """ % ROOT)
    assert launch.spawn_ranks(2, [str(tmp_path)], script=script, devices=0, timeout_s=120) == 0
    r0, r1 = (json.load(open(tmp_path / f"r{r}.json")) for r in range(2))
    assert r0["reps"] == r1["reps"] and r0["calls"] == r1["calls"] == r0["reps"] + 1
    assert abs(r0["elapsed"] - r1["elapsed"]) < 1e-12 and r0["elapsed"] >= 0.03       # MAX over ranks: rank 1 sleeps 30 ms


def _strings(o, path=""):
    if isinstance(o, str):
        yield path, o
    elif isinstance(o, dict):
        for k, v in o.items():
            yield from _strings(v, f"{path}.{k}")
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o):
            yield from _strings(v, f"{path}[{i}]")


def test_bench_line_is_compact_and_carries_the_contract_keys():
    """the r05 line had grown to 20.7 KB and the driver could not parse it.  The ONE stdout line is the compact
    form (<= 6000 bytes, no string longer than 200 characters, `summary` last); everything else goes to bench_detail.json.  Checked
    on a committed full record of a real run (tests/golden/bench_detail_r05.json: the r05 line) and on a worst case with every
    optional sub-record present."""
    import copy
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r05.json")))
    worst = copy.deepcopy(full)
    worst["batch_512_one_gpu"] = copy.deepcopy(worst["batch_512"])
    worst["projection_8_ranks"] = {"rank_local_us_per_step": 73.0, "samples_per_rank": 128, "one_rank_us_per_step": 129.6,
                                   "predicted_speedup_at_8_ranks_before_the_collective": 1.77, "collective": "x" * 400,
                                   "stage_ms_one_of_8": {"a": 1.0}}
    worst["collective_breakdown"] = {"ranks": 8, "samples_per_rank": 128, "measured_us_per_step": 1.0, "rank_local_us_per_step": 1.0,
                                     "collective_us_per_step": 0.0, "how": "y" * 400}
    worst["config"] = {k: v + " " + "z" * 300 for k, v in worst["config"].items()}
    for rec in (full, worst):
        rec["summary"] = bench.summary_of(rec)
        line = bench.compact_line(rec)
        text = json.dumps(line)
        assert len(text) <= bench.LINE_LIMIT == 6000, len(text)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
            assert key in line, key
        assert line["config"]["workload"] and "model" not in line["config"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
        assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["openmp"]["value"] > 0
        assert list(line)[-1] == "summary"
        too_long = [(p, len(v)) for p, v in _strings(line) if len(v) > bench.STR_LIMIT]
        assert not too_long, too_long
        assert json.loads(text)["value"] == line["value"]
    assert "batch_512_one_gpu" in bench.compact_line(worst)["summary"]


def test_committed_traffic_table_is_this_rounds():
    """profiles/pmc_traffic.json (read by bench.py for roofline.traffic) must be a table produced by
    tools/pmc_aggregate.py with its commit stamp, not the round-1 file (which listed rocBLAS / ATen kernels)."""
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert t.get("collected_at", "").startswith("commit "), t.get("collected_at")
    assert not any(k.startswith("Cijk_") or "at::native::sigmoid" in k for k in t)
    lik = [k for k in t if k.startswith("loglik_paths_wide_kernel")]
    assert lik and all(t[k]["hbm_bytes_per_launch"] > 0 for k in lik)
    # round 4: the sub-records of the default line read their own tables, taken by the same collection
    t5 = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_config5.json")))
    t3 = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_config3.json")))
    assert t5["collected_at"] == t["collected_at"] == t3["collected_at"]
    assert t5["loglik_paths_mask_kernel<2, false>"]["hbm_bytes_per_launch"] > 1e8          # 2 GiB table: hundreds of MB of sectors per launch
    # round 5: the FETCH multiplier is calibrated per kernel against its algorithmic read volume (tools/kernel_bytes.py), not by name
    for k in ("paths_bwd_regs<25>", "paths_fwd_regs<2>", "cov_b_kernel<true, false>", "prior_fused_split_kernel<true, 2>"):
        assert t5[k]["fetch_multiplier"] in (1, 2) and t5[k]["algorithmic_read_bytes"] > 0 and "multiplier_basis" in t5[k], k
    assert t5["paths_bwd_regs<25>"]["fetch_multiplier"] == 2 and t5["loglik_paths_mask_kernel<2, false>"]["fetch_multiplier"] == 1
    assert any(k.startswith("prior_fused_small16_kernel") for k in t3)       # (the f16-split few-sample kernel)
    stamp = open(os.path.join(ROOT, "profiles", "r06", "final", "COLLECTED_AT")).read().strip()
    assert stamp == t["collected_at"]


def test_cpu_pool_of_the_problem_parallel_baseline(tmp_path):
    """bench.py's problem-parallel CPU baseline: worker processes (no torch, no GPU) started ahead of time, woken with the
    scene, timed together; aggregate problem-steps/s over the workers that finished."""
    import argparse
    import numpy as np
    import bench
    from vgpmp_amd import robots, scenes
    pool = bench.CpuPool(2)
    try:
        ps = robots.load_problemset("franka", "industrial")
        spec = robots.load_robot("franka", *ps.robot_pos_and_orn)
        grid = scenes.synthetic_boxes_sdf(n=24, delta=0.1, origin=(-1.2, -1.2, -0.6), seed=0)
        args = argparse.Namespace(samples=4, timesteps=8, inducing=4)
        pool.go(ps, spec, grid, args)
        pool.wait_ready(timeout_s=120)
        assert pool.ready == [0, 1]
        omp = pool.run_omp(0.6)            # the compiled restatement: one thread, OpenMP over the pool's cores, a problem per core
        rec = pool.run(0.5)
    finally:
        if any(p.poll() is None for p in pool.procs):
            pool.close()
    assert rec["cores"] == 2 and rec["processes"] == 2 and rec["unit"] == "problem-steps/sec"
    assert rec["value"] > 0 and abs(rec["per_process"] * 2 - rec["value"]) < 1e-9
    assert all(p.poll() == 0 for p in pool.procs) and pool.omp.poll() == 0
    assert omp["kind"] == "port" and omp["threads"] == 2 and omp["value"] > 0 and omp["single_thread"]["value"] > 0
    assert omp["problem_parallel"]["value"] > 0 and omp["problem_parallel"]["threads"] == 2 and "cpu_step.cpp" in omp["sample"]
