#!/usr/bin/env python3
"""ELBO iterations/sec + plans/sec of the MI355X-native vGPMP hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...                      # starts N fresh rank processes itself (vgpmp_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W            # or under an external launcher

A "step" is one full optimisation step of one planning problem: fresh Philox noise, ELBO forward
(Kuu/Kuf + Cholesky, RFF prior, Matheron update, FK, SDF lookup, hinge likelihood, KL), its reverse
pass and the Adam update (reference utils/miscellaneous.py:68-84).  Default workload at every N:
BASELINE config 2 -- Franka 7-DoF, industrial scene (SDF generated on the device from the reference's
industrial collision mesh: its .sdf blobs are missing from the checkout), ONE start-goal problem per
GPU, S=128, M=30, T=100, B=1024.  Ranks hold independent problems (no data-path collective): weak
scaling, value = total steps / max time over ranks.  With N > 1 the line also carries `batch_512`:
BASELINE config 5's share of the 512-problem batch (64 problems of the 14-DoF arm per GPU, 512^3 table).

Other workloads: --workload config3 (Franka / bookshelves, all 55 start-goal pairs as one batch),
--workload stress (config 5 share), --shard samples (config 4: UR10, S=1024 split over the ranks,
one all-reduce of the gradient buffer per step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBPS = 6290.0    # same guide: measured float4 copy
SUMMARY = {"auto": None, "on": True, "off": False}
F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_16x16x4_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2516.6   # v_mfma_f32_16x16x32_f16: 1024 flop / cycle / SIMD x 1024 SIMDs x 2.4 GHz (the guide's ~2.5 PF dense)
METRIC = "ELBO iters/sec + plans/sec, Franka-7DoF industrial S=128 M=30 T=100, 1/2/4/8 GPU"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=0, help="Monte-Carlo samples (0 = the workload's own: 128 / 7 / 1024)")
    ap.add_argument("--inducing", type=int, default=0, help="inducing points (0 = the workload's own)")
    ap.add_argument("--timesteps", type=int, default=0, help="time stamps (0 = the workload's own)")
    ap.add_argument("--problems", type=int, default=0, help="problems per GPU (0 = the workload's own: config 2: 1, config 3: 55, stress: 64)")
    ap.add_argument("--grid", type=int, default=0, help="SDF voxels per axis of the synthetic scenes (0 = 128; stress: 512)")
    ap.add_argument("--split-k", type=int, default=0, help="K-slices of the prior GEMM (0 = engine default)")
    ap.add_argument("--no-fuse", action="store_true", help="one launch per kernel even for small batches (measurement)")
    ap.add_argument("--also-train", default="", help="comma list of further trainable_params flags (inducing_variable, sigma_obs, "
                    "alpha) on top of the reference's defaults: the schedules those variables run on (config 2 / 3 workloads)")
    ap.add_argument("--scene", choices=("mesh", "industrial", "synthetic"), default="mesh",
                    help="mesh = SDF generated from the reference's collision mesh of the workload's scene; synthetic = boxes/spheres")
    ap.add_argument("--unroll", type=int, default=0,
                    help="steps per captured hipGraph; 0 = plain launches (measured 2-3 %% faster than graph replay here)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the measured solve_planning_problem() calls (plans/sec)")
    ap.add_argument("--profile-steps", type=int, default=40)
    ap.add_argument("--allow-nan", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--traffic-file", default="", help="pmc traffic table (tools/pmc_aggregate.py) for roofline.traffic")
    ap.add_argument("--shard", choices=("problems", "samples"), default="problems",
                    help="problems: independent problems per GPU, no collective (config 2 / 3 / 5); samples: BASELINE config 4, "
                         "UR10, one problem, 1024 Monte-Carlo samples split over the ranks, one all-reduce per step")
    ap.add_argument("--collective", choices=("torch", "capi"), default="torch",
                    help="samples sharding: torch.distributed all_reduce (RCCL) or the C ABI's vgpmp_allreduce_grads (RCCL)")
    ap.add_argument("--layout", choices=("brick", "linear"), default="brick", help="voxel table layout (include/vgpmp.h)")
    ap.add_argument("--summary", choices=("auto", "on", "off"), default="auto",
                    help="free-space brick summary in the batch likelihood kernel (auto: tables beyond the Infinity Cache)")
    ap.add_argument("--mask", choices=("auto", "on", "off"), default="auto",
                    help="free-space bit masks held in LDS by the batch likelihood kernel (auto: as --summary)")
    ap.add_argument("--lik-form", choices=("auto", "lanes", "lanes-lds"), default="auto",
                    help="likelihood kernel form (measurement): lanes = the batch form at any batch size, lanes-lds = with its per-frame sums in LDS")
    ap.add_argument("--flags", type=int, default=0, help="extra VGPMP_* measurement flags (include/vgpmp.h) OR-ed into every step")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the timed region repeats the K steps until this much time has passed (K = --steps alone is ~10 ms)")
    ap.add_argument("--workload", choices=("config2", "config3", "stress"), default="config2",
                    help="config2 = the benchmark line; config3 = Franka / bookshelves, 55 pairs, S=7 M=24 T=70; "
                         "stress = BASELINE config 5 per-GPU share (64 problems, 512^3 table)")
    ap.add_argument("--also-stress", choices=("auto", "on", "off"), default="auto",
                    help="append the config-5 share (batch_512) to the line; auto = with the default workload, at every N")
    ap.add_argument("--also-config3", choices=("auto", "on", "off"), default="auto",
                    help="append BASELINE config 3 (55 pairs) and 64 problems of config 2's shape (batch_64); auto = default workload at N = 1")
    return ap.parse_args(argv)


WORKLOAD_DEFAULTS = {            # samples, inducing, timesteps, problems per GPU, grid
    "config2": (128, 30, 100, 1, 128),
    "config3": (7, 24, 70, 55, 128),
    "stress": (128, 30, 100, 64, 512),
}


def resolve(args):
    """Fill in the workload's own sizes where the command line left them at 0."""
    s, m, n, p, g = WORKLOAD_DEFAULTS[args.workload]
    if args.shard == "samples":
        s, m, n, p = 1024, 18, 70, 1                       # BASELINE config 4 (data/problemsets/ur10.py:71-84)
    args.samples = args.samples or s
    args.inducing = args.inducing or m
    args.timesteps = args.timesteps or n
    args.problems = args.problems or p
    args.grid = args.grid or g
    return args


def build_problem(rank: int, args, world: int = 1):
    import numpy as np
    import torch
    from vgpmp_amd import engine, robots, scenes
    ps = robots.load_problemset("franka", "industrial")
    pp = ps.planner_params
    if args.workload == "stress":
        # BASELINE config 5, one GPU's share: synthetic 14-DoF arm, 512^3 float4 table (2 GiB > Infinity Cache),
        # random start-goal pairs in +-2 rad (seed 0)
        spec = robots.synthetic_arm(14)
        rows = scenes.AnalyticSceneRows(args.grid, 2.0 / args.grid, (-1.0, -1.0, -1.0), seed=0, n_boxes=24, n_spheres=16,
                                        round_to=torch.float32)      # evaluated on the device, slab by slab
        grid = (rows, rows.origin, rows.delta)
        rng = np.random.default_rng(rank)
        qs = rng.uniform(-2.0, 2.0, (args.problems, 2, 14))
        scene = engine.DeviceScene(spec, grid, (0.0, 0.0, 0.0), sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"],
                                   layout=args.layout, free_space_summary=SUMMARY[args.summary],
                                   free_space_mask=SUMMARY[args.mask])
        planner = engine.PlannerBatch(scene, qs, num_samples=args.samples, num_inducing=args.inducing,
                                      num_data=args.timesteps, num_bases=1024, lengthscales=[2.0] * 14, variance=0.2,
                                      alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=1234,
                                      problem_base=rank * args.problems)
        return ps, spec, grid, scene, planner
    if args.shard == "samples":
        # BASELINE config 4: UR10 6-DoF, industrial, ONE problem whose S Monte-Carlo samples are split over the ranks;
        # one in-place all-reduce of the contiguous [gradient | lik | kl] buffer per step (vgpmp_amd/sharding.py)
        from vgpmp_amd import sharding
        ps = robots.load_problemset("ur10", "industrial")
        pp = ps.planner_params
        spec = robots.load_robot("ur10", *ps.robot_pos_and_orn)
        grid = scenes.scene_sdf("industrial", delta=0.0125, padding=20) if args.scene != "synthetic" else \
            scenes.synthetic_boxes_sdf(n=args.grid, delta=1.6 / args.grid, origin=(-0.8, -0.8, -0.2), seed=0)
        scene = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"],
                                   layout=args.layout, free_space_summary=SUMMARY[args.summary],
                                   free_space_mask=SUMMARY[args.mask])
        s_loc, s_off = sharding.shard_samples(args.samples, world, rank)
        planner = engine.PlannerBatch(scene, np.array([ps.queries[0]]), num_samples=s_loc, samples_total=args.samples,
                                      sample_offset=s_off, kl_scale=1.0 if rank == 0 else 0.0,
                                      num_inducing=args.inducing, num_data=args.timesteps, num_bases=1024,
                                      lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                                      learning_rate=pp["learning_rate"], seed=1234, split_k=args.split_k or None)
        return ps, spec, grid, scene, planner
    scene_name = "industrial"
    if args.workload == "config3":
        # BASELINE config 3: Franka, bookshelves offset (0.62, -0.15, 0.834), all C(11,2) = 55 start-goal pairs in one
        # problem-parallel batch, S=7 M=24 N=70, lr 0.09 (data/problemsets/franka.py:11-24,91-104)
        ps = robots.load_problemset("franka", "bookshelves")
        pp = ps.planner_params
        scene_name = "bookshelves"
    spec = robots.load_robot("franka", *ps.robot_pos_and_orn)
    if args.scene != "synthetic":
        # the reference's scene: grid generated on the device from its collision mesh (vgpmp_mesh_sdf),
        # SDFGen-style extent (bounding box + 20 cells) at 1.25 cm (industrial: 129 x 154 x 80 voxels)
        grid = scenes.scene_sdf(scene_name, delta=0.0125, padding=20)
    else:
        grid = scenes.synthetic_boxes_sdf(n=args.grid, delta=1.6 / args.grid, origin=(-0.8, -0.8, -0.2), seed=0)
    scene = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"],
                               layout=args.layout, free_space_summary=SUMMARY[args.summary],
                                   free_space_mask=SUMMARY[args.mask])
    queries = ps.queries
    qs = np.array([queries[(rank * args.problems + i) % len(queries)] for i in range(args.problems)])
    planner = engine.PlannerBatch(scene, qs, num_samples=args.samples, num_inducing=args.inducing,
                                  num_data=args.timesteps, num_bases=1024, lengthscales=pp["lengthscales"],
                                  variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"],
                                  seed=1234, problem_base=rank * args.problems, split_k=args.split_k or None,
                                  trainable=(dict(engine.DEFAULT_TRAINABLE, **{k: True for k in args.also_train.split(",") if k})
                                             if args.also_train else None))
    planner.fuse = not args.no_fuse
    return ps, spec, grid, scene, planner


def cpu_baseline(ps, spec, grid, args, budget_s: float = 10.0, pool=None):
    """The float64 NumPy oracle (a restatement, not the GPflow/TF stack) timed on the host cores: one process on the BLAS pool
    (`cores` = the cores it actually kept busy: process CPU time / wall time), the same with the pool limited to ONE thread, and
    -- `pool`, started before this process touched the GPU -- one single-threaded process per core over independent problems
    (the axis the GPU batches): aggregate problem-steps/s (SURVEY 8(d) "CPU baseline timing")."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_scene
    from oracle import vgpmp_oracle as orc
    pp = ps.planner_params
    if pool is not None:
        pool.go(ps, spec, grid, args)                  # the workers build their problems while this process times its own
    sc = oracle_scene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    y = np.array(ps.queries[0], dtype=np.float64)
    S, N, M, B, D = args.samples, args.timesteps, args.inducing, 1024, spec.dof
    p = orc.init_params(sc.robot, y, M, pp["lengthscales"], pp["variance"])
    st = orc.adam_init(p)
    X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
    rng = np.random.default_rng(0)
    one = lambda: orc.optimization_step(p, st, sc, X, Zy, y, orc.draw_noise(rng, S, D, D, B, M + 2),
                                        float(pp["alpha"]), float(pp["learning_rate"]))

    def timed(budget):
        one()                                   # warm caches / BLAS threads
        n, t0, c0 = 0, time.perf_counter(), time.process_time()
        while True:
            one(); n += 1
            el = time.perf_counter() - t0
            if el > budget and n >= 3:
                return n, el, (time.process_time() - c0) / el

    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        threadpool_limits, threads = None, os.cpu_count() or 1
    if pool is not None:
        pool.wait_ready()                        # (their set-up must not share the cores with the single-process timing below)
    n, el, busy = timed(budget_s)
    out = {"value": n / el, "unit": "ELBO iters/sec", "cores": max(1, int(round(busy))), "kind": "port",
           "blas_threads": int(threads), "cores_busy_measured": round(busy, 2), "logical_cores": os.cpu_count(),
           "sample": f"{n} full steps of the same workload (draw + forward + reverse + Adam), float64 NumPy oracle, {el:.1f} s, one process, "
                     f"{busy:.1f} cores busy",
           "sample_detail": f"BLAS pool of {threads} threads of which {busy:.1f} cores were busy on average (process CPU time / wall time: "
                            f"the restatement is element-wise NumPy, it does not thread); host has {os.cpu_count()} logical cores"}
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n1, el1, _ = timed(budget_s)
        out["single_thread"] = {"value": n1 / el1, "cores": 1,
                                "sample": f"{n1} steps in {el1:.1f} s with the BLAS pool limited to one thread"}
    if pool is not None:
        out["openmp"] = pool.run_omp(min(budget_s, 6.0))
        out["problem_parallel"] = pool.run(budget_s)
    return out


def oracle_plan_check(n_queries: int = 3):
    """The float64 oracle run as a PLANNER on the first queries of the industrial problem set at the reference's own planner
    parameters (data/problemsets/franka.py:77-90), noise = the device's Philox stream (seed 0, problem = query, step): the
    clearance of its posterior-mean path after num_steps, to be read beside the device's (tests/plan_report.py; the same
    comparison is asserted by tests/test_gpu_plans.py).  Part of the CPU leg: the only place bench.py runs the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import plan_report
    rep, (ps, spec, grid, pp, pl) = plan_report.device_report("franka", "industrial")
    out = []
    for k in range(n_queries):
        o = plan_report.oracle_plan(ps, spec, grid, pp, k)
        d = rep["queries"][k]
        out.append({"query": k, "oracle_mean_path_clearance": round(o["mean_path"], 4), "device_mean_path_clearance": round(d["mean_path"], 4),
                    "device_best_sample_clearance": round(d["best_sample"], 4), "oracle_loss_first_last": [round(o["loss_first"], 1), round(o["loss_last"], 1)],
                    "device_loss_first_last": [round(d["loss_first"], 1), round(d["loss_last"], 1)]})
    return {"planner_params": "the reference's own (S=20 M=10 T=50, 200 steps, lr 0.02)", "queries": out}


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host may
    own far fewer: the r04 GPU box gave 64 single-threaded workers the throughput of ~9.4 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


class CpuPool:
    """One single-threaded oracle process per core, over independent start-goal problems of the bench workload.  The children
    are started BEFORE the parent touches the GPU (fresh interpreters of this script in `--cpu-worker` mode; they never import
    torch) and sleep until `go()` hands them the scene: they cost nothing while the GPU part is timed."""

    def __init__(self, n: int):
        import subprocess
        import tempfile
        self.n = int(n)
        self.dir = tempfile.mkdtemp(prefix="vgpmp_cpu_pool_")
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", NUMEXPR_NUM_THREADS="1")
        self.err = open(os.path.join(self.dir, "workers.err"), "w")
        self.procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", self.dir, str(i)], env=env,
                                       stdout=subprocess.DEVNULL, stderr=self.err) for i in range(self.n)]
        # ... and ONE process for the compiled OpenMP restatement (oracle/cpu_step.cpp): its own interpreter so that its OpenMP
        # runtime is the only one in the process, waiting passively between parallel regions (spinning threads of an idle pool
        # cost a container with a CPU quota most of its cores)
        env_omp = dict(os.environ, OMP_WAIT_POLICY="passive", OMP_NUM_THREADS=str(self.n), OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        self.omp = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", self.dir, "-1"], env=env_omp,
                                    stdout=subprocess.DEVNULL, stderr=self.err)

    def go(self, ps, spec, grid, args) -> None:
        import numpy as np
        data, origin, delta = grid
        np.save(os.path.join(self.dir, "grid.npy"), np.ascontiguousarray(np.asarray(data, dtype=np.float64)))
        job = {"robot": ps.robot, "problemset": ps.name, "origin": [float(v) for v in origin], "delta": float(delta),
               "S": args.samples, "N": args.timesteps, "M": args.inducing, "B": 1024}
        tmp = os.path.join(self.dir, "go.tmp")
        json.dump(job, open(tmp, "w"))
        os.replace(tmp, os.path.join(self.dir, "go.json"))

    def _wait(self, pattern: str, timeout_s: float):
        t0 = time.time()
        while time.time() - t0 < timeout_s:
            have = [i for i in range(self.n) if os.path.exists(os.path.join(self.dir, pattern % i))]
            alive = [p.poll() is None for p in self.procs]
            if len(have) == self.n or not any(alive):
                return have
            time.sleep(0.05)
        return [i for i in range(self.n) if os.path.exists(os.path.join(self.dir, pattern % i))]

    def wait_ready(self, timeout_s: float = 120.0):
        self.ready = self._wait("ready_%d", timeout_s)
        t0 = time.time()      # (the compiled restatement builds its library for this host on first use: a few seconds of one core)
        while not os.path.exists(os.path.join(self.dir, "ready_omp")) and self.omp.poll() is None and time.time() - t0 < timeout_s:
            time.sleep(0.05)

    def run_omp(self, budget_s: float):
        """The compiled restatement (oracle/cpu_step.cpp) on one thread, on all usable cores (OpenMP inside one problem) and as
        one single-threaded problem per core; the other workers sleep meanwhile."""
        json.dump({"budget": budget_s, "threads": self.n}, open(os.path.join(self.dir, "start_omp.tmp"), "w"))
        os.replace(os.path.join(self.dir, "start_omp.tmp"), os.path.join(self.dir, "start_omp.json"))
        path = os.path.join(self.dir, "result_omp.json")
        t0 = time.time()
        while not os.path.exists(path) and self.omp.poll() is None and time.time() - t0 < 6 * budget_s + 120:
            time.sleep(0.05)
        if not os.path.exists(path):
            return {"value": None, "error": "the OpenMP worker did not finish (see " + self.err.name + ")"}
        return json.load(open(path))

    def run(self, budget_s: float):
        json.dump({"budget": budget_s}, open(os.path.join(self.dir, "start.tmp"), "w"))
        os.replace(os.path.join(self.dir, "start.tmp"), os.path.join(self.dir, "start.json"))
        done = self._wait("result_%d.json", budget_s * 3 + 60.0)
        res = [json.load(open(os.path.join(self.dir, "result_%d.json" % i))) for i in done]
        self.close()
        if not res:
            return {"value": None, "error": "no worker finished (see " + self.err.name + ")"}
        steps, wall = sum(r["steps"] for r in res), max(r["elapsed"] for r in res)
        busy = sum(r.get("cpu_seconds", r["elapsed"]) / r["elapsed"] for r in res)
        return {"value": steps / wall, "unit": "problem-steps/sec", "cores": max(1, int(round(busy))), "processes": self.n, "kind": "port",
                "per_process": steps / wall / len(res), "cores_busy_measured": round(busy, 1), "usable_cores": usable_cores(),
                "sample": f"{len(res)} single-threaded processes (one per usable core: {usable_cores()} by affinity / cgroup quota; host has "
                          f"{os.cpu_count()} logical cores; {busy:.1f} cores busy by the workers' own CPU time), each the same "
                          f"float64 oracle step on its own start-goal problem of the workload's problem set, all timed together "
                          f"for {wall:.1f} s: {steps} problem-steps; to be compared with the GPU BATCH figure (batch_64), not with the "
                          f"one-problem line"}

    def close(self) -> None:
        open(os.path.join(self.dir, "cancel"), "w").close()
        for p in self.procs + [self.omp]:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.err.close()
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


def cpu_omp_worker_main(jobdir: str) -> int:
    """`bench.py --cpu-worker <dir> -1`: the compiled OpenMP restatement (oracle/cpu_step.cpp, kind "port") timed on the same
    workload -- noise draw (compiled, std::mt19937_64) + forward + reverse + Adam per step: one thread; all usable cores inside
    one problem; one single-threaded problem per core (Python threads around the GIL-free C call)."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_scene
    from oracle import cpu_step
    from oracle import vgpmp_oracle as orc
    from vgpmp_amd import robots
    parent = os.getppid()

    def wait_for(name, timeout_s):
        t0 = time.time()
        path = os.path.join(jobdir, name)
        while not os.path.exists(path):
            if os.path.exists(os.path.join(jobdir, "cancel")) or os.getppid() != parent or time.time() - t0 > timeout_s:
                return None
            time.sleep(0.05)
        return json.load(open(path))

    job = wait_for("go.json", 3600.0)
    if job is None:
        return 0
    cpu_step.load()                                           # (builds the library for this host's CPU if it is not there)
    ps = robots.load_problemset(job["robot"], job["problemset"])
    pp = ps.planner_params
    spec = robots.load_robot(job["robot"], *ps.robot_pos_and_orn)
    data = np.load(os.path.join(jobdir, "grid.npy"))
    sc = oracle_scene(spec, (data, np.array(job["origin"]), job["delta"]), ps.object_positions[0], sigma_obs=pp["sigma_obs"],
                      epsilon=pp["epsilon"])
    S, N, M, B, D = job["S"], job["N"], job["M"], job["B"], spec.dof
    X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
    alpha, lr = float(pp["alpha"]), float(pp["learning_rate"])

    def problem(k):
        y = np.array(ps.queries[k % len(ps.queries)], dtype=np.float64)
        return cpu_step.Problem(sc, X, Zy, y, orc.init_params(sc.robot, y, M, pp["lengthscales"], pp["variance"]))

    def run(pr, nb, threads, budget, seed0):
        pr.step(nb.draw(seed0, threads), alpha, lr, threads=threads)
        n, t0 = 0, time.perf_counter()
        while True:
            pr.step(nb.draw(seed0 + 1 + n, threads), alpha, lr, threads=threads)
            n += 1
            el = time.perf_counter() - t0
            if el > budget and n >= 3:
                return n, el

    problem(0).step(cpu_step.NoiseBuffers(S, D, D, B, M + 2).draw(0, 1), alpha, lr, threads=1)      # warm
    open(os.path.join(jobdir, "ready_omp"), "w").close()
    start = wait_for("start_omp.json", 3600.0)
    if start is None:
        return 0
    budget, T = float(start["budget"]), int(start["threads"])
    nb = cpu_step.NoiseBuffers(S, D, D, B, M + 2)
    n1, e1 = run(problem(0), nb, 1, budget / 2, 0)
    nT, eT = run(problem(0), nb, T, budget / 2, 1000)
    counts, t_par = [0] * T, [0.0] * T

    def worker(i):
        counts[i], t_par[i] = run(problem(i), cpu_step.NoiseBuffers(S, D, D, B, M + 2), 1, budget / 2, 10000 * (i + 1))

    c0 = time.process_time()
    w0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    wall = time.perf_counter() - w0
    busy = (time.process_time() - c0) / wall
    res = {"value": nT / eT, "unit": "ELBO iters/sec", "threads": T, "kind": "port",
           "single_thread": {"value": n1 / e1, "threads": 1},
           "problem_parallel": {"value": sum(counts) / max(t_par), "unit": "problem-steps/sec", "threads": T,
                                "cores_busy_measured": round(busy, 1)},
           "sample": f"oracle/cpu_step.cpp (float64 C++ / OpenMP restatement of the same step, g++ -O3 -march=native; cross-checked against "
                     f"the NumPy oracle to 1e-9 by tests/test_oracle_cpu_step.py): {n1} steps in {e1:.1f} s on one thread; {nT} steps in "
                     f"{eT:.1f} s with {T} OpenMP threads inside the one problem; {sum(counts)} problem-steps in {max(t_par):.1f} s as {T} "
                     f"single-threaded problems side by side; every step draws its own noise (compiled generator)"}
    tmp = os.path.join(jobdir, "result_omp.tmp")
    json.dump(res, open(tmp, "w"))
    os.replace(tmp, os.path.join(jobdir, "result_omp.json"))
    return 0


def cpu_worker_main(jobdir: str, index: int) -> int:
    """`bench.py --cpu-worker <dir> <i>`: one process of CpuPool.  NumPy + the oracle only (no torch, no GPU)."""
    if index < 0:
        return cpu_omp_worker_main(jobdir)
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_scene
    from oracle import vgpmp_oracle as orc
    from vgpmp_amd import robots
    parent = os.getppid()

    def wait_for(name, timeout_s):
        t0 = time.time()
        path = os.path.join(jobdir, name)
        while not os.path.exists(path):
            if os.path.exists(os.path.join(jobdir, "cancel")) or os.getppid() != parent or time.time() - t0 > timeout_s:
                return None
            time.sleep(0.05)
        return json.load(open(path))

    job = wait_for("go.json", 3600.0)
    if job is None:
        return 0
    ps = robots.load_problemset(job["robot"], job["problemset"])
    pp = ps.planner_params
    spec = robots.load_robot(job["robot"], *ps.robot_pos_and_orn)
    data = np.load(os.path.join(jobdir, "grid.npy"), mmap_mode="r")
    sc = oracle_scene(spec, (data, np.array(job["origin"]), job["delta"]), ps.object_positions[0], sigma_obs=pp["sigma_obs"],
                      epsilon=pp["epsilon"])
    queries = ps.queries
    y = np.array(queries[index % len(queries)], dtype=np.float64)
    S, N, M, B, D = job["S"], job["N"], job["M"], job["B"], spec.dof
    p = orc.init_params(sc.robot, y, M, pp["lengthscales"], pp["variance"])
    st = orc.adam_init(p)
    X, Zy = orc.init_trainset(N, D), orc.inducing_Zy(M, D)
    rng = np.random.default_rng(index)
    one = lambda: orc.optimization_step(p, st, sc, X, Zy, y, orc.draw_noise(rng, S, D, D, B, M + 2),
                                        float(pp["alpha"]), float(pp["learning_rate"]))
    one()
    open(os.path.join(jobdir, "ready_%d" % index), "w").close()
    start = wait_for("start.json", 600.0)
    if start is None:
        return 0
    n, t0, c0 = 0, time.perf_counter(), time.process_time()
    while True:
        one(); n += 1
        el = time.perf_counter() - t0
        if el > start["budget"] and n >= 3:
            break
    tmp = os.path.join(jobdir, "result_%d.tmp" % index)
    json.dump({"steps": n, "elapsed": el, "cpu_seconds": time.process_time() - c0}, open(tmp, "w"))
    os.replace(tmp, os.path.join(jobdir, "result_%d.json" % index))
    return 0


def measured_solves(args, world, rank, dist, backend):
    """plans/sec as the metric words it: complete solve_planning_problem() calls through the reference-shaped surface
    (gpflow_vgpmp.utils.miscellaneous, utils/miscellaneous.py:141-321 minus GUI / simulated execution: model construction,
    disable_param_opt, num_steps optimisation steps, 150 posterior paths, best-sample pick, clearance check), one after
    another as benchmarking.py:68-85 does, and the whole problem set as ONE device batch (solve_planning_problems_batched).
    BASELINE config 2 sizes (S=128, M=30, T=100) on the reference's industrial problem set."""
    import warnings
    import numpy as np
    import torch
    ns = {}
    exec("from gpflow_vgpmp.utils.miscellaneous import *", ns)
    from gpflow_vgpmp.utils.simulation_manager import SimulationManager
    import contextlib
    # (the reference's parameter loader prints the size of the problem set: that belongs on stderr here -- stdout carries the one JSON line)
    with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
        warnings.simplefilter("ignore")
        env = SimulationManager(file_path=os.path.join(ROOT, "parameters.yaml"))
    env.config["planner_params"].update(num_samples=args.samples, num_inducing=args.inducing, time_spacing_X=args.timesteps)
    pp = env.config["planner_params"]
    solve, batched = ns["solve_planning_problem"], ns["solve_planning_problems_batched"]
    queries = env.config["scene_params"]["queries"]
    dof = env.robot.dof
    mine = [queries[(rank * 7 + i) % len(queries)] for i in range(7)]      # one warm-up + six timed, per rank
    per, solved = [], 0
    devnull = open(os.devnull, "w")
    for k, (a, b) in enumerate(mine):
        a = np.array(a, dtype=np.float64).reshape(1, dof)
        b = np.array(b, dtype=np.float64).reshape(1, dof)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, sys.stdout = sys.stdout, devnull          # the reference prints "Starting training...." per call
        try:
            ok, traj = solve(env=env, start_joints=a, end_joints=b)
        finally:
            sys.stdout = out
        torch.cuda.synchronize()
        if k:
            per.append(time.perf_counter() - t0)
            solved += bool(ok)
    batched(env, queries[:4])                                                # warm (allocations of the batch shapes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = batched(env, queries)
    torch.cuda.synchronize()
    tb = time.perf_counter() - t0
    # are the plans plans?  per query: clearance of its own end states, of the straight line it starts from, of the best sample
    rep = {}
    res_rep = batched(env, queries, report=rep)
    quality = plan_quality(rep, [bool(r[0]) for r in res_rep])
    tot = float(sum(per))
    if dist is not None:
        t = torch.tensor([tot, tb], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tot, tb = float(t[0]), float(t[1])
    return {"plan_quality": quality,
            "plans_per_sec_measured": world * len(per) / tot,
            "plans_per_sec_batched_measured": world * len(queries) / tb,
            "plan_measurement": f"{len(per)} sequential solve_planning_problem() calls per rank after one warm-up call "
                                f"({pp['num_steps']} steps, S={pp['num_samples']} M={pp['num_inducing']} T={pp['time_spacing_X']}, "
                                f"150 posterior paths at {pp['time_spacing_Xnew']} time points, clearance check): "
                                f"{1e3 * tot / len(per):.2f} ms per plan, {solved}/{len(per)} paths collision-free on rank 0; "
                                f"solve_planning_problems_batched on all {len(queries)} industrial queries as one device batch: "
                                f"{1e3 * tb:.1f} ms, {sum(int(r[0]) for r in res)}/{len(queries)} collision-free"}


def plan_quality(rep, solved):
    """Summary of solve_planning_problems_batched's clearance report (metres, signed: distance to the obstacles minus the sphere
    radius, minimum over spheres and time).  `solved` is the strict headless check (best sample clear everywhere, end states
    included); `within_end_states` accepts what the query's own end states already violate."""
    import numpy as np
    st, go, ini, best = (np.asarray(rep[k]) for k in ("start", "goal", "initial_path", "best_sample"))
    free_ends = (st > 0) & (go > 0)
    floor = np.minimum(0.0, np.minimum(st, go))
    r3 = lambda a: [round(float(v), 4) for v in a]
    return {"queries": int(st.size), "end_states_collision_free": int(free_ends.sum()),
            "initial_straight_line_clear": int((ini > 0).sum()),
            "solved": int(np.sum(solved)), "solved_among_free_end_states": int(np.sum(np.asarray(solved) & free_ends)),
            "within_end_states": int((best >= floor - 1e-3).sum()),
            "not_worse_than_initial": int((best >= ini - 1e-3).sum()),
            "clearance_start": r3(st), "clearance_goal": r3(go), "clearance_initial_path": r3(ini), "clearance_best_sample": r3(best),
            "note": "signed clearance by the planner's own sphere model against the mesh-generated SDF; the reference counts a query "
                    "solved when pybullet can drive the arm along the path (utils/robot.py:455-480): link meshes, not inflated spheres. "
                    "A query whose start or goal state has negative clearance by the spheres cannot pass the strict check"}


def timed_region(run_steps, args, dist, backend):
    """W warm-up steps are done by the caller; here: barrier + synchronize, K steps (repeated until --min-seconds have passed,
    every rank the same number of blocks), barrier + synchronize; MAX over ranks.  Returns (seconds per K-step block, blocks)."""
    import numpy as np
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    first = time.perf_counter() - t0
    reps = max(1, int(np.ceil(args.min_seconds / max(first, 1e-6)))) if args.min_seconds > 0 else 1
    dev = "cuda" if backend == "nccl" else "cpu"
    if dist is not None:
        r = torch.tensor([reps], dtype=torch.int64, device=dev)
        dist.broadcast(r, 0)
        reps = int(r[0])
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        run_steps(args.steps)
    barrier()
    elapsed = (time.perf_counter() - t0) / reps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    return elapsed, reps


def run_sample_sharded(args, world, rank, dist, backend):
    """BASELINE config 4: one problem, the sample axis split over the ranks; a step = local forward + reverse, ONE in-place
    all-reduce of the contiguous [gradient | lik | kl] buffer, the replicated Adam update.  Strong scaling in S."""
    import torch
    from vgpmp_amd import capi, sharding
    ps, spec, grid, scene, planner = build_problem(rank, args, world)
    planner.extra_flags |= args.flags
    rccl = world > 1 and backend == "nccl"
    comm = sharding.CapiComm(world, rank) if (args.collective == "capi" and (world == 1 or rccl)) else None
    sp = sharding.SampleShardedPlanner(planner, comm=comm)
    if world == 1 and comm is None:
        sp._allreduce = lambda buf=None: None                 # nothing to exchange on one rank
        sp._single_rank = True                                # ... and the whole loop is one C call (vgpmp_elbo_steps_reduced)
    sp.run_steps(args.warmup)
    elapsed, reps = timed_region(sp.run_steps, args, dist, backend)
    assert args.allow_nan or torch.isfinite(planner.q_mu).all(), "optimisation diverged"
    times = planner.profile_steps(max(1, args.profile_steps))
    S_loc, N, D, P = planner.S, planner.N, spec.dof, spec.num_spheres
    t_sdf = times["loglik_kernel"] * 1e-3
    sdf_bytes = S_loc * N * (28 * P + 8 * D + 4)
    line = None
    if rank == 0:
        how = ("nothing (one rank)" if world == 1 and comm is None else
               f"RCCL via the C ABI (vgpmp_allreduce_grads), {world} rank(s) in the communicator" if comm is not None else
               f"torch.distributed ({backend}{' = RCCL' if backend == 'nccl' else ', host-staged: ranks share a device'}), {world} ranks")
        line = {
            "metric": "ELBO iters/sec, BASELINE config 4: UR10-6DoF industrial S=1024 M=18 T=70, samples sharded over the GPUs",
            "value": args.steps / elapsed, "unit": "ELBO iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "timed_blocks": reps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32; prior products f16-split x2, f32 accumulate (512 samples or more on this rank); covariance path and Adam f64"
                      if S_loc >= 512 and not (planner.extra_flags & capi.PRIOR_F32) else "f32 (f32 MFMA prior products; covariance path and Adam f64)"),
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: UR10 6-DoF, industrial, SDF {'x'.join(str(v) for v in scene.shape)}, ONE problem, "
                                   f"S={args.samples} samples in total ({S_loc} on this rank), M={planner.M} T={N} B={planner.B}",
                       "parallelism": f"samples sharded x{world}; one in-place all-reduce(sum) of {planner.reduce_buf.numel()} float64 per step, "
                                      "then the replicated Adam update",
                       "collective": how,
                       "collective_ranks": world if (comm is not None or world > 1) else 0,
                       "launch": getattr(sp, "schedule", "one vgpmp_elbo_step (forward + reverse) + one vgpmp_adam_step per step")},
            # (the 16 MB table of this scene lives in L2 / Infinity Cache: requested bytes per second for scale, no HBM fraction)
            "roofline": {"kernel": next((k for k in capi.last_schedule(planner.lib) if k.startswith("loglik_")), "likelihood of the local samples"),
                         "bound": "cache (voxel table resident in L2 / Infinity Cache: not an HBM-bound launch)",
                         "achieved": S_loc * N * (16 * P + 8 * D + 4) / t_sdf / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                         "algorithmic_bytes_per_launch": sdf_bytes, "requested_bytes_per_launch": S_loc * N * (16 * P + 8 * D + 4),
                         "avg_launch_ms": times["loglik_kernel"],
                         "by_contract_not_hbm": {"bytes_per_launch": sdf_bytes, "GBps": sdf_bytes / t_sdf / 1e9,
                                                 "ratio_to_hbm_peak": sdf_bytes / t_sdf / 1e9 / HBM_PEAK_GBPS}},
            "stage_ms": {k: round(v, 5) for k, v in times.items()},
        }
    if rank == 0 and world == 1:
        # a prediction for the SCALE run to be held against (no 8-GPU node has ever run this): ONE rank's share of the 8-rank
        # job timed here -- the same planner with S / 8 local samples of the S-sample stream (sample_offset 0, KL on this rank),
        # nothing exchanged -- so the 8-rank step = that + one all-reduce of the gradient buffer (latency-bound, unmeasured)
        import copy
        a8 = copy.copy(args)
        s8, _ = sharding.shard_samples(args.samples, 8, 0)
        from vgpmp_amd import engine
        pp = ps.planner_params
        pl8 = engine.PlannerBatch(scene, __import__("numpy").array([ps.queries[0]]), num_samples=s8, samples_total=args.samples,
                                  sample_offset=0, kl_scale=1.0, num_inducing=args.inducing, num_data=args.timesteps, num_bases=1024,
                                  lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                                  learning_rate=pp["learning_rate"], seed=1234)
        sp8 = sharding.SampleShardedPlanner(pl8)
        sp8._allreduce = lambda buf=None: None
        sp8._single_rank = True
        sp8.run_steps(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n8 = max(args.steps, 200)
        sp8.run_steps(n8)
        torch.cuda.synchronize()
        local_us = 1e6 * (time.perf_counter() - t0) / n8
        t8 = pl8.profile_steps(max(1, args.profile_steps))
        line["projection_8_ranks"] = {
            "rank_local_us_per_step": round(local_us, 2), "samples_per_rank": s8,
            "one_rank_us_per_step": round(1e3 * line["ms_per_step"], 2),
            "stage_ms_one_of_8": {k: round(v, 5) for k, v in t8.items()},
            "predicted_speedup_at_8_ranks_before_the_collective": round(1e3 * line["ms_per_step"] / local_us, 2),
            "collective": f"+ one in-place all-reduce(sum) of {planner.reduce_buf.numel()} float64 per step over RCCL: latency-bound "
                          "(16 KB), never run on more than one rank here -- every microsecond of it lowers the speed-up",
            "what_does_not_shrink": "stage 1 / stage 2's covariance roles, the gradient assembly (mid_hyper_final) and Adam are per "
                                    "problem, not per sample: they are the floor of the step as ranks are added"}
    if world > 1:
        # what the step costs WITHOUT its collective, measured by the same ranks (every rank runs the same local steps -- forward,
        # reverse, Adam, nothing exchanged -- behind the same barriers; MAX over ranks): measured - local = what the all-reduce
        # adds per step.  Held against the N = 1 line's projection_8_ranks, an N-rank run judges itself.  (Last thing on the
        # planner: without the exchange the ranks' variables drift apart.)
        sp_l = sharding.SampleShardedPlanner(planner)
        sp_l._allreduce = lambda buf=None: None
        sp_l._single_rank = True
        sp_l.run_steps(args.warmup)
        a_l = type(args)(**vars(args))
        a_l.min_seconds = min(args.min_seconds, 1.0)
        el_l, _ = timed_region(sp_l.run_steps, a_l, dist, backend)
        if rank == 0:
            local_us = 1e6 * el_l / args.steps
            line["collective_breakdown"] = {
                "ranks": world, "samples_per_rank": S_loc, "measured_us_per_step": round(1e3 * line["ms_per_step"], 2),
                "rank_local_us_per_step": round(local_us, 2),
                "collective_us_per_step": round(1e3 * line["ms_per_step"] - local_us, 2),
                "how": "the same ranks ran the same number of steps without the all-reduce (vgpmp_elbo_steps_reduced, no communicator) "
                       "behind the same barriers; the N = 1 line of this bench carries projection_8_ranks (one rank's share of the "
                       "8-rank job timed alone) to hold rank_local_us_per_step against"}
    if comm is not None:
        comm.close()
    return line


def run_problem_sharded(args, world, rank, dist, backend, want_extras=True):
    """Independent problems per GPU (configs 2, 3, 5): no data-path collective, weak scaling."""
    import numpy as np
    import torch
    from vgpmp_amd import capi
    ps, spec, grid, scene, planner = build_problem(rank, args, world)
    planner.extra_flags |= {"auto": 0, "lanes": capi.LIK_LANES, "lanes-lds": capi.LIK_LDS_STATE}[args.lik_form] | args.flags
    for _ in range(args.warmup):
        planner.step()
    if args.unroll > 0:
        planner.capture(args.unroll)
    torch.cuda.synchronize()

    def plan_block(k):
        # every timed block of K steps starts from the freshly initialised models, as the reference does per start-goal query
        # (a new VGPMP per call of solve_planning_problem, utils/miscellaneous.py:162-169); thousands of steps on one model
        # drive the lengthscales of config 3 (lr 0.09) to ~1e-5 and the factorisation to NaN
        planner.reset()
        planner.run_steps(k)

    elapsed, reps = timed_region(plan_block, args, dist, backend)
    assert args.allow_nan or torch.isfinite(planner.q_mu).all(), "optimisation diverged"
    timed_kernels = capi.last_schedule(planner.lib) if args.unroll == 0 else []      # (include/vgpmp_debug.h: what the library ran)
    pp = ps.planner_params

    # ---- one plan = num_steps optimisation steps + 150 posterior paths + best-sample pick (models/vgpmp.py:312-339)
    t_sample = None
    if args.workload in ("config2", "config3"):
        n_new = int(pp["time_spacing_Xnew"])
        Xnew = np.tile(np.linspace(0.0, 1.0, n_new)[:, None], (1, spec.dof))
        planner.sample_from_posterior(150, Xnew)            # warm (allocations)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        planner.sample_from_posterior(150, Xnew)[1].cpu()
        t_sample = time.perf_counter() - t1

    # ---- per-kernel durations with HIP events (separate pass so the timed region stays clean), over the steps of a plan from
    #      fresh models like a timed block: the SDF pass of a batch is 10-15 % slower in a plan's first steps than in its last
    planner.reset()
    times = planner.profile_steps(max(1, args.profile_steps))
    kernel_ms = {k: times.pop(k) for k in ("loglik_kernel", "prior_gemm_kernel")}      # device start-to-end
    stage_ms = times
    S, N, M, D, P, B = args.samples, args.timesteps, args.inducing, spec.dof, spec.num_spheres, 1024
    npb = args.problems
    sdf_bytes = npb * S * N * (28 * P + 8 * D + 4)                 # 7 fp32 voxels per sphere query
    sdf_bytes16 = npb * S * N * (16 * P + 8 * D + 4)               # what the packed table moves: one 16-byte record per query
    gemm_flops = npb * 2 * (2.0 * S * (N + M + 2) * D * B)         # F0 and H (lengthscales trainable)
    t_sdf, t_gemm = kernel_ms["loglik_kernel"] * 1e-3, kernel_ms["prior_gemm_kernel"] * 1e-3
    # the kernels of the pass the events come from (one launch per kernel), as the library reports them -- not re-derived here
    profiled_kernels = capi.last_schedule(planner.lib)
    lik_kernel = next(k for k in profiled_kernels if k.startswith("loglik_"))
    gemm_kernel = next(k for k in profiled_kernels if k.startswith("prior_"))
    table_bytes = int(scene.table.numel() * 4)
    # a table that fits the 256 MiB Infinity Cache (beside the step's own streams) is served on-die: the launch is then not an HBM
    # launch and is not priced as one (MI355X_MICROARCH.md, "Infinity Cache"; counter traffic << algorithmic bytes confirms it)
    cache_resident = table_bytes <= (256 << 20)
    rate28, rate16 = sdf_bytes / t_sdf / 1e9, sdf_bytes16 / t_sdf / 1e9
    roof_sdf = {"kernel": lik_kernel,
                "bound": "cache (voxel table resident in L2 / Infinity Cache: not an HBM-bound launch)" if cache_resident else "hbm",
                # HBM-resident table: SURVEY 8(d)'s 28-byte contract against the HBM peak.  Cache-resident table: the bytes the
                # launch actually requests (16 per query) per second, for scale only -- no fraction of a peak it never touches
                "achieved": rate16 if cache_resident else rate28,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None if cache_resident else rate28 / HBM_PEAK_GBPS,
                "traffic": None, "algorithmic_bytes_per_launch": sdf_bytes, "requested_bytes_per_launch": sdf_bytes16,
                "avg_launch_ms": kernel_ms["loglik_kernel"],
                "table": {"layout": args.layout, "bytes": table_bytes, "cache_resident": cache_resident,
                          "free_space_summary": bool(scene.free_space_summary), "free_space_masks_in_lds": bool(scene.free_space_mask),
                          "mask_bytes": int(scene.free_mask.numel() * 4) if scene.free_space_mask else 0},
                "timing": "HIP events stamped with the kernel's start and end on its stream (hipExtLaunchKernel), "
                          f"mean of the first {max(1, args.profile_steps)} launches of a plan from fresh models"}
    if cache_resident:
        roof_sdf["by_contract_not_hbm"] = {"bytes_per_launch": sdf_bytes, "GBps": rate28, "ratio_to_hbm_peak": rate28 / HBM_PEAK_GBPS,
                                           "note": "SURVEY 8(d)'s 28 B per query divided by the launch time; the table never leaves "
                                                   "the caches, so this is NOT a fraction of HBM bandwidth in use"}
    else:
        if rate28 <= HBM_ACHIEVABLE_GBPS:
            roof_sdf["frac_of_achievable_6.29TBps"] = rate28 / HBM_ACHIEVABLE_GBPS
        roof_sdf["by_16B_per_query"] = {"bytes_per_launch": sdf_bytes16, "achieved": rate16, "frac": rate16 / HBM_PEAK_GBPS}
    # the kernel that forms the prior draws in the pass the events come from (one launch per kernel, device-drawn noise);
    # same selection as vg_elbo_steps (csrc/gp_path.hip)
    sk = planner.dims.split_k
    peak_gemm = F32_MFMA_PEAK_TFLOPS
    if gemm_kernel.startswith(("prior_fused_split_kernel", "prior_fused_small16_kernel", "prior_split_", "mid_cov_b_prior16")):
        # the library's own W stream is float16 (vgpmp_device.h, "The W stream"): a float32 product = TWO f16 MFMAs (w b_hi + w b_lo,
        # float32 accumulators; gp_prior_split.h:22-27, WX = true in the few-sample form), so the matrix-pipe ceiling for the
        # ALGORITHMIC flops is half the f16 peak.  (Weights injected by a caller are arbitrary float32: three MFMAs -- not timed here.)
        peak_gemm = F16_MFMA_PEAK_TFLOPS / 2.0
    vs_f32 = gemm_flops / t_gemm / 1e12 / F32_MFMA_PEAK_TFLOPS
    if gemm_kernel.startswith(("prior_fused_split_kernel", "prior_split_")):
        gemm_note = "W + features formed inside the GEMM; f16-split x2: peak = f16 MFMA peak / 2 (NOT an f32-MFMA kernel; see profiles/README.md)"
    elif gemm_kernel.startswith(("prior_fused_small16_kernel", "mid_cov_b_prior16")):
        gemm_note = ("few samples, register-resident f16-split x2 products, peak = f16 MFMA peak / 2; 16-row tiles hold %d samples "
                     "(%.0f %% of the tile flops are algorithmic)" % (S, 100.0 * S / (16 * ((S + 15) // 16))))
    elif gemm_kernel.startswith(("prior_fused_batch_kernel", "prior_fused_small_kernel")):
        gemm_note = "W / the features formed inside the GEMM, float32 MFMAs (VGPMP_PRIOR_F32)"
    else:
        gemm_note = "a role of stage2_kernel in the timed schedule; timed alone here"
    roof_gemm = {"kernel": gemm_kernel, "note": gemm_note, "bound": "mfma", "achieved": gemm_flops / t_gemm / 1e12,
                 "peak": peak_gemm, "unit": "TFLOP/s", "frac": gemm_flops / t_gemm / 1e12 / peak_gemm,
                 "traffic": None, "algorithmic_flops_per_launch": gemm_flops, "avg_launch_ms": kernel_ms["prior_gemm_kernel"],
                 "ratio_to_f32_mfma_peak_157TF": vs_f32}
    dominant = max(stage_ms, key=stage_ms.get)
    # counter traffic of THIS round's collection, per workload (tools/run_collect.sh copies the tables to these names)
    own = {("config2", 1): "pmc_traffic.json", ("stress", 64): "pmc_traffic_config5.json", ("config3", 55): "pmc_traffic_config3.json",
           ("config2", 64): "pmc_traffic_franka64.json"}
    tfile = os.path.join(ROOT, "profiles", own.get((args.workload, npb), "none"))
    if args.traffic_file:
        tfile = args.traffic_file
    if os.path.exists(tfile) and (args.traffic_file or ((args.workload, npb) in own and args.scene == "mesh" and not args.also_train)):
        try:
            t = json.load(open(tfile))
            roof_sdf["traffic"] = t.get(lik_kernel, {}).get("hbm_bytes_per_launch")
            roof_sdf["traffic_source"] = t.get("source")
            roof_sdf["traffic_collected_at"] = t.get("collected_at")
            roof_gemm["traffic"] = t.get(gemm_kernel, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass

    names = {"config2": "BASELINE config 2: Franka 7-DoF, industrial",
             "config3": "BASELINE config 3: Franka 7-DoF, bookshelves, all C(11,2)=55 pairs",
             "stress": "BASELINE config 5: synthetic 14-DoF arm" + (" (per-GPU share of the 512-problem batch)" if npb * world <= 512 and npb < 512 else "")}
    line = {
        "metric": METRIC,
        "value": world * npb * args.steps / elapsed, "unit": "ELBO iters/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "timed_blocks": reps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32; prior products f16-split x2, f32 accumulate (v_mfma_f32_16x16x32_f16); covariance path and Adam f64"
                  # (batches beyond the few-problem schedule: the f16-split kernels -- large-batch or few-sample form)
                  if (not (planner.extra_flags & capi.PRIOR_F32) and not (planner.fuse and npb * D <= (64 if S <= 32 else 32))
                      and (sk == 1 or S <= 32))
                  else "f32 (f32 MFMA prior products; covariance path and Adam f64)"),
        "data": "synthetic",
        "config": {"workload": names[args.workload] + ", SDF " + "x".join(str(v) for v in scene.shape)
                               + (" (from the scene's collision mesh)" if args.workload != "stress" and args.scene != "synthetic"
                                  else " (synthetic boxes/spheres)")
                               + f", {npb} problem(s)/GPU, S={S} M={M} T={N} B={B}"
                               + (", also trained: " + args.also_train if args.also_train else ""),
                   "parallelism": f"problems sharded x{world}, no collective"
                                  + ("; ONE problem per GPU here: the multi-GPU figure is summary.batch_512 (64 problems per GPU)"
                                     if world > 1 and npb == 1 else ""),
                   "launch": (f"hipGraph x{args.unroll} steps" if args.unroll else "plain launches")
                             + ("; few-problem schedule: the step's kernels share 4-5 launches (timed_schedule_kernels)"
                                if planner.fuse and npb * D <= (64 if S <= 32 else 32) else "; batch schedule (timed_schedule_kernels)")},
        "plans_per_sec": (world * npb / (float(pp["num_steps"]) * elapsed / args.steps + t_sample)
                          if t_sample is not None else None),
        "plan_definition": (f"computed: {pp['num_steps']} optimisation steps at the timed rate + 150 posterior paths at "
                            f"{pp['time_spacing_Xnew']} time points + best-sample pick (sampling measured: {1e3 * t_sample:.2f} ms "
                            f"for the batch); plans_per_sec_measured is the timed solve_planning_problem() figure")
                           if t_sample is not None else None,
        "roofline": roof_sdf, "roofline_secondary": roof_gemm,
        "timed_schedule_kernels": timed_kernels,
        "dominant_stage": dominant, "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
        "stage_ms_schedule": "separate pass after the timed region with ONE launch per kernel and a HIP event around every stage "
                             "(vgpmp_elbo_step_profiled); at 4 problems or fewer the timed region runs the shared stage launches "
                             "instead, so stage_ms / dominant_stage describe that pass, not the timed schedule; the two "
                             "roofline kernels are timed by their own start / end events in the same pass",
    }
    del planner, scene
    torch.cuda.empty_cache()
    if want_extras and args.workload == "config2" and not args.no_solve:
        line.update(measured_solves(args, world, rank, dist, backend))
    if rank == 0:
        line["_ctx"] = (ps, spec, grid, args)          # main() times the CPU baseline last, after every GPU figure
    return line if rank == 0 else None


SUB_KEYS = ("value", "unit", "ms_per_step", "steps", "warmup", "timed_blocks", "scaling", "dtype", "config", "roofline",
            "roofline_secondary", "timed_schedule_kernels", "dominant_stage", "stage_ms")


def sub_record(argv, world, rank, dist, backend):
    """A further workload timed by the same ranks behind the same barriers, after (and outside) the line's own timed region;
    compact form of its line."""
    a = resolve(parse_args(list(argv) + ["--gpus", str(world)]))
    l2 = run_problem_sharded(a, world, rank, dist, backend, want_extras=False)
    if rank != 0:
        return None
    l2.pop("_ctx", None)
    rec = {k: l2[k] for k in SUB_KEYS}
    rec["problems_total"] = world * a.problems
    return rec


def summary_of(line):
    """<= 800 bytes: per (sub-)record ms per step, the SDF kernel's fraction of the HBM peak (null where its table is cache
    resident), the prior kernel's fraction of its matrix peak."""
    r3 = lambda v: None if v is None else round(float(v), 4)
    out = {}
    for name in ("batch_512", "batch_512_one_gpu", "config3", "batch_64"):
        rec = line.get(name)
        if rec:
            out[name] = {"ms_per_step": r3(rec["ms_per_step"]), "sdf_frac_hbm": r3(rec["roofline"]["frac"]),
                         "prior_frac": r3(rec["roofline_secondary"]["frac"]), "problems_total": rec.get("problems_total")}
    if out.get("batch_512") and out.get("batch_512_one_gpu"):
        # (at N = 1 this is 8 -- 64 against 512 problems on the same device -- only when the step is perfectly linear in the batch)
        out["batch_512_one_gpu"]["over_64_problem_share"] = r3(out["batch_512_one_gpu"]["ms_per_step"] / out["batch_512"]["ms_per_step"])
    out["line"] = {"ms_per_step": r3(line["ms_per_step"]), "value": round(float(line["value"]), 1), "n_gpus": line["n_gpus"],
                   "sdf_frac_hbm": r3(line["roofline"]["frac"])}
    if line.get("collective_breakdown"):
        c8 = line["collective_breakdown"]
        out["collective"] = {k: c8[k] for k in ("ranks", "rank_local_us_per_step", "collective_us_per_step")}
    if line.get("projection_8_ranks"):
        p8 = line["projection_8_ranks"]
        out["projection_8_ranks"] = {"rank_local_us_per_step": p8["rank_local_us_per_step"],
                                     "speedup_before_collective": p8["predicted_speedup_at_8_ranks_before_the_collective"]}
    cb = line.get("cpu_baseline") or {}
    if cb.get("openmp"):
        out["cpu_openmp"] = {"value": r3(cb["openmp"].get("value")), "threads": cb["openmp"].get("threads")}
    return out


LINE_LIMIT = 6000          # bytes of the ONE stdout line (the r05 line had grown to 20.7 KB and the driver's parser gave up on it)
STR_LIMIT = 200            # ... and no string value in it longer than this
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _num(v, nd=6):
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float(f"{float(v):.{nd}g}")
    except Exception:
        return v


def _pick(d, keys, nd=6):
    return {k: _num(d[k], nd) for k in keys if d is not None and k in d}


def _short(s, n=STR_LIMIT):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_roofline(r):
    """Numbers + kernel + bound; the prose (how the counters were collected and corrected) is profiles/README.md."""
    if not r:
        return r
    out = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch",
                    "requested_bytes_per_launch", "algorithmic_flops_per_launch", "traffic_collected_at", "ratio_to_f32_mfma_peak_157TF"))
    out["bound"] = _short(out.get("bound", ""), 40).split(" (")[0]
    out["kernel"] = _short(out.get("kernel", ""), 80)
    if "table" in r:
        out["table"] = _pick(r["table"], ("layout", "bytes", "cache_resident", "free_space_masks_in_lds"))
    if "by_contract_not_hbm" in r:
        out["by_contract_not_hbm"] = _pick(r["by_contract_not_hbm"], ("GBps", "ratio_to_hbm_peak"))
    if "by_16B_per_query" in r:
        out["by_16B_per_query"] = _pick(r["by_16B_per_query"], ("achieved", "frac"))
    out["notes"] = "profiles/README.md"
    return out


def compact_cpu_baseline(cb):
    if not cb:
        return cb
    out = _pick(cb, ("value", "unit", "cores", "kind", "logical_cores"))
    out["sample"] = _short(cb.get("sample", ""), 160)
    if cb.get("single_thread"):
        out["single_thread"] = _pick(cb["single_thread"], ("value",))
    om = cb.get("openmp")
    if om:
        out["openmp"] = _pick(om, ("value", "threads", "kind", "error"))
        if om.get("single_thread"):
            out["openmp"]["single_thread"] = _num(om["single_thread"].get("value"))
        if om.get("problem_parallel"):
            out["openmp"]["problem_parallel"] = _num(om["problem_parallel"].get("value"))
    if cb.get("problem_parallel"):
        out["problem_parallel"] = _pick(cb["problem_parallel"], ("value", "cores", "error"))
    return out


def compact_line(line):
    """The ONE stdout line: <= LINE_LIMIT bytes, no string longer than STR_LIMIT.  Everything else (the sub-records' full rooflines,
    plan-quality arrays, the oracle plan check, stage tables, the sentences) is bench_detail.json beside this script."""
    out = {k: _num(line.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_blocks",
                                          "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["dtype"] = _short(out["dtype"], 120)
    out["config"] = {k: _short(v) for k, v in line["config"].items() if isinstance(v, str)}
    out["config"].update({k: v for k, v in line["config"].items() if isinstance(v, (int, float))})
    out["roofline"] = compact_roofline(line.get("roofline"))
    if line.get("roofline_secondary"):
        out["roofline_secondary"] = compact_roofline(line["roofline_secondary"])
    if line.get("cpu_baseline"):
        out["cpu_baseline"] = compact_cpu_baseline(line["cpu_baseline"])
    for k in ("gpu_over_cpu", "gpu_over_cpu_openmp", "gpu_batch_over_cpu_openmp_problem_parallel", "gpu_batch_over_cpu_problem_parallel",
              "plans_per_sec", "plans_per_sec_measured", "plans_per_sec_batched_measured"):
        if line.get(k) is not None:
            out[k] = _num(line[k], 5)
    if line.get("timed_schedule_kernels"):
        out["timed_schedule_kernels"] = [_short(k, 60) for k in line["timed_schedule_kernels"]][:8]
    if line.get("stage_ms"):
        out["stage_ms"] = {k: _num(v, 4) for k, v in line["stage_ms"].items()}
    for k in ("projection_8_ranks", "collective_breakdown"):
        if line.get(k):
            out[k] = {a: b for a, b in line[k].items() if not isinstance(b, (str, dict))}
    out["detail"] = "bench_detail.json"
    out["summary"] = line["summary"]          # LAST key: a driver that keeps the tail of the line keeps this
    return out


def emit(line):
    """Full record -> bench_detail.json (and gpurun_out/ when it exists); compact record -> the one stdout line."""
    text = json.dumps(line)
    for path in (DETAIL_FILE, os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path + ".tmp", "w") as f:
                    f.write(text + "\n")
                os.replace(path + ".tmp", path)
        except OSError as e:                                     # (a read-only tree: the line itself still goes out)
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    out = json.dumps(compact_line(line))
    assert len(out) <= LINE_LIMIT, f"bench line grew to {len(out)} bytes (> {LINE_LIMIT}): trim compact_line()"
    print(out, flush=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        sys.exit(cpu_worker_main(sys.argv[2], int(sys.argv[3])))
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no external launcher: start the ranks as fresh child interpreters BEFORE this process touches the GPU
        # (torch.cuda.device_count() does not initialise it) and relay rank 0's line
        import torch
        from vgpmp_amd import launch
        assert not torch.cuda.is_initialized(), "the rank launcher must run before this process touches the GPU"
        sys.exit(launch.spawn_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__),
                                    devices=torch.cuda.device_count()))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    default_workload = args.workload == "config2" and args.shard == "problems" and not args.problems
    # the problem-parallel CPU baseline: its worker processes start NOW, before this process touches the GPU, and sleep
    pool = None
    if world == 1 and not args.no_cpu_baseline and args.shard == "problems" and args.workload == "config2" and args.scene == "mesh":
        pool = CpuPool(max(1, min(usable_cores(), 64)))
    import torch
    dist = None
    backend = os.environ.get("VGPMP_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path on a 1-GPU box
    if world > 1 and "VGPMP_DIST_BACKEND" not in os.environ and torch.cuda.device_count() < world:
        backend = "gloo"      # (an external launcher on a box with fewer devices than ranks: RCCL refuses two ranks on one device)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        import torch.distributed as dist
        dev = local % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        # ranks started by vgpmp_amd/launch.py meet through a file store (no port to lose to a parallel job); under an external
        # launcher (torch.distributed.run) the usual env:// rendezvous on MASTER_ADDR:MASTER_PORT
        rdzv = dict(init_method=os.environ["VGPMP_INIT_METHOD"], rank=rank, world_size=world) if os.environ.get("VGPMP_INIT_METHOD") else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), **rdzv)
        else:
            dist.init_process_group(backend, **rdzv)
    else:
        torch.cuda.set_device(0)
    resolve(args)
    try:
        if args.shard == "samples":
            line = run_sample_sharded(args, world, rank, dist, backend)
        else:
            line = run_problem_sharded(args, world, rank, dist, backend)
            ctx = line.pop("_ctx", None) if rank == 0 else None
            quick = ["--no-cpu-baseline", "--no-solve", "--warmup", "3", "--min-seconds", "0.5"]      # (+ --profile-steps = a whole plan)
            auto = default_workload and not (args.no_solve and args.no_cpu_baseline)      # (measurement runs of the line alone pass both)
            if args.also_stress == "on" or (args.also_stress == "auto" and auto):
                # the batch regime in front of the driver: the 512-problem batch of the north star, this GPU's 64 (BASELINE
                # config 5 share), whole plans of 200 steps from fresh models
                rec = sub_record(["--workload", "stress", "--steps", "200", "--profile-steps", "200"] + quick, world, rank, dist, backend)
                if rank == 0:
                    line["batch_512"] = rec
                if world == 1:
                    # ... and the whole 512-problem batch on ONE device: the denominator of the north star's ">= 6x at 8 GPUs on a
                    # 512-problem batch" (an N = 8 line's batch_512 is the same 512 problems, 64 per GPU: strong scaling = this / that)
                    rec = sub_record(["--workload", "stress", "--problems", "512", "--steps", "200", "--profile-steps", "40",
                                      "--no-cpu-baseline", "--no-solve", "--warmup", "3", "--min-seconds", "2.0"], world, rank, dist, backend)
                    if rank == 0:
                        line["batch_512_one_gpu"] = rec
            if args.also_config3 == "on" or (args.also_config3 == "auto" and auto and world == 1):
                # BASELINE config 3, the reference's literal benchmark workload: 55 Franka / bookshelves pairs, S=7, 130 steps
                rec = sub_record(["--workload", "config3", "--steps", "130", "--profile-steps", "130"] + quick, world, rank, dist, backend)
                if rank == 0:
                    line["config3"] = rec
                # 64 problems of the line's own shape (config 2) as one batch: the GPU figure the problem-parallel CPU baseline
                # is to be held against
                rec = sub_record(["--problems", "64", "--steps", "200", "--profile-steps", "200"] + quick, world, rank, dist, backend)
                if rank == 0:
                    line["batch_64"] = rec
            if rank == 0 and ctx is not None and world == 1 and not args.no_cpu_baseline:
                torch.cuda.synchronize()
                line["cpu_baseline"] = cpu_baseline(*ctx, pool=pool)
                pool = None
                if default_workload and not args.no_solve:
                    line["cpu_baseline"]["oracle_plan_check"] = oracle_plan_check()
                line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
                omp_ = line["cpu_baseline"].get("openmp") or {}
                if omp_.get("value"):      # the compiled restatement on all usable cores: the CPU figure to hold the GPU's against
                    line["gpu_over_cpu_openmp"] = line["value"] / omp_["value"]
                    if (omp_.get("problem_parallel") or {}).get("value") and "batch_64" in line:
                        line["gpu_batch_over_cpu_openmp_problem_parallel"] = line["batch_64"]["value"] / omp_["problem_parallel"]["value"]
                pp_ = line["cpu_baseline"].get("problem_parallel")
                if pp_ and pp_.get("value") and "batch_64" in line:
                    line["gpu_batch_over_cpu_problem_parallel"] = line["batch_64"]["value"] / pp_["value"]
    finally:
        if pool is not None:
            pool.close()
    if rank == 0:
        line["summary"] = summary_of(line)
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
