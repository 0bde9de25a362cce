"""Processes that ARRIVE on and LEAVE the GPU while a test of the pytest process is running (tests/test_gpu_attach.py).

    python tests/attach_worker.py wait <dir>     # started by tests/conftest.py at session start, before the GPU is touched

The waiting launcher never touches the GPU itself; whenever a test writes <dir>/go_attach_<k> (k = 1, 2, ...) it starts, side by side, the process mix beside
which the likelihood kernels of round 5 returned wrong gradients in 7 of 8 sessions (tools/flake_session, profiles/r06/flake.md): the
two rank processes of the sample-sharding test (tests/shard_worker.py, gloo) and `bench.py --gpus 2 --shard samples` (which starts two
more ranks), then two plain visitors (`attach_worker.py visit`: open the device, build a planner, twenty steps, exit).  What mattered about that mix
turned out to be the bench's KERNELS (this library's prior draws: wide f16 matrix instructions) beside the test's, not the arrivals.  When all have
exited it writes <dir>/done_attach_<k> with their exit codes.  <dir>/go_mfma_<j> starts ONE process instead that runs nothing but f16 matrix
instructions (`attach_worker.py mfma`: vgpmp_debug_mfma_load) until <dir>/stop_mfma_<j> appears -- the neighbour beside which the round-5
library was wrong at every step (profiles/r06/flake.md, "What triggers it")."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def visit() -> int:
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from vgpmp_amd import engine, robots, scenes
    torch.cuda.set_device(0)
    ps = robots.load_problemset("franka", "industrial")
    spec = robots.load_robot("franka", *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    pl = engine.PlannerBatch(sc, np.array([ps.queries[0], ps.queries[1]]), num_samples=64, num_inducing=12, num_data=40,
                             num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=3)
    pl.run_steps(20)
    torch.cuda.synchronize()
    return 0 if bool(torch.isfinite(pl.q_mu).all()) else 1


def mix(out: str, tag: str, log) -> list:
    """The reproducer's process mix, once: two gloo rank processes + a two-rank bench side by side, then two pairs of plain visitors."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sub = os.path.join(out, "attach_ranks_" + tag)
    os.makedirs(sub, exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    py = sys.executable
    procs = [subprocess.Popen([py, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2", str(port), sub], env=env, stdout=log, stderr=log)
             for r in range(2)]
    procs.append(subprocess.Popen([py, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard", "samples", "--steps", "5", "--warmup", "2",
                                   "--min-seconds", "0", "--profile-steps", "1"], env=env, stdout=subprocess.DEVNULL, stderr=log))
    rcs = [p.wait() for p in procs]
    for _ in range(2):
        p = subprocess.Popen([py, os.path.abspath(__file__), "visit"], env=env, stdout=log, stderr=log)
        time.sleep(0.7)
        q = subprocess.Popen([py, os.path.abspath(__file__), "visit"], env=env, stdout=log, stderr=log)
        rcs += [p.wait(), q.wait()]
    return rcs


def mfma(out: str, tag: str, seconds: float) -> int:
    """A process that does nothing but f16 matrix instructions (vgpmp_debug_mfma_load, include/vgpmp_debug.h) until <out>/stop_mfma_<tag>
    appears or `seconds` have passed; <out>/ready_mfma_<tag> says when its kernels are running."""
    sys.path.insert(0, ROOT)
    import torch
    from vgpmp_amd import capi
    lib = capi.load(require=True)
    torch.cuda.set_device(0)
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    stream = int(torch.cuda.current_stream().cuda_stream)
    # the matrix kernels in bursts on a second stream, a small vector kernel on the first one synchronised every round: kernels of
    # this process start and end all the time, as those of tools/pk_probe.hip's aggressor mode do (the pattern beside which the
    # round-5 ELBO step was wrong at every step; a single stream of back-to-back matrix kernels left it alone)
    second = torch.cuda.Stream()
    x = torch.linspace(0.0, 1.0, 1 << 16, device="cuda")
    t0, ready, n = time.time(), False, 0
    while time.time() - t0 < seconds and not os.path.exists(os.path.join(out, f"stop_mfma_{tag}")):
        capi.check(lib.vgpmp_debug_mfma_load(capi.ptr(sink), 1024, 2000, int(second.cuda_stream)), "vgpmp_debug_mfma_load")
        x = torch.sin(x) * 0.999 + 1e-3
        torch.cuda.current_stream().synchronize()
        n += 1
        if n % 8 == 0:
            second.synchronize()
        if not ready and n >= 8:
            open(os.path.join(out, f"ready_mfma_{tag}"), "w").close()
            ready = True
    torch.cuda.synchronize()
    return 0


def wait(out: str) -> int:
    """Serves the trigger files <out>/go_attach_1, go_attach_2, ... in turn (one mix each, answered by done_attach_<k>) until the parent goes."""
    parent = os.getppid()
    log = open(os.path.join(out, "attach_visitors.log"), "w")
    k = j = 1
    while True:
        go, go_m = os.path.join(out, f"go_attach_{k}"), os.path.join(out, f"go_mfma_{j}")
        while not os.path.exists(go) and not os.path.exists(go_m):
            if os.getppid() != parent:
                return 0
            time.sleep(0.05)
        if os.path.exists(go_m):      # <out>/go_mfma_<j>: ONE process that runs f16 matrix instructions until <out>/stop_mfma_<j> (at most 120 s)
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
            rc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "mfma", out, str(j), "120"], env=env, stdout=log, stderr=log).wait()
            log.flush()
            with open(os.path.join(out, f"done_mfma_{j}.tmp"), "w") as f:
                f.write(str(rc))
            os.replace(os.path.join(out, f"done_mfma_{j}.tmp"), os.path.join(out, f"done_mfma_{j}"))
            j += 1
            continue
        rcs = mix(out, str(k), log)
        log.flush()
        with open(os.path.join(out, f"done_attach_{k}.tmp"), "w") as f:
            f.write(" ".join(str(r) for r in rcs))
        os.replace(os.path.join(out, f"done_attach_{k}.tmp"), os.path.join(out, f"done_attach_{k}"))
        k += 1


if __name__ == "__main__":
    if sys.argv[1] == "visit":
        sys.exit(visit())
    if sys.argv[1] == "mfma":
        sys.exit(mfma(sys.argv[2], sys.argv[3], float(sys.argv[4])))
    sys.exit(wait(sys.argv[2]))
