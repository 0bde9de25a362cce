"""Processes that ARRIVE on and LEAVE the GPU while a test of the pytest process is running (tests/test_gpu_attach.py).

    python tests/attach_worker.py wait <dir>     # started by tests/conftest.py at session start, before the GPU is touched

The waiting launcher never touches the GPU itself; whenever a test writes <dir>/go_attach_<k> (k = 1, 2, ...) it starts, side by side, the process mix beside
which the likelihood kernels of round 5 returned wrong gradients in 7 of 8 sessions (tools/flake_session, profiles/r06/flake.md): the
two rank processes of the sample-sharding test (tests/shard_worker.py, gloo) and `bench.py --gpus 2 --shard samples` (which starts two
more ranks), then two plain visitors (`attach_worker.py visit`: open the device, build a planner, twenty steps, exit).  Every arrival
and every exit makes the hardware scheduler rebuild its run list: each queue of the device is preempted and resumed.  When all have
exited it writes <dir>/done_attach_<k> with their exit codes."""
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


def wait(out: str) -> int:
    """Serves the trigger files <out>/go_attach_1, go_attach_2, ... in turn (one mix each, answered by done_attach_<k>) until the parent goes."""
    parent = os.getppid()
    log = open(os.path.join(out, "attach_visitors.log"), "w")
    k = 1
    while True:
        go = os.path.join(out, f"go_attach_{k}")
        while not os.path.exists(go):
            if os.getppid() != parent:
                return 0
            time.sleep(0.05)
        rcs = mix(out, str(k), log)
        log.flush()
        with open(os.path.join(out, f"done_attach_{k}.tmp"), "w") as f:
            f.write(" ".join(str(r) for r in rcs))
        os.replace(os.path.join(out, f"done_attach_{k}.tmp"), os.path.join(out, f"done_attach_{k}"))
        k += 1


if __name__ == "__main__":
    if sys.argv[1] == "visit":
        sys.exit(visit())
    sys.exit(wait(sys.argv[2]))
