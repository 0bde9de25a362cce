"""Processes that ARRIVE on and LEAVE the GPU while a test of the pytest process is running (tests/test_gpu_attach.py).

    python tests/attach_worker.py wait <dir> <visits> <at a time>     # started by tests/conftest.py at session start
    python tests/attach_worker.py visit                               # one visitor: open the device, run a little, leave

The waiting launcher is started BEFORE the pytest process touches the GPU and never touches it itself; when the test
writes <dir>/go_attach it starts `visits` visitors, `at a time` of them side by side, and writes <dir>/done_attach.  A
visitor is a fresh interpreter that loads the library, builds a scene and a planner, runs a few optimisation steps and
exits: its arrival and its exit are the moments at which the hardware scheduler rebuilds its run list and every queue
of the device is preempted and resumed -- the condition under which the likelihood kernels of round 5 returned wrong
gradients (profiles/r06/flake.md)."""
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


def wait(out: str, visits: int, at_a_time: int) -> int:
    parent = os.getppid()
    go = os.path.join(out, "go_attach")
    while not os.path.exists(go):
        if os.getppid() != parent:
            return 0
        time.sleep(0.05)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    log = open(os.path.join(out, "attach_visitors.log"), "w")
    running, started, rcs = [], 0, []
    while started < visits or running:
        while started < visits and len(running) < at_a_time:
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "visit"], env=env, stdout=log, stderr=log))
            started += 1
            time.sleep(0.7)                    # staggered: arrivals and exits at different moments
        for p in [p for p in running if p.poll() is not None]:
            rcs.append(p.returncode)
            running.remove(p)
        time.sleep(0.05)
    log.close()
    with open(os.path.join(out, "done_attach.tmp"), "w") as f:
        f.write(" ".join(str(r) for r in rcs))
    os.replace(os.path.join(out, "done_attach.tmp"), os.path.join(out, "done_attach"))
    return 0


if __name__ == "__main__":
    if sys.argv[1] == "visit":
        sys.exit(visit())
    sys.exit(wait(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])))
