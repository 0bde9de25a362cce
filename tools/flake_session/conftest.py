"""Measurement aid (NOT part of the test suite: `python -m pytest tools/flake_session -q -s`): a session that starts the GPU test session's rank
processes beside its first test WITHOUT waiting for them -- the context in which bit-for-bit comparisons differed in ~5 % of sessions
(DESIGN section 4)."""
import os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_sessionstart(session):
    if os.environ.get("FLAKE_NO_SPAWN"):      # (the tests alone, e.g. beside a process that only RUNS kernels)
        return
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tempfile.mkdtemp(prefix="vgpmp_flake_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    session.config._procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2", str(port), out],
                                              env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(2)]
    session.config._procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard", "samples", "--steps", "5",
                                                   "--warmup", "2", "--min-seconds", "0", "--profile-steps", "1"], env=env,
                                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))


def pytest_sessionfinish(session, exitstatus):
    for p in getattr(session.config, "_procs", []):
        if p.poll() is None:
            p.kill()
