"""One process per GPU without an external launcher.

`spawn_ranks(n, argv)` starts `n` FRESH child interpreters of the same script (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in their environment, rendezvous on 127.0.0.1), relays rank 0's standard output and returns
the first non-zero exit code (0 when every rank succeeded).  It must be called BEFORE the calling process touches the
GPU: the children are new programs, and a process that has initialised HIP must not start others by exec (the parent
here only forks children that exec immediately; it never replaces itself).  `torch.cuda.device_count()` does not
initialise the GPU on this image, so the parent may use it to decide the backend: with fewer devices than ranks (a 1-GPU
box) the ranks share devices and rendezvous over gloo -- RCCL refuses two ranks on one device -- which rehearses the
N > 1 path; with enough devices the backend stays "nccl" (= RCCL over xGMI).

Reference role: the outer loop of benchmarking.py:68-85 (one process there); here its problems / samples are sharded.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None, devices: Optional[int] = None,
             store: Optional[str] = None) -> Dict[str, str]:
    """Environment of one rank.  `store`: path of a file-store rendezvous (VGPMP_INIT_METHOD = file://...), which scripts
    that know about it (bench.py) prefer to MASTER_PORT: free_port() closes its socket before the ranks bind it, so a job
    started beside this one could take the port in between; a fresh file cannot be taken."""
    env = dict(os.environ if base is None else base)
    if store:
        env["VGPMP_INIT_METHOD"] = "file://" + store
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC unless the user chose otherwise
    if devices is not None and devices < world and "VGPMP_DIST_BACKEND" not in env:
        env["VGPMP_DIST_BACKEND"] = "gloo"          # ranks share devices: RCCL cannot, gloo rehearses the path
    return env


def spawn_ranks(world: int, argv: Sequence[str], script: Optional[str] = None, devices: Optional[int] = None,
                timeout_s: float = 3600.0, poll_s: float = 0.2) -> int:
    """Run `script argv...` as `world` ranks; rank 0 inherits stdout, the others' stdout goes to stderr."""
    script = script or os.path.abspath(sys.argv[0])
    port = free_port()
    store_dir = tempfile.mkdtemp(prefix="vgpmp_rdzv_")
    store = os.path.join(store_dir, "store")
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=rank_env(r, world, port, devices=devices, store=store),
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or all(c is not None for c in codes) or time.monotonic() > deadline:
            if bad:
                rc = bad[0]
            elif any(c is None for c in codes):
                rc = 124                                # timed out
            break
        time.sleep(poll_s)
    for p in procs:                                      # a failed / hung job: end exactly the children started here
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    try:
        if os.path.exists(store):
            os.unlink(store)
        os.rmdir(store_dir)
    except OSError:
        pass
    return rc
