#!/usr/bin/env python3
"""Measurement aid (profiles/r06/flake.md, "What triggers it"): WHICH neighbour makes the defect show?

    python tools/trigger_probe.py <kind> [seconds = 10] [probe = pk_probe_010]

Runs tools/<probe> (plain HIP: one packed-FP32 instruction form checked against its definition, tools/pk_probe.hip) for <seconds> and,
beside it, visitors of ONE kind, again and again until the probe ends; prints the probe's verdict.  Kinds:

    none       nobody
    hip        8 plain-HIP processes at a time on 4 streams each (tools/pk_probe_dflt: kernels only)
    hip_host   one plain-HIP process allocating and freeing pinned host memory (PK_CHURN=host)
    hip_vram   one plain-HIP process allocating and freeing 1 GiB of device memory (PK_CHURN=vram)
    torchinit  python: import torch, one tensor on the device, exit
    visit      tests/attach_worker.py visit (torch + this library: a planner, twenty steps)
    pinned     python: torch pinned host tensors allocated, copied to the device, dropped
    gloo2      two python ranks: torch.distributed gloo, a CPU all-reduce, a device kernel each
    shard      the two gloo rank processes of tests/shard_worker.py
    bench2     bench.py --gpus 2 --shard samples (two ranks sharing the device)
    bench1s / bench1 / bench2p   one bench process on config 4 / on the default line / two ranks with problem sharding
    x:<mode>   another plain-HIP process keeps ONE kind of kernel running (agg_kernel of tools/pk_probe.hip: 1 float64 FMA, 2 f16 MFMA,
               3 float64 MFMA, 4 LDS, 5 sin / cos, 6 global memory, 7 float32 MFMA, 8 float32 FMA)
    in:<mode>  the same kernel on a second stream of the probe's OWN process, nobody else on the device
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sys.executable
ENV = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

GLOO_RANK = r"""
import os, sys, torch, torch.distributed as dist
r, port = int(sys.argv[1]), sys.argv[2]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=r, world_size=2)
x = torch.ones(1 << 20, device="cuda:0")
for i in range(20):
    y = (x * 2).sum().cpu().reshape(1)
    dist.all_reduce(y)
torch.cuda.synchronize(); dist.destroy_process_group()
"""
PINNED = r"""
import time, torch
t0 = time.time()
while time.time() - t0 < 1.5:
    h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    d = h.to("cuda:0", non_blocking=True); torch.cuda.synchronize(); del h, d
"""


def free_port() -> str:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return str(p)


def one_round(kind: str, tmp: str):
    q = dict(stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    tools = os.path.join(ROOT, "tools")
    if kind == "none" or kind.startswith("in:"):
        time.sleep(0.5); return []
    if kind.startswith("x:"):      # another PROCESS keeps one kind of kernel running (tools/pk_probe.hip, agg_kernel)
        return [subprocess.Popen([os.path.join(tools, "pk_probe_dflt"), "3", "400", "64"], env=dict(ENV, PK_AGG=kind[2:]), **q)]
    if kind == "hip":
        return [subprocess.Popen([os.path.join(tools, "pk_probe_dflt"), "1.5", "400", "512"], env=dict(ENV, PK_STREAMS="4"), **q) for _ in range(8)]
    if kind in ("hip_host", "hip_vram"):
        return [subprocess.Popen([os.path.join(tools, "pk_probe_dflt"), "1.5", "400", "512"], env=dict(ENV, PK_CHURN=kind[4:]), **q)]
    if kind == "torchinit":
        return [subprocess.Popen([PY, "-c", "import torch; x = torch.zeros(1, device='cuda:0'); torch.cuda.synchronize()"], env=ENV, **q)]
    if kind == "visit":
        return [subprocess.Popen([PY, os.path.join(ROOT, "tests", "attach_worker.py"), "visit"], env=ENV, **q)]
    if kind == "pinned":
        return [subprocess.Popen([PY, "-c", PINNED], env=ENV, **q)]
    if kind == "gloo2":
        port = free_port()
        return [subprocess.Popen([PY, "-c", GLOO_RANK, str(r), port], env=ENV, **q) for r in range(2)]
    if kind == "shard":
        port = free_port()
        sub = os.path.join(tmp, "ranks_%d" % int(time.time() * 1000)); os.makedirs(sub, exist_ok=True)
        return [subprocess.Popen([PY, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2", port, sub], env=ENV, **q) for r in range(2)]
    if kind == "bench2":
        return [subprocess.Popen([PY, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard", "samples", "--steps", "5", "--warmup", "2", "--min-seconds", "0",
                                  "--profile-steps", "1"], env=ENV, **q)]
    b = [PY, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--min-seconds", "0", "--profile-steps", "1"]
    if kind == "bench1s":      # ONE process: config 4 whole (no second rank, no collective)
        return [subprocess.Popen(b + ["--gpus", "1", "--shard", "samples"], env=ENV, **q)]
    if kind == "bench1":       # ONE process: the default line without its sub-records
        return [subprocess.Popen(b + ["--also-stress", "off", "--also-config3", "off"], env=ENV, **q)]
    if kind == "bench2p":      # two ranks, problem sharding (no collective on the data path)
        return [subprocess.Popen(b + ["--gpus", "2", "--also-stress", "off", "--also-config3", "off"], env=ENV, **q)]
    raise SystemExit("unknown kind " + kind)


def main():
    kind = sys.argv[1]
    seconds = sys.argv[2] if len(sys.argv) > 2 else "10"
    probe = sys.argv[3] if len(sys.argv) > 3 else "pk_probe_010"
    import tempfile
    tmp = tempfile.mkdtemp(prefix="trigger_")
    log = os.path.join(tmp, "probe.txt")      # (a file, not a pipe: a probe that logs many events must not block on its reader)
    env = dict(ENV, PK_AGG=kind[3:]) if kind.startswith("in:") else ENV      # in:<mode>: the aggressor kernel on a second stream of the probe's OWN process
    main_p = subprocess.Popen([os.path.join(ROOT, "tools", probe), seconds, "400", "4096"], stdout=open(log, "w"), stderr=subprocess.STDOUT, env=env)
    rounds = 0
    time.sleep(1.0)
    while main_p.poll() is None:
        procs = one_round(kind, tmp)
        while any(p.poll() is None for p in procs) and main_p.poll() is None:
            time.sleep(0.05)
        for p in procs:      # (the probe has ended: whoever is still there goes, with its children)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    p.kill()
        rounds += 1
        time.sleep(0.2)
    main_p.wait()
    out = open(log).read()
    verdict = [l for l in out.split("\n") if l.startswith("pk_probe (") or "logged" in l]
    print("[%s] %d rounds of visitors | %s" % (kind, rounds, " | ".join(v.strip()[:230] for v in verdict)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
