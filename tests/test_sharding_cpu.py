"""World-size-2 gloo tests (CPU) of the multi-GPU host logic: sample-axis sharding with one in-place
all-reduce of the contiguous gradient buffer, and problem sharding.  The arithmetic inside each rank is the oracle here
(no GPU in this container); what is under test is the partitioning / packing / reduction contract that
vgpmp_amd.sharding applies to the HIP planner on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import small_problem
    from oracle import vgpmp_oracle as orc
    from vgpmp_amd import sharding

    S, N, M, B = 10, 7, 4, 32
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=9, n_grid=24)
    nz = pb["noise"]
    s_loc, off = sharding.shard_samples(S, world, rank)
    local = orc.Noise(nz.omega, nz.beta, nz.w[off:off + s_loc], nz.eps[off:off + s_loc], nz.eps2[off:off + s_loc])
    # local loss / gradient: likelihood part of the local samples scaled by alpha / S_total, KL on rank 0 only
    alpha_local = pb["alpha"] * s_loc / S
    fw = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], local, alpha_local)
    g, _ = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], local, alpha_local, fw)
    fw0 = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], local, 0.0)
    gk, _ = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], local, 0.0, fw0)   # pure KL gradient
    keep = 1.0 if rank == 0 else 0.0
    parts = [torch.tensor(getattr(g, n) - (1.0 - keep) * getattr(gk, n)) for n in ("q_mu", "q_sqrt", "raw_ell", "raw_var")]
    parts += [torch.tensor([fw["lik"]]), torch.tensor([keep * fw["cv"]["kl"]])]
    # the planner's layout: ONE contiguous float64 buffer [q_mu | q_sqrt | raw_ell | raw_var | lik | kl], reduced in place
    flat = torch.cat([t.reshape(-1) for t in parts]).contiguous()
    sharding.allreduce_sum_(flat)
    tensors, o = [], 0
    for t in parts:
        tensors.append(flat[o:o + t.numel()].reshape(t.shape))
        o += t.numel()
    # problem sharding: contiguous blocks, gathered in order
    b, e = sharding.partition(7, world, rank)
    gathered = sharding.gather_results(list(range(b, e)))
    if rank == 0:
        full = orc.elbo_forward(pb["params"], pb["scene"], pb["X"], pb["Zy"], pb["y"], nz, pb["alpha"])
        gf, _ = orc.elbo_backward(pb["params"], pb["scene"], pb["X"], pb["Zy"], nz, pb["alpha"], full)
        np.savez(os.path.join(out_dir, "r0.npz"), q_mu=tensors[0].numpy(), q_sqrt=tensors[1].numpy(),
                 ell=tensors[2].numpy(), var=tensors[3].numpy(), lik=tensors[4].numpy(), kl=tensors[5].numpy(),
                 f_q_mu=gf.q_mu, f_q_sqrt=gf.q_sqrt, f_ell=gf.raw_ell, f_var=gf.raw_var, f_lik=full["lik"],
                 f_kl=full["cv"]["kl"], gathered=np.array(gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_sharding_allreduce_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(tmp_path / "r0.npz")
    for k in ("q_mu", "q_sqrt", "ell", "var"):
        np.testing.assert_allclose(z[k], z["f_" + k], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(z["lik"][0], z["f_lik"], rtol=1e-10)
    np.testing.assert_allclose(z["kl"][0], z["f_kl"], rtol=1e-12)
    assert list(z["gathered"]) == list(range(7))


def test_partition_properties():
    from vgpmp_amd import sharding
    for n in (0, 1, 7, 36, 55, 512):
        for world in (1, 2, 3, 8):
            parts = [sharding.partition(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_samples(1024, 8, 3) == (128, 384)
