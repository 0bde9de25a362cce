"""Multi-GPU sharding of the ELBO path: one process per GPU, torch.distributed (RCCL on ROCm).

Two axes shard naturally:
  * start-goal problems -- fully independent: block-partition the query list, no data-path collective,
    a host-side gather of results at the end;
  * the Monte-Carlo sample axis -- every rank draws the SAME Fourier basis and its own slice of the
    global sample stream (dims.sample_offset), evaluates loss/gradient of its samples with the KL term
    owned by rank 0 (problem.kl_scale), then ONE all-reduce(sum) of the packed gradient (+ ELBO pieces)
    per step; every rank applies the identical Adam update.  The payload is ~14 KB at M=30, L=7:
    latency bound, so it is sent as a single flat float64 buffer.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch


def partition(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of `n_items` for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_samples(num_samples: int, world: int, rank: int) -> Tuple[int, int]:
    """(local sample count, offset of the first local sample) for sample-axis sharding."""
    b, e = partition(num_samples, world, rank)
    return e - b, b


def pack(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])


def unpack_into(flat: torch.Tensor, tensors: Sequence[torch.Tensor]) -> None:
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].reshape(t.shape).to(t.dtype))
        o += n


def allreduce_sum(tensors: Sequence[torch.Tensor], group=None) -> None:
    """One all-reduce for a list of tensors (packed into a single flat buffer, then scattered back)."""
    import torch.distributed as dist
    flat = pack(tensors)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    unpack_into(flat, tensors)


class SampleShardedPlanner:
    """Drives a PlannerBatch that holds this rank's slice of the samples."""

    def __init__(self, planner, group=None):
        self.planner, self.group = planner, group

    def step(self) -> None:
        pl = self.planner
        pl.loss_and_grad(generate=True, step=pl.t)            # local samples; KL only where kl_scale = 1
        allreduce_sum(list(pl.grad) + [pl.lik, pl.kl], self.group)
        pl.adam_only()                                         # identical update on every rank

    def elbo(self) -> torch.Tensor:
        pl = self.planner
        pl.elbo(generate=True, step=pl.t)
        allreduce_sum([pl.lik, pl.kl], self.group)
        return pl.lik - pl.kl


def gather_results(local: List, group=None) -> List:
    """Host-side gather of per-problem results (problem sharding): list of picklable objects."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return [x for part in out for x in part]
