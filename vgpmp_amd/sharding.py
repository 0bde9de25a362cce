"""Multi-GPU sharding of the ELBO path: one process per GPU, torch.distributed (RCCL on ROCm).

Two axes shard naturally:
  * start-goal problems -- fully independent: block-partition the query list, no data-path collective,
    a host-side gather of results at the end;
  * the Monte-Carlo sample axis -- every rank draws the SAME Fourier basis and its own slice of the
    global sample stream (dims.sample_offset), evaluates loss/gradient of its samples with the KL term
    owned by rank 0 (problem.kl_scale), then ONE in-place all-reduce(sum) of the planner's contiguous
    [gradient | lik | kl] float64 buffer per step; every rank applies the identical Adam update.  The
    payload is ~28 KB at M=30, L=7: latency bound, one message.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

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


def allreduce_sum_(flat: torch.Tensor, group=None) -> None:
    """In-place sum over the ranks of one contiguous buffer: ONE collective, no packing.  Device tensors go through the
    group's own backend (RCCL when it is "nccl"); with a gloo group (the CPU rehearsal of the N > 1 path, or two ranks on
    one GPU in the tests) a device buffer is staged through the host."""
    import torch.distributed as dist
    if flat.is_cuda and dist.get_backend(group) != "nccl":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


class CapiComm:
    """The collective behind the C ABI (vgpmp_comm_init / vgpmp_allreduce_grads: RCCL resolved by libvgpmp_hip.so itself).
    The rendezvous id travels over an existing torch.distributed group (any backend) or, for a single rank, nowhere."""

    def __init__(self, world: int, rank: int, group=None):
        import ctypes as C

        from . import capi
        self.lib = capi.load(require=True)
        ident = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * capi.COMM_ID_BYTES)()
            capi.check(self.lib.vgpmp_comm_unique_id(buf), "vgpmp_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            import torch.distributed as dist
            if dist.get_backend(group) == "nccl":
                dev = ident.cuda()
                dist.broadcast(dev, 0, group=group)
                ident = dev.cpu()
            else:
                dist.broadcast(ident, 0, group=group)
        raw = (C.c_ubyte * capi.COMM_ID_BYTES)(*ident.tolist())
        self.handle = C.c_void_p()
        capi.check(self.lib.vgpmp_comm_init(raw, int(world), int(rank), C.byref(self.handle)), "vgpmp_comm_init")

    def allreduce_sum_(self, flat: torch.Tensor) -> None:
        from . import capi
        assert flat.is_cuda and flat.dtype == torch.float64 and flat.is_contiguous()
        capi.check(self.lib.vgpmp_allreduce_grads(self.handle, capi.ptr(flat), flat.numel(),
                                                  int(torch.cuda.current_stream(flat.device).cuda_stream)),
                   "vgpmp_allreduce_grads")

    def close(self) -> None:
        if self.handle:
            self.lib.vgpmp_comm_destroy(self.handle)
            self.handle = None


class SampleShardedPlanner:
    """Drives a PlannerBatch that holds this rank's slice of the samples (dims.sample_offset / S_total, KL on the rank
    with kl_scale = 1).  One step = local forward + reverse, ONE in-place all-reduce of the planner's contiguous
    [gradient | lik | kl] buffer, the identical Adam update on every rank."""

    def __init__(self, planner, group=None, comm: Optional["CapiComm"] = None):
        self.planner, self.group, self.comm = planner, group, comm
        self.schedule = ("per step: vgpmp_elbo_step (forward + reverse: stage 1-3, likelihood, reverse paths, one launch for the "
                         "gradients; the next step's prior noise drawn beside the path assembly), the all-reduce, vgpmp_adam_step (one launch)")

    def _allreduce(self, buf: Optional[torch.Tensor] = None) -> None:
        buf = self.planner.reduce_buf if buf is None else buf
        if self.comm is not None:
            self.comm.allreduce_sum_(buf)
        else:
            allreduce_sum_(buf, self.group)

    def step(self) -> None:
        from . import capi
        pl = self.planner
        # consecutive steps: step t draws the prior noise of step t + 1 beside its path assembly (no noise launches)
        # (which step's draws the buffers hold is tracked by the planner itself -- PlannerBatch.noise_ahead_step -- so a direct
        #  call on it between two steps here, which redraws them, cannot be paired with the wrong eps)
        keep = pl.extra_flags
        pl.extra_flags = keep | capi.NOISE_AHEAD | capi.NOISE_READY
        try:
            pl.accumulate_grad(generate=True, step=pl.t)      # local samples; KL only where kl_scale = 1
        finally:
            pl.extra_flags = keep
        self._allreduce()
        pl.adam_only()                                         # identical update on every rank

    def run_steps(self, steps: int) -> None:
        """`steps` sharded steps.  With the C ABI's communicator (or a single rank and no group) the whole loop is ONE call,
        vgpmp_elbo_steps_reduced: step, all-reduce and Adam enqueued from C -- at eight ranks a step is ~75 us of device work,
        less than three host calls through Python."""
        pl = self.planner
        one_rank = self.comm is None and getattr(self, "_single_rank", False)
        if steps > 0 and (self.comm is not None or one_rank):
            import ctypes as C

            from . import capi
            from .engine import trainable_mask
            if pl.lik_variables or pl.z_variables:
                # (the C call refuses them too: its exchange buffer and update carry q_mu, q_sqrt, lengthscales, variance only)
                raise NotImplementedError("trainable sigma_obs / alpha / inducing locations do not shard over samples")
            what = (0 if pl.fuse else capi.NO_FUSE) | pl.extra_flags
            ready = pl.noise_ahead_step == pl.t
            capi.check(pl.lib.vgpmp_elbo_steps_reduced(
                C.byref(pl.dims), capi.ptr(pl.scene.dev_robot), C.byref(pl.scene.sdf), C.byref(pl._problem), C.byref(pl._params),
                C.byref(pl._am), C.byref(pl._av), C.byref(pl._noise), C.byref(pl._out), capi.ptr(pl.workspace), pl.workspace.numel(),
                what | (capi.NOISE_READY if ready else 0), trainable_mask(pl.trainable), pl.lr, pl.t, pl.seed, pl.problem_base, pl.t,
                int(steps), self.comm.handle if self.comm is not None else None, capi.ptr(pl.reduce_buf), pl.reduce_buf.numel(),
                pl.scene._stream()), "vgpmp_elbo_steps_reduced")
            pl.t += int(steps)
            # the last step drew ahead only if the library ran the few-problem schedule (the large-batch one ignores the flags)
            fused = any(n.startswith("stage1_kernel") for n in capi.last_schedule(pl.lib))
            pl.noise_ahead_step = pl.t if fused else None
            return
        for _ in range(steps):
            self.step()

    def elbo(self) -> torch.Tensor:
        pl = self.planner
        pl.elbo(generate=True, step=pl.t)                      # (draws its own noise: clears pl.noise_ahead_step)
        # only the [lik | kl] tail is fresh after a forward-only pass: reduce just that (contiguous, 2 P doubles) and
        # leave the gradient part of the buffer alone (summing it in place would scale a stale gradient by the world size)
        self._allreduce(pl.reduce_buf[pl.reduce_buf.numel() - 2 * pl.P:])
        return pl.lik - pl.kl


def gather_results(local: List, group=None) -> List:
    """Host-side gather of per-problem results (problem sharding): list of picklable objects."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return [x for part in out for x in part]
