"""The sample-sharded ELBO step (BASELINE config 4) as it runs on N GPUs, rehearsed with two processes on the
one GPU of the box: each rank owns half of the Monte-Carlo samples of the SAME global Philox stream, the KL term lives on
rank 0, the contiguous [gradient | lik | kl] buffer is summed in place by ONE collective per step, and both ranks apply the
same Adam update.  After three steps the parameters must equal those of the unsharded 16-sample planner."""
import numpy as np
import pytest
import torch

from shard_worker import M, N, S_TOTAL, STEPS, problem

pytestmark = pytest.mark.gpu


def test_two_ranks_three_steps_equal_the_unsharded_planner(shard_workers):
    from vgpmp_amd import engine
    out, _ = shard_workers
    r0, r1 = np.load(f"{out}/rank0.npz"), np.load(f"{out}/rank1.npz")
    assert int(r0["t"]) == STEPS and int(r1["t"]) == STEPS
    # replicated optimizer state: the two ranks hold the same bits
    for k in ("q_mu", "q_sqrt", "raw_ell", "raw_var", "elbo"):
        assert np.array_equal(r0[k], r1[k]), k
    spec, grid, off, q, kw = problem()
    sc = engine.DeviceScene(spec, grid, off)
    full = engine.PlannerBatch(sc, q, num_samples=S_TOTAL, **kw)
    start = {k: getattr(full, k).cpu().numpy().copy() for k in ("q_mu", "q_sqrt", "raw_ell", "raw_var")}
    elbos = []
    for _ in range(STEPS):
        full.loss_and_grad(generate=True, step=full.t)
        elbos.append(float((full.lik - full.kl)[0]))
        full.adam_only()
    torch.cuda.synchronize()
    # the sharded sum differs from the one-launch sum by float32 summation order only
    np.testing.assert_allclose(r0["elbo"], np.array(elbos), rtol=2e-5)
    lr = kw["learning_rate"]
    for k in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
        got, want = r0[k], getattr(full, k).cpu().numpy()
        assert np.abs(want - start[k]).max() > 0.5 * lr, f"{k} did not move"
        # Adam normalises the gradient: a float32-level difference moves a parameter by far less than lr per step
        assert np.abs(got - want).max() < STEPS * lr * 2e-2, (k, np.abs(got - want).max())


def test_capi_communicator_single_rank_is_identity():
    """vgpmp_comm_* over RCCL with a one-rank communicator on the box's GPU: the in-place sum leaves the buffer as it is
    (the N-rank exchange is the same call; an 8-GPU node is not available to the tests)."""
    from vgpmp_amd import sharding
    comm = sharding.CapiComm(1, 0)
    buf = torch.arange(3479, dtype=torch.float64, device="cuda") * 0.25 - 100.0
    want = buf.clone()
    comm.allreduce_sum_(buf)
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    comm.close()


def test_sharded_loop_through_rccl_on_one_rank():
    """vgpmp_elbo_steps_reduced with a REAL communicator (one rank: the all-reduce of the gradient buffer is enqueued between
    the reverse pass and Adam of every step, on the step's stream) equals the same loop without one, bit for bit."""
    from vgpmp_amd import engine, sharding
    spec, grid, off, q, kw = problem()
    sc = engine.DeviceScene(spec, grid, off)
    out = []
    for with_comm in (False, True):
        pl = engine.PlannerBatch(sc, q, num_samples=S_TOTAL, **kw)
        comm = sharding.CapiComm(1, 0) if with_comm else None
        sp = sharding.SampleShardedPlanner(pl, comm=comm)
        if comm is None:
            sp._allreduce = lambda buf=None: None
            sp._single_rank = True
        sp.run_steps(4)
        torch.cuda.synchronize()
        out.append([t.clone() for t in (pl.q_mu, pl.q_sqrt, pl.raw_ell, pl.raw_var, pl.reduce_buf)])
        if comm is not None:
            comm.close()
    for x, y in zip(*out):
        assert torch.equal(x, y)


def test_bench_starts_its_own_ranks(bench_gpus2):
    """`python bench.py --gpus 2 --shard samples` with no external launcher: bench.py starts two fresh
    rank processes before touching the GPU, they rendezvous on 127.0.0.1 (gloo: both ranks share this box's one GPU), run
    the sample-sharded step with one all-reduce per step, and rank 0's ONE JSON line comes back on stdout."""
    import json
    so, se = bench_gpus2
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (so[-2000:], se[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 5
    assert d["config"]["collective_ranks"] == 2 and "512 on this rank" in d["config"]["workload"]
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0 and len(lines[0]) <= 6000
    # (config 4's 16 MB table is cache resident: the line prices no fraction of the HBM peak for it)
    assert d["roofline"]["frac"] is None and d["roofline"]["bound"].startswith("cache")


def test_bench_eight_ranks_both_sharding_modes(bench_gpus8):
    """What a SCALE run launches at N = 8, rehearsed on one GPU (eight ranks share the device over gloo: NO scaling number can be
    earned this way, only that the paths start and agree): `python bench.py --gpus 8` -- problems sharded, no collective, the
    config-5 share (batch_512: 64 problems per rank = the 512-problem batch) riding along -- and `--gpus 8 --shard samples` --
    one all-reduce per step over eight ranks, with the collective's cost per step separated by the ranks themselves."""
    import json
    so, se = bench_gpus8["problems"]
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (so[-1500:], se[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert "problems sharded x8" in d["config"]["parallelism"]
    # (the sub-record itself is in bench_detail.json; the one stdout line carries its summary: <= 6 KB, tests/test_launch_cpu.py)
    assert len(lines[0]) <= 6000 and "batch_512" not in d
    b = d["summary"]["batch_512"]
    assert b["ms_per_step"] > 0 and b["sdf_frac_hbm"] > 0 and b["problems_total"] == 512
    assert list(d)[-1] == "summary" and d["summary"]["line"]["n_gpus"] == 8
    so, se = bench_gpus8["samples"]
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (so[-1500:], se[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["collective_ranks"] == 8
    assert "128 on this rank" in d["config"]["workload"]
    cb = d["collective_breakdown"]
    assert cb["ranks"] == 8 and cb["samples_per_rank"] == 128
    assert cb["rank_local_us_per_step"] > 0 and abs(cb["measured_us_per_step"] - 1e3 * d["ms_per_step"]) < 0.02
    assert abs(cb["collective_us_per_step"] - (cb["measured_us_per_step"] - cb["rank_local_us_per_step"])) < 0.02
    assert list(d)[-1] == "summary"
