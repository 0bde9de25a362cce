"""The likelihood kernels while other processes arrive on and leave the device.

Round 5 found that `vgpmp_log_prob` / the likelihood launch of the ELBO step returned WRONG GRADIENTS (by up to 150, sixteen
consecutive configurations at a time) in a launch that was in flight while another process attached to or left the GPU; the suite
stayed green only because tests/conftest.py waits for its rank processes before the first test.  These tests do the opposite on
purpose: the SAME process mix (tests/attach_worker.py: two gloo rank processes, a two-rank `bench.py --shard samples`, then plain
visitors) comes and goes WHILE the kernels run over and over, and every output is compared bit for bit.

  * the stand-alone likelihood (`vgpmp_log_prob`): the damage sat in the gradient's second sweep and went with the SCHEDULE of that
    loop; an operand fence pins it (csrc/fk_sdf.hip, `vg_sweep_fence`): 0 of 8 reproducer sessions, 7 of 8 before;
  * the batch form inside the ELBO step: ONE of its forms -- the prefix-scalar form that batches of up to 8 joints ran -- parted two
    same-seed planners in 31 of 38 reproducer sessions whatever was fenced; every other form (LDS state, 8 lanes per configuration,
    the pipelined register form at 7 and at 14 joints) 0 of 36.  The prefix form was retired; batches of 7-joint arms run the
    pipelined form with 8-wide per-frame sums (the retired form's speed).

What the hardware does to the two schedules that failed is NOT established; these tests are the standing check, and one process per
GPU remains the stated deployment rule ("Deployment constraints").
History, what was tried and the deployment constraint that follows: profiles/r06/flake.md, INTEGRATION.md ("Deployment constraints"),
include/vgpmp.h (vgpmp_elbo_step).  Semantics protected: likelihoods/likelihood.py:146-176, utils/sampler.py:103-120 (deterministic
given the inputs)."""
import ctypes as C
import os
import time

import numpy as np
import pytest
import torch

from vgpmp_amd import capi, engine, robots as rb, scenes

pytestmark = pytest.mark.gpu


def _scene():
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    return ps, spec, engine.DeviceScene(spec, grid, ps.object_positions[0])


def _visitors(out_dir, k, body, min_seconds=8.0, max_seconds=180.0):
    """Runs body() over and over from the moment the k-th mix of visitors is triggered until it has left (and min_seconds have passed)."""
    open(os.path.join(out_dir, f"go_attach_{k}"), "w").close()
    done = os.path.join(out_dir, f"done_attach_{k}")
    t0, reps, bad = time.time(), 0, []
    while time.time() - t0 < max_seconds:
        reps += 1
        d = body()
        if d:
            bad.append((reps, round(time.time() - t0, 2), d))
            if len(bad) >= 5:
                break
        if os.path.exists(done) and time.time() - t0 > min_seconds:
            break
    t_wait = time.time()
    while not os.path.exists(done) and time.time() - t_wait < max_seconds:      # (the visitors must be gone before the next test starts)
        time.sleep(0.1)
    assert os.path.exists(done), "the visitors did not finish"
    rcs = open(done).read().split()
    # (what matters here is that processes ARRIVED and LEFT; a visitor that failed for a reason of its own still did both)
    assert len(rcs) >= 7, open(os.path.join(out_dir, "attach_visitors.log")).read()[-2000:]
    if any(r != "0" for r in rcs):
        print(f"NOTE attach: visitor exit codes {rcs} (see attach_visitors.log)")
    return reps, time.time() - t0, bad


def test_likelihood_is_stable_while_processes_attach(attach_visitors):
    """The stand-alone likelihood on 200 000 fixed joint configurations, every repetition against the first (outputs pre-filled with a
    sentinel: a row that keeps it was never written)."""
    out_dir, _ = attach_visitors
    ps, spec, sc = _scene()
    rng = np.random.default_rng(0)
    n = 200000
    g = torch.tensor(rng.uniform(spec.low, spec.high, size=(n, spec.dof)).astype(np.float32), device="cuda")
    logp = torch.empty(n, dtype=torch.float32, device="cuda")
    dl = torch.empty((n, spec.dof), dtype=torch.float32, device="cuda")

    def run():
        logp.fill_(12345.0); dl.fill_(12345.0)
        capi.check(sc.lib.vgpmp_log_prob(capi.ptr(sc.dev_robot), spec.dof, C.byref(sc.sdf), capi.ptr(g), n, capi.ptr(logp),
                                         capi.ptr(dl), sc._stream()), "vgpmp_log_prob")
        return logp.clone(), dl.clone()

    ref = run()
    torch.cuda.synchronize()

    def body():
        out = run()
        torch.cuda.synchronize()
        return [(i, int((x != y).sum()), float((x.double() - y.double()).abs().max())) for i, (x, y) in enumerate(zip(out, ref))
                if not torch.equal(x, y)]

    reps, secs, bad = _visitors(out_dir, 1, body)
    print(f"PARITY attach (vgpmp_log_prob): {reps} repetitions in {secs:.1f} s beside the visiting processes; repetitions that differed: {len(bad)}")
    assert not bad, bad[:5]


def test_two_planners_stay_together_while_processes_attach(attach_visitors):
    """Two planners of the same seed (12 problems: the large-batch schedule), one optimisation step each per repetition; their variables,
    log-densities and gradients bit for bit after every repetition."""
    out_dir, _ = attach_visitors
    ps, spec, sc = _scene()
    qs = np.array([ps.queries[i % 36] for i in range(12)])
    kw = dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)

    def body():
        a.run_steps(1); b.run_steps(1)
        torch.cuda.synchronize()
        pairs = {"q_mu": (a.q_mu, b.q_mu), "q_sqrt": (a.q_sqrt, b.q_sqrt), "f": (a.f, b.f), "logp": (a.logp, b.logp), "G": (a.view("G"), b.view("G"))}
        return [(k, int((x != y).sum())) for k, (x, y) in pairs.items() if not torch.equal(x, y)]

    reps, secs, bad = _visitors(out_dir, 2, body)
    print(f"PARITY attach (two planners): {reps} repetitions in {secs:.1f} s beside the visiting processes; first differences: {bad[:1]}")
    assert not bad, bad[:2]
