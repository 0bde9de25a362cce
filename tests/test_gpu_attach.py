"""The likelihood kernels beside what made them return wrong values up to round 5.

Round 5 found that `vgpmp_log_prob` / the likelihood launch of the ELBO step returned WRONG GRADIENTS (by up to 150, sixteen
consecutive configurations at a time) in a launch that ran while another process used the GPU; the suite stayed green only because
tests/conftest.py waited for its rank processes before the first test.  Round 6 took the failure apart (profiles/r06/flake.md):

  * the victim is ONE instruction form: a packed-FP32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- what the compiler's
    SLP vectoriser makes of three-component arithmetic) whose op_sel and op_sel_hi both select the HIGH register of source 1 reads
    0.0 for that operand in the low result of lanes 48-63 (tools/depack_pk.py, tools/pk_probe.hip);
  * the trigger is not a process arriving or a preemption but a NEIGHBOUR ON THE COMPUTE UNIT: while another wave of the same unit runs
    one of gfx950's wide f16 / bf16 matrix instructions (v_mfma_f32_16x16x32_f16, 32x32x16_f16, 16x16x32_bf16) -- from another
    process OR from another stream of the same process -- every launch of the probe is wrong; with the two streams on disjoint halves
    of the compute units none is (tools/trigger_probe.py).  The "other process" of round 5 was a bench running this library's own
    prior draws, which ARE such instructions.

The library is built without packed instructions (vgpmp_amd/build.py; tests/test_capi_load.py holds the disassembly at zero), so it
has no victim.  These tests hold that where it matters, on the device, both ways:

  * beside the matrix instructions themselves: `vgpmp_debug_mfma_load` (include/vgpmp_debug.h) keeps a second stream of THIS process
    busy with nothing but v_mfma_f32_16x16x32_f16 while the stand-alone likelihood and the ELBO step run over and over -- the
    condition under which a library WITH the packed form is wrong in every launch (checked on a build with the vectorisers on:
    profiles/r06/flake.md);
  * beside the process mix of round 5 (tests/attach_worker.py: two gloo rank processes, a two-rank `bench.py --shard samples`, plain
    visitors), as VERDICT r5 asked; beside another planner batch on a second stream; beside another process that runs matrix kernels;
    and beside the in-process matrix kernel AND the process mix together -- the neighbourhood in which the library as round 5 shipped
    it fails fastest (which waves meet on a compute unit is a matter of placement that other processes' work changes:
    profiles/r06/flake.md, "What triggers it").

Every output is compared bit for bit.  Also kept from the hunt: an operand fence in the gradient's second sweep, and the
prefix-scalar batch form retired for the pipelined form with 8-wide per-frame sums.  Deployment: INTEGRATION.md ("Deployment
constraints"), include/vgpmp.h (vgpmp_elbo_step).  Semantics protected: likelihoods/likelihood.py:146-176, utils/sampler.py:103-120
(deterministic given the inputs)."""
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


class _MatrixLoad:
    """Work kept running on a second stream of this process: by default `launches` kernels of vgpmp_debug_mfma_load (~10 ms each) per
    start(); with `planner`, that many optimisation steps of another planner batch (the library's own kernels, prior draws included)."""

    def __init__(self, lib, launches=40, workgroups=1024, iterations=20000, planner=None):
        iterations = int(os.environ.get("VGPMP_TEST_MFMA_ITERS", iterations))      # (measurement knobs: profiles/r06/flake.md)
        workgroups = int(os.environ.get("VGPMP_TEST_MFMA_WGS", workgroups))
        self.lib, self.launches, self.workgroups, self.iterations, self.planner = lib, launches, workgroups, iterations, planner
        self.stream = torch.cuda.Stream()
        self.sink = torch.zeros(4, dtype=torch.float32, device="cuda")
        self.end, self.stop = None, False

    def start(self):
        if self.planner is not None:      # a thread of its own keeps the other batch stepping (the host side of a step is most of its time)
            import threading

            def steps():      # until finish(): the queue kept at most `launches` steps deep
                torch.cuda.set_device(0)
                with torch.cuda.stream(self.stream):
                    while not self.stop:
                        for _ in range(self.launches):
                            self.planner.run_steps(1)
                        self.stream.synchronize()

            self.thread = threading.Thread(target=steps, daemon=True)
            self.thread.start()
            return
        for _ in range(self.launches):
            capi.check(self.lib.vgpmp_debug_mfma_load(capi.ptr(self.sink), self.workgroups, self.iterations, int(self.stream.cuda_stream)),
                       "vgpmp_debug_mfma_load")
        self.end = torch.cuda.Event()
        self.end.record(self.stream)

    def running(self) -> bool:
        if self.planner is not None:
            return getattr(self, "thread", None) is not None and self.thread.is_alive()
        return self.end is not None and not self.end.query()

    def finish(self):
        self.stop = True
        if getattr(self, "thread", None) is not None:
            self.thread.join()
        self.stream.synchronize()


def _beside_matrix_load(lib, body, seconds=4.0, max_reps=1 << 30, **load_kw):
    """body() over and over while the matrix kernel runs on the other stream; returns (repetitions, repetitions that ended while the
    matrix kernel was still running, differences)."""
    load = _MatrixLoad(lib, **load_kw)
    t0, reps, beside, bad = time.time(), 0, 0, []
    while time.time() - t0 < seconds and len(bad) < 5 and reps < max_reps:
        if not load.running():
            load.start()
        d = body()
        reps += 1
        beside += load.running()
        if d:
            bad.append((reps, d))
    load.finish()
    return reps, beside, bad


def test_likelihood_is_bit_stable_beside_f16_matrix_kernels():
    """The stand-alone likelihood on 200 000 fixed joint configurations while a second stream of this process runs f16 matrix
    instructions and nothing else: every repetition against the first one (taken on an idle device)."""
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
        torch.cuda.current_stream().synchronize()
        return logp.clone(), dl.clone()

    torch.cuda.synchronize()
    ref = run()

    def body():
        out = run()
        return [(i, int((x != y).sum()), float((x.double() - y.double()).abs().max())) for i, (x, y) in enumerate(zip(out, ref))
                if not torch.equal(x, y)]

    reps, beside, bad = _beside_matrix_load(sc.lib, body)
    print(f"PARITY matrix load (vgpmp_log_prob): {reps} repetitions, {beside} of them beside the running f16 matrix kernel; repetitions that differed: {len(bad)}")
    assert not bad, bad[:5]
    assert beside >= 20, (reps, beside)


_STEP_CASES = [      # one per schedule of vg_elbo_steps and likelihood form
    ("franka_x12", "franka", 12, dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256)),      # large batch, pipelined likelihood (8-wide sums)
    ("franka_x2", "franka", 2, dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256)),        # few problems: 8 lanes per configuration
    ("franka_x4", "franka", 4, dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256)),        # 3-4 problems: stage3 + batch form
    ("franka_x9_s7", "franka", 9, dict(num_samples=7, num_inducing=24, num_data=70, num_bases=256)),      # few samples: small16 prior, sc8 paths
    ("arm14_x8", "arm14", 8, dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256)),          # 14 joints: 16-wide pipelined likelihood
]


@pytest.mark.parametrize("tag,robot,num_problems,shape", _STEP_CASES, ids=[c[0] for c in _STEP_CASES])
def test_elbo_steps_are_bit_stable_beside_f16_matrix_kernels(tag, robot, num_problems, shape):
    """Two planners of the same seed: one takes its steps on an idle device first, the other the same steps while the f16 matrix kernel
    runs on the second stream; variables, paths, log-densities and gradients bit for bit after every step -- one case per launch
    schedule and likelihood form.  (On the library as round 5 shipped it the 14-joint case is wrong at the first step; the 7-joint
    cases need a second process on the device as well: profiles/r06/flake.md, "What triggers it".)"""
    if robot == "arm14":
        spec = rb.synthetic_arm(14)
        grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -1.2), seed=0)
        sc = engine.DeviceScene(spec, grid, (0, 0, 0))
        qs = np.random.default_rng(0).uniform(-2.0, 2.0, (num_problems, 2, 14))
    else:
        ps, spec, sc = _scene()
        qs = np.array([ps.queries[i % 36] for i in range(num_problems)])
    kw = dict(lengthscales=[2.0] * spec.dof, variance=0.2, seed=4, **shape)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    names = ("q_mu", "q_sqrt", "f", "logp")
    steps, alone = 60, []
    for _ in range(steps):
        a.run_steps(1)
        torch.cuda.current_stream().synchronize()
        alone.append({k: getattr(a, k).clone() for k in names} | {"G": a.view("G")})
    it = iter(alone)

    def body():
        want = next(it)
        b.run_steps(1)
        torch.cuda.current_stream().synchronize()
        got = {k: getattr(b, k) for k in names} | {"G": b.view("G")}
        return [(k, int((got[k] != want[k]).sum())) for k in want if not torch.equal(got[k], want[k])]

    reps, beside, bad = _beside_matrix_load(sc.lib, body, seconds=30.0, max_reps=steps)
    print(f"PARITY matrix load (ELBO steps, {tag}): {reps} steps, {beside} repetitions beside the running f16 matrix kernel; first differences: {bad[:1]}")
    assert not bad, bad[:2]
    assert beside >= 20, (reps, beside)


def test_elbo_steps_are_bit_stable_beside_another_planner_on_a_second_stream():
    """One process, two streams, two planner batches: 12 problems take their steps while 64 OTHER problems (the large-batch schedule:
    f16 matrix instructions in the prior draws, float64 matrix instructions in the covariance path, the masked likelihood) step on a
    second stream.  The 12 problems' variables, paths, log-densities and gradients against the same steps taken on an idle device."""
    ps, spec, sc = _scene()
    qs = np.array([ps.queries[i % 36] for i in range(12)])
    kw = dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256, lengthscales=[2.0] * 7, variance=0.2)
    a, b = engine.PlannerBatch(sc, qs, seed=4, **kw), engine.PlannerBatch(sc, qs, seed=4, **kw)
    other = engine.PlannerBatch(sc, np.array([ps.queries[(5 * i + 1) % 36] for i in range(64)]), seed=9, **kw)
    names = ("q_mu", "q_sqrt", "f", "logp")
    steps, alone = 80, []
    for _ in range(steps):
        a.run_steps(1)
        torch.cuda.current_stream().synchronize()
        alone.append({k: getattr(a, k).clone() for k in names} | {"G": a.view("G")})
    it = iter(alone)

    def body():
        want = next(it)
        b.run_steps(1)
        torch.cuda.current_stream().synchronize()
        got = {k: getattr(b, k) for k in names} | {"G": b.view("G")}
        return [(k, int((got[k] != want[k]).sum())) for k in want if not torch.equal(got[k], want[k])]

    reps, beside, bad = _beside_matrix_load(sc.lib, body, seconds=60.0, max_reps=steps, planner=other, launches=50)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(other.q_mu).all())
    print(f"PARITY second stream (ELBO steps beside another planner batch): {reps} steps, {beside} of them while the other batch was stepping; first differences: {bad[:1]}")
    assert not bad, bad[:2]
    assert beside >= 20, (reps, beside)


def test_elbo_steps_are_bit_stable_beside_a_process_running_f16_matrix_kernels(attach_visitors):
    """ANOTHER PROCESS keeps the device busy with f16 matrix instructions (tests/attach_worker.py mfma: bursts on a second stream, a
    small vector kernel on the first) while 12 problems take 60 steps: against the same steps taken before that process started.  A
    scenario check, not a proven detector: the library as round 5 shipped it was wrong at EVERY step beside tools/pk_probe_dflt in its
    aggressor mode (PK_AGG=2; thousands of wrong log-densities and gradients per step, profiles/r06/flake_sessions/matrix_load_tests_r5.txt)
    but not beside this visitor; the detectors with proven teeth are the in-process test of the stand-alone likelihood above (wrong in
    every repetition on that library) and the two process-mix tests (wrong in every session)."""
    out_dir, _ = attach_visitors
    ps, spec, sc = _scene()
    qs = np.array([ps.queries[i % 36] for i in range(12)])
    kw = dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    names = ("q_mu", "q_sqrt", "f", "logp")
    steps, alone = 60, []
    for _ in range(steps):
        a.run_steps(1)
        torch.cuda.current_stream().synchronize()
        alone.append({k: getattr(a, k).clone() for k in names} | {"G": a.view("G")})
    open(os.path.join(out_dir, "go_mfma_1"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(out_dir, "ready_mfma_1")) and time.time() - t0 < 180:
        time.sleep(0.05)
    try:
        assert os.path.exists(os.path.join(out_dir, "ready_mfma_1")), open(os.path.join(out_dir, "attach_visitors.log")).read()[-2000:]
        bad = []
        for i, want in enumerate(alone):
            b.run_steps(1)
            torch.cuda.current_stream().synchronize()
            got = {k: getattr(b, k) for k in names} | {"G": b.view("G")}
            d = [(k, int((got[k] != want[k]).sum())) for k in want if not torch.equal(got[k], want[k])]
            if d:
                bad.append((i + 1, d))
                if len(bad) >= 3:
                    break
        still = not os.path.exists(os.path.join(out_dir, "done_mfma_1"))
    finally:
        open(os.path.join(out_dir, "stop_mfma_1"), "w").close()
        t1 = time.time()
        while not os.path.exists(os.path.join(out_dir, "done_mfma_1")) and time.time() - t1 < 180:
            time.sleep(0.05)
    print(f"PARITY matrix process (ELBO steps): {steps} steps beside a process running f16 matrix kernels (still running at the end: {still}); first differences: {bad[:1]}")
    assert not bad, bad[:2]
    assert still, "the matrix-kernel process had gone before the steps were taken"
    assert open(os.path.join(out_dir, "done_mfma_1")).read().strip() == "0", open(os.path.join(out_dir, "attach_visitors.log")).read()[-2000:]


def test_two_planners_stay_together_beside_matrix_load_and_visiting_processes(attach_visitors):
    """The strongest neighbourhood found (profiles/r06/flake_sessions/neighbour_vs_r5.txt): f16 matrix kernels on a second stream of THIS
    process AND other processes running kernels of any kind on the device -- the library as round 5 shipped it parts two same-seed planners
    within seven steps there (within one step beside a process that runs matrix kernels itself).  Two planners of the same seed, 12
    problems, one step each per repetition, bit for bit, while the round-5 process mix comes and goes and the matrix kernel runs."""
    out_dir, _ = attach_visitors
    ps, spec, sc = _scene()
    qs = np.array([ps.queries[i % 36] for i in range(12)])
    kw = dict(num_samples=64, num_inducing=30, num_data=40, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    load = _MatrixLoad(sc.lib)
    beside = [0]

    def body():
        if not load.running():
            load.start()
        a.run_steps(1); b.run_steps(1)
        torch.cuda.current_stream().synchronize()
        beside[0] += load.running()
        pairs = {"q_mu": (a.q_mu, b.q_mu), "q_sqrt": (a.q_sqrt, b.q_sqrt), "f": (a.f, b.f), "logp": (a.logp, b.logp), "G": (a.view("G"), b.view("G"))}
        return [(k, int((x != y).sum())) for k, (x, y) in pairs.items() if not torch.equal(x, y)]

    reps, secs, bad = _visitors(out_dir, 3, body)
    load.finish()
    print(f"PARITY matrix load + attach (two planners): {reps} repetitions in {secs:.1f} s, {beside[0]} of them beside the running f16 matrix kernel "
          f"and the visiting processes; first differences: {bad[:1]}")
    assert not bad, bad[:2]
    assert beside[0] >= 20, (reps, beside[0])
