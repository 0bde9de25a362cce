"""Measurement / QA aid: does any kernel read memory nobody wrote?  Freed blocks of torch's caching allocator are poisoned (NaN or large random
values) before the planners' tensors and workspaces are carved from them; the merged batch schedule and the one-launch-per-kernel schedule
(VGPMP_NO_FUSE) must still agree bit for bit and stay finite.      python tools/poison_probe.py [reps] [--lds]
--lds: the LDS of every CU is filled with NaNs before every call as well (tools/lds_poison.hip -> tools/liblds_poison.so): a kernel that reads
an LDS word it never wrote then shows."""
import ctypes, os
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vgpmp_amd import capi, engine, robots as rb, scenes
ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
LDS = "--lds" in sys.argv
if LDS:
    _pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_poison.so"))
    _pl.lds_poison.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    _sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def poison_lds(k=0):
    if LDS:
        pat = 0x7fc00000 if k % 2 == 0 else 0x7ff80000      # float32 / (upper word of) float64 quiet NaN
        assert _pl.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), pat, ctypes.c_void_p(_sink.data_ptr())) == 0
rng = np.random.default_rng(0)
bad = 0
for rep in range(reps):
    for (S, M, N, P) in ((64, 30, 40, 12), (7, 24, 70, 20), (128, 30, 100, 8), (20, 10, 50, 12)):
        # poison: many blocks of many sizes, filled, then freed (the allocator hands them out again)
        junk = []
        for k in range(3):      # (large blocks: the workspaces are carved from these)
            t = torch.empty((1 << 28) // 4, dtype=torch.float32, device="cuda")
            t.fill_(float("nan") if (rep + k) % 2 else 3.0e38)
            junk.append(t)
        for k in range(40):
            n = int(rng.choice([256, 4096, 65536, 1 << 20, 1 << 22, 1 << 24]))
            t = torch.empty(n // 4, dtype=torch.float32, device="cuda")
            if (rep + k) % 2:
                t.fill_(float("nan"))
            else:
                t.copy_(torch.from_numpy((rng.standard_normal(n // 4) * 1e6).astype(np.float32)))
            junk.append(t)
        torch.cuda.synchronize()
        del junk
        qs = np.array([ps.queries[i % 36] for i in range(P)])
        kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
        a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
        b.extra_flags |= capi.NO_FUSE
        msg = None
        for blk in range(2 if not LDS else 6):
            if LDS:      # (one step per call: poison between every two steps)
                for k in range(4):
                    poison_lds(k); a.run_steps(1); poison_lds(k + 1); b.run_steps(1)
            else:
                a.run_steps(7); b.run_steps(7)
            poison_lds(blk); a.step(); poison_lds(blk); b.step()
            torch.cuda.synchronize()
            xs = a._variables() + a._moments() + [a.f, a.lik, a.kl]; ys = b._variables() + b._moments() + [b.f, b.lik, b.kl]
            nf = [i for i, x in enumerate(xs) if not torch.isfinite(x).all()] + [100 + i for i, y in enumerate(ys) if not torch.isfinite(y).all()]
            diff = [(i, float((x.double() - y.double()).abs().max())) for i, (x, y) in enumerate(zip(xs, ys)) if not torch.equal(x, y)]
            if (nf or diff) and msg is None:
                msg = (blk, "non-finite", nf, "different", diff[:6])
        if msg:
            bad += 1
            print("rep", rep, (S, M, N, P), msg, flush=True)
print("repetitions", reps, "shape-runs with a finding:", bad)
