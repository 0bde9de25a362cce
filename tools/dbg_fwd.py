import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vgpmp_amd import capi, engine, robots as rb, scenes
ps = rb.load_problemset("franka", "industrial"); spec = rb.load_robot("franka")
grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
S, M, N, P = 64, 30, 40, 12
qs = np.array([ps.queries[i % 36] for i in range(P)])
kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
b.extra_flags |= capi.NO_FUSE
def cmp(tag):
    torch.cuda.synchronize()
    out = []
    for name, x, y in (("f", a.f, b.f), ("R", a.view("R"), b.view("R")), ("q_mu", a.q_mu, b.q_mu), ("q_sqrt", a.q_sqrt, b.q_sqrt), ("lik", a.lik, b.lik), ("raw_ell", a.raw_ell, b.raw_ell), ("ctr", a.step_counter, b.step_counter)):
        d = (x.double() - y.double()).abs()
        out.append(f"{name}={float(d.max()):.2e}")
    print(tag, " ".join(out), "ctr", int(a.step_counter), int(b.step_counter))
for rep in range(2):
    for n in (1, 1, 2, 3, 40):
        a.run_steps(n); b.run_steps(n); cmp(f"run_steps({n})")
    a.step(); b.step(); cmp("step()")
