"""BASELINE config 5 (synthetic 14-DoF arm, P = 45 spheres, large voxel table) against the float64 oracle, through the
C ABI, and the voxel-table layouts against each other.  Run with -m gpu on MI355X.

dof = 14 > 8 takes code no other robot reaches: two joints per lane and the non-scan chain of the 8-lane likelihood
kernel, a 15-frame run-time chain in the 1-lane kernels.  Layout tests: the voxel INDEX a query resolves to is the
reference's (utils/sdf_utils.py:62-66) under both table layouts and the records are the same floats, so everything
downstream must be BITWISE equal between layouts and with / without the free-space summary.
"""
import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
from helpers import TOL_GRAD, TOL_LIK, TOL_LOGP, assert_grads, device_centres, flipped_share, oracle_scene, synthetic_problem
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes

pytestmark = pytest.mark.gpu


def _engine():
    from vgpmp_amd import engine
    return engine


def _noise32(noise):
    r = lambda a: a.astype(np.float32).astype(np.float64)
    return orc.Noise(r(noise.omega), r(noise.beta), r(noise.w), r(noise.eps), r(noise.eps2))


@pytest.mark.parametrize("shape", [(41, 38, 45), (4, 4, 4), (5, 1, 9), (64, 64, 64)])
def test_sdf_layouts_bit_exact_against_oracle(shape):
    """Ragged extents (not multiples of the brick), a one-voxel-thick axis, exact lattice points and far-outside queries:
    index, value and gradient equal the oracle's under both layouts."""
    rng = np.random.default_rng(3)
    data = rng.normal(0.0, 0.3, shape)
    data[rng.random(shape) < 0.2] = 0.25          # equal neighbours -> exact-zero central differences -> 0.1
    origin, delta = np.array([-0.4, 0.1, -1.0]), 0.03
    og = orc.SDFGrid(data, origin, delta)
    ext = delta * np.array(shape)
    pos = origin + rng.uniform(-0.3, 1.3, (50000, 3)) * ext
    pos[:4000] = origin + delta * rng.integers(-2, np.array(shape) + 2, (4000, 3))       # cell boundaries hit exactly
    spec = rb.load_robot("franka")
    want_idx, want_d, want_g = orc.sdf_index(og, pos), orc.sdf_distance(og, pos), orc.sdf_gradient(og, pos)
    assert (want_g == 0.1).any()
    for layout in ("linear", "brick"):
        sc = _engine().DeviceScene(spec, (data, origin, delta), (0, 0, 0), layout=layout)
        idx, dist, grad = sc.sdf_query(torch.tensor(pos))
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), want_idx), layout
        assert np.array_equal(dist.cpu().numpy(), want_d.astype(np.float32)), layout
        assert np.array_equal(grad.cpu().numpy(), want_g.astype(np.float32)), layout
        if layout == "brick":
            # the summary is the minimum over each brick's existing voxels
            nb = [(s + 3) // 4 for s in shape]
            pad = np.full([4 * b for b in nb], np.inf)
            pad[:shape[0], :shape[1], :shape[2]] = data
            want_min = pad.reshape(nb[0], 4, nb[1], 4, nb[2], 4).min(axis=(1, 3, 5)).astype(np.float32)
            assert np.array_equal(sc.brick_min.cpu().numpy().reshape(nb), want_min)


def test_slab_upload_equals_whole_upload():
    """A grid packed in several slabs (halo rows re-sent) gives the same table as one call."""
    rng = np.random.default_rng(5)
    data = rng.normal(0.0, 0.3, (37, 20, 24))
    spec = rb.load_robot("franka")
    for layout in ("linear", "brick"):
        a = _engine().DeviceScene(spec, (data, np.zeros(3), 0.05), (0, 0, 0), layout=layout)
        b = _engine().DeviceScene(spec, (data, np.zeros(3), 0.05), (0, 0, 0), layout=layout, slab_bytes=8 * 8 * 20 * 24)
        assert torch.equal(a.table, b.table)
        if layout == "brick":
            assert torch.equal(a.brick_min, b.brick_min)


def test_synthetic14_log_prob_and_gradient():
    pb = synthetic_problem(dof=14, n_grid=48)
    spec = pb["spec"]
    assert spec.num_spheres == 45 and not spec.craig
    sc = _engine().DeviceScene(spec, pb["grid"], pb["offset"])
    rng = np.random.default_rng(2)
    g = rng.uniform(-2.5, 2.5, (4096, 14)).astype(np.float32)
    logp, dl = sc.log_prob(torch.tensor(g), want_grad=True)
    want_lp, want_dl = orc.log_prob(pb["scene"], g.astype(np.float64), want_grad=True)
    logp, dl = logp.cpu().numpy(), dl.cpu().numpy()
    assert (want_lp < 0).mean() > 0.2, "scene must put spheres inside the hinge band"
    ok = np.isclose(logp, want_lp, rtol=2e-4, atol=1e-5)
    assert ok.mean() > 0.99, f"only {ok.mean():.4f} of configurations agree"       # voxel flips are rare (45 spheres each)
    scale = np.abs(want_dl).max(axis=1, keepdims=True) + 1e-6
    okg = (np.abs(dl - want_dl) / scale).max(axis=1) < 5e-4
    assert (okg | ~ok).mean() > 0.99


def _batch(pb, sc, S, N, M, B, **kw):
    P = len(pb["ys"])
    pl = _engine().PlannerBatch(sc, pb["ys"], num_samples=S, num_inducing=M, num_data=N, num_bases=B,
                                lengthscales=[2.0] * pb["spec"].dof, variance=0.2, alpha=pb["alpha"],
                                learning_rate=pb["lr"], **kw)
    for k, p in enumerate(pb["params"]):
        pl.q_mu[k].copy_(torch.tensor(p.q_mu.T)); pl.q_sqrt[k].copy_(torch.tensor(p.q_sqrt))
        pl.raw_ell[k].copy_(torch.tensor(p.raw_ell)); pl.raw_var[k].copy_(torch.tensor(p.raw_var))
    nz = [_noise32(n) for n in pb["noise"]]
    st = lambda name: np.stack([getattr(n, name) for n in nz])
    pl.set_noise(st("omega"), st("beta"), st("w"), st("eps"), st("eps2"))
    return pl, nz


# (S, N, problems): 8 lanes per configuration (few configurations) / 1 lane (> 65 536 configurations in the launch);
# 5 problems = 70 latents of few samples: the few-sample prior kernel at 14 joints (projections by four MFMAs per tile, joint extent
# padded to 16), one and two 16-row sample tiles
@pytest.mark.parametrize("S,N,M,B,P", [(8, 12, 6, 64, 1), (40, 30, 10, 128, 2), (128, 100, 30, 256, 6), (8, 30, 10, 128, 5),
                                       (24, 30, 10, 128, 5)])
def test_synthetic14_elbo_forward_backward_against_oracle(S, N, M, B, P):
    pb = synthetic_problem(dof=14, S=S, N=N, M=M, B=B, seed=13, n_grid=48, n_problems=P)
    sc = _engine().DeviceScene(pb["spec"], pb["grid"], pb["offset"], free_space_summary=True)
    pl, nz = _batch(pb, sc, S, N, M, B)
    loss, grads = pl.loss_and_grad(generate=False)
    torch.cuda.synchronize()
    check = range(P) if S * N <= 2000 else (0, P - 1)      # the oracle takes seconds per full-size problem
    for k in check:
        p, y = pb["params"][k], pb["ys"][k]
        # the oracle looks its voxels up at the device's own float32 sphere centres (include/vgpmp_debug.h): fixed tolerances
        fw = orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], y, nz[k], pb["alpha"], lookup_pos=device_centres(pl, k))
        og, _ = orc.elbo_backward(p, pb["scene"], pb["X"], pb["Zy"], nz[k], pb["alpha"], fw)
        np.testing.assert_allclose(pl.f[k].cpu().numpy(), fw["f"], rtol=0, atol=1e-4)
        assert (fw["logp"] < 0).any()
        logp = pl.logp[k].cpu().numpy()
        tag = f"synthetic14[S={S},N={N},P={P},k={k}]"
        np.testing.assert_allclose(logp, fw["logp"], rtol=0, atol=TOL_LOGP * np.abs(fw["logp"]).max(), err_msg=tag)
        np.testing.assert_allclose(float(pl.kl[k]), fw["cv"]["kl"], rtol=1e-9)
        np.testing.assert_allclose(float(pl.lik[k]), fw["lik"], rtol=TOL_LIK)
        assert_grads(tag, grads, og, k=k)
        if k == 0:
            flipped_share(tag, logp, orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], y, nz[k], pb["alpha"], want_dell=False))


@pytest.mark.parametrize("S,N,P", [(8, 12, 1), (128, 100, 6)])
def test_layouts_and_summary_are_bitwise_equivalent(S, N, P):
    """Same inputs through {linear, brick, brick + free-space summary}: identical bits in f, logp, lik and every
    gradient (the summary only skips table reads whose hinge cost is exactly zero)."""
    M, B = 6, 64
    pb = synthetic_problem(dof=14, S=S, N=N, M=M, B=B, seed=17, n_grid=40, n_problems=P)
    outs = []
    # ... and the free-space masks in LDS (batch form, 9-15 joints): alone, with the summary behind them, with coarser blocks
    # (a budget that forces 8^3-voxel blocks on this 40^3 grid)
    for layout, summary, mask, budget in (("linear", False, False, 0), ("brick", False, False, 0), ("brick", True, False, 0),
                                          ("brick", False, True, 32 << 10), ("brick", True, True, 32 << 10),
                                          ("brick", False, True, 64), ("brick", True, True, 64)):
        sc = _engine().DeviceScene(pb["spec"], pb["grid"], pb["offset"], layout=layout, free_space_summary=summary,
                                   free_space_mask=mask, mask_budget_bytes=budget)
        assert sc.free_space_summary == summary and sc.free_space_mask == mask
        if mask:
            assert sc.mask_shift == {32 << 10: 2, 64: 3}[budget]
        pl, _ = _batch(pb, sc, S, N, M, B)
        if P > 1:
            pl.extra_flags |= _engine().capi.LIK_LANES      # (the batch form whatever the batch size: the one that reads the masks)
        loss, grads = pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        outs.append([pl.f.clone(), pl.logp.clone(), pl.lik.clone(), pl.view("G")] + [g.clone() for g in grads])
    assert float((outs[0][1] < 0).float().mean()) > 0.05, "hinge must be active"
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_free_space_masks_are_the_block_minima_against_the_clearances():
    """vgpmp_sdf_free_mask: bit b of mask k is set iff every voxel of block b lies at least clearance k from the obstacles
    (ragged extents; several radius classes), and the classes are epsilon + radius one float32 ulp up."""
    rng = np.random.default_rng(11)
    shape = (37, 22, 45)
    # a distance-like field (two balls) with a little noise: blocks far from both are free for every class, near ones for none
    g = np.stack(np.meshgrid(*[0.05 * np.arange(n) for n in shape], indexing="ij"), axis=-1)
    data = np.minimum(np.linalg.norm(g - [0.5, 0.4, 0.6], axis=-1) - 0.25, np.linalg.norm(g - [1.4, 0.8, 1.7], axis=-1) - 0.3)
    data = data + rng.normal(0.0, 0.004, shape)
    spec = rb.synthetic_arm(14)
    spec.sphere_radii = np.asarray([0.03, 0.05, 0.08] * 15, dtype=np.float64)
    for budget, shift in ((32 << 10, 2), (200, 3)):
        sc = _engine().DeviceScene(spec, (data, np.zeros(3), 0.05), (0, 0, 0), free_space_mask=True, mask_budget_bytes=budget)
        assert sc.mask_shift == shift and len(sc.mask_clearances) == 3
        e = 1 << shift
        nb = [(n + e - 1) // e for n in shape]
        pad = np.full([e * b for b in nb], np.inf)
        pad[:shape[0], :shape[1], :shape[2]] = data.astype(np.float32)
        bmin = pad.reshape(nb[0], e, nb[1], e, nb[2], e).min(axis=(1, 3, 5)).reshape(-1)
        words = sc.sdf.mask_words
        got = sc.free_mask.cpu().numpy().view(np.uint32).reshape(3, words)
        for k, (r, clr) in enumerate(zip((0.03, 0.05, 0.08), sc.mask_clearances)):
            assert np.float32(clr) == np.nextafter(np.float32(np.float32(0.05) + np.float32(r)), np.float32(np.inf))
            want = np.zeros(words * 32, dtype=bool)
            want[:bmin.size] = bmin >= np.float32(clr)
            bits = ((got[k][:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
            assert np.array_equal(bits, want), (shift, k)
            assert 0 < want.sum() < bmin.size


def test_mask_builder_stays_inside_its_words_on_ragged_word_counts():
    """mask_words is a multiple of 4 words, the builder's grid of 8 (256 threads): with mask_words % 8 == 4 the last
    workgroup's upper waves lie beyond the mask and must write nothing -- neither into the first words of the next mask nor
    past the caller's buffer (an exact-size C allocation).  Sentinel-filled buffer, two extents with that remainder."""
    import ctypes as C
    eng = _engine()
    capi = eng.capi
    rng = np.random.default_rng(5)
    for shape in ((16, 16, 8), (100, 100, 100)):
        g = np.stack(np.meshgrid(*[0.02 * np.arange(n) for n in shape], indexing="ij"), axis=-1)
        data = np.linalg.norm(g - 0.01 * np.asarray(shape), axis=-1) - 0.35 * 0.02 * min(shape) + rng.normal(0.0, 0.002, shape)
        spec = rb.synthetic_arm(14)
        spec.sphere_radii = np.asarray([0.03, 0.05, 0.08] * 15, dtype=np.float64)
        sc = eng.DeviceScene(spec, (data, np.zeros(3), 0.02), (0, 0, 0), free_space_mask=True, mask_budget_bytes=1 << 20)
        words = int(sc.sdf.mask_words)
        assert sc.mask_shift == 2 and words % 8 == 4
        want = sc.free_mask.cpu().numpy().copy().reshape(3, words)
        tail = 64
        buf = torch.full((3 * words + tail,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        full = capi.Sdf.from_buffer_copy(sc.sdf)
        full.brick_min, full.free_mask = capi.ptr(sc.brick_min), capi.ptr(buf)
        for _ in range(3):          # the lost bits of the unguarded form varied from run to run
            buf.fill_(0x5A5A5A5A)
            capi.check(sc.lib.vgpmp_sdf_free_mask(C.byref(full), sc._stream()), "vgpmp_sdf_free_mask")
            got = buf.cpu().numpy()
            assert np.all(got[3 * words:] == 0x5A5A5A5A), "wrote past the last mask"
            assert np.array_equal(got[:3 * words].reshape(3, words), want)
        # ... and the masks are the block minima (first words of masks 1, 2 included)
        e = 4
        nb = [(n + e - 1) // e for n in shape]
        pad = np.full([e * b for b in nb], np.inf)
        pad[:shape[0], :shape[1], :shape[2]] = data.astype(np.float32)
        bmin = pad.reshape(nb[0], e, nb[1], e, nb[2], e).min(axis=(1, 3, 5)).reshape(-1)
        for k, clr in enumerate(sc.mask_clearances):
            ref = np.zeros(words * 32, dtype=bool)
            ref[:bmin.size] = bmin >= np.float32(clr)
            bits = ((want[k].view(np.uint32)[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
            assert np.array_equal(bits, ref), (shape, k)


def test_masks_with_several_radius_classes_are_bitwise_neutral():
    """Spheres of three radii: every sphere uses the mask of the smallest clearance that covers it; outputs equal the
    mask-free form bit for bit (batch form, 14 joints)."""
    S, N, M, B, P = 64, 40, 6, 64, 3
    pb = synthetic_problem(dof=14, S=S, N=N, M=M, B=B, seed=19, n_grid=40, n_problems=P)
    pb["spec"].sphere_radii = np.asarray([0.03, 0.05, 0.08] * 15, dtype=np.float64)
    outs, free = [], []
    for mask in (False, True):
        sc = _engine().DeviceScene(pb["spec"], pb["grid"], pb["offset"], free_space_summary=False, free_space_mask=mask)
        pl, _ = _batch(pb, sc, S, N, M, B)
        pl.extra_flags |= _engine().capi.LIK_LANES
        loss, grads = pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        outs.append([pl.f.clone(), pl.logp.clone(), pl.lik.clone(), pl.view("G")] + [g.clone() for g in grads])
        if mask:
            bits = sc.free_mask.cpu().numpy().view(np.uint32)
            free.append(float(np.unpackbits(bits.view(np.uint8)).mean()))
    assert 0.02 < free[0] < 0.98, free
    assert float((outs[0][1] < 0).float().mean()) > 0.05, "hinge must be active"
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_config5_full_size_properties():
    """BASELINE config 5, one GPU's share at full size: 14-DoF arm, 512^3 voxels (2 GiB table, beyond the Infinity Cache),
    64 problems, S=128, M=30, N=100.  No oracle at this size: bitwise replay, layout / summary independence, finite
    outputs, KL >= 0, decreasing loss, and spot checks of table records against the analytic scene."""
    eng = _engine()
    n = 512
    spec = rb.synthetic_arm(14)
    rows = scenes.AnalyticSceneRows(n, 2.0 / n, (-1.0, -1.0, -1.0), seed=0, n_boxes=24, n_spheres=16, round_to=torch.float32)
    grid = (rows, rows.origin, rows.delta)
    rng = np.random.default_rng(0)
    qs = rng.uniform(-2.0, 2.0, (64, 2, 14))
    kw = dict(num_samples=128, num_inducing=30, num_data=100, num_bases=1024, lengthscales=[2.0] * 14, variance=0.2,
              learning_rate=0.02, seed=1)
    sc = eng.DeviceScene(spec, grid, (0, 0, 0))
    assert sc.layout == 1 and sc.free_space_summary and sc.table.numel() * 4 == 2 << 30
    assert sc.free_space_mask and sc.mask_shift == 3 and sc.free_mask.numel() * 4 == 32 << 10      # 64^3 bits of 8^3-voxel blocks
    # table records against the analytic distance at random voxels (values are float32 of the float64 scene)
    pos = rows.origin + rows.delta * rng.integers(0, n, (2000, 3))
    idx, dist, _ = sc.sdf_query(torch.tensor(pos + 0.25 * rows.delta))
    assert np.array_equal(idx.cpu().numpy(), np.round((pos - rows.origin) / rows.delta).astype(np.int32))
    xs = np.round((pos[:, 0] - rows.origin[0]) / rows.delta).astype(int)
    for k in range(0, 2000, 97):
        r = rows.rows(int(xs[k]), int(xs[k]) + 1, sc.device)[0]
        assert float(r[idx[k, 1], idx[k, 2]]) == float(dist[k])
    a = eng.PlannerBatch(sc, qs, **kw)
    b = eng.PlannerBatch(sc, qs, **kw)
    l0 = -a.elbo(step=10**6)
    for _ in range(6):
        a.step(); b.step()
    assert torch.equal(a.q_mu, b.q_mu) and torch.equal(a.q_sqrt, b.q_sqrt) and torch.equal(a.raw_ell, b.raw_ell)
    l1 = -a.elbo(step=10**6)
    assert bool(torch.isfinite(l0).all() and torch.isfinite(l1).all()) and bool((a.kl >= 0).all())
    assert float(l1.sum()) < float(l0.sum())
    assert float((a.logp < 0).float().mean()) > 0.05
    # the other table forms give the same bits from the same state and noise key
    ref = [a.f.clone(), a.logp.clone(), a.lik.clone()]
    del b
    for layout, summary, mask in (("brick", False, False), ("brick", True, False), ("brick", False, True), ("linear", False, False)):
        sc2 = eng.DeviceScene(spec, grid, (0, 0, 0), layout=layout, free_space_summary=summary, free_space_mask=mask)
        c = eng.PlannerBatch(sc2, qs, **kw)
        for name in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
            getattr(c, name).copy_(getattr(a, name))
        c.elbo(step=10**6)
        assert torch.equal(c.f, ref[0]) and torch.equal(c.logp, ref[1]) and torch.equal(c.lik, ref[2])
        del c, sc2


@pytest.mark.parametrize("robot", ["franka", "wam", "ur10", "kuka", "synthetic14", "synthetic15", "synthetic9"])
def test_batch_form_of_the_likelihood_against_oracle_on_every_robot_shape(robot):
    """The batch form (one lane per configuration, sphere gathers in batches across frames; per-frame sums in indexed registers --
    the pipelined form -- or, measurement flag, in LDS) forced on small problems of every robot shape: Craig and classic DH,
    6 / 7 / 9 / 14 / 15 joints, ragged sphere counts per frame.  Against the oracle, and the two state placements and the
    free-space summary against each other bit for bit."""
    from vgpmp_amd import capi
    from helpers import small_problem
    S, N, M, B = 9, 21, 6, 64        # S * N = 189: not a multiple of the workgroup
    if robot.startswith("synthetic"):
        pb = synthetic_problem(dof=int(robot[9:]), S=S, N=N, M=M, B=B, seed=23, n_grid=40, n_problems=2)
    else:
        sp = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=23, n_grid=40)
        pb = dict(sp, ys=np.stack([sp["y"], sp["y"][::-1]]), params=[sp["params"], sp["params"]], noise=[sp["noise"]] * 2)
    sc = _engine().DeviceScene(pb["spec"], pb["grid"], pb["offset"], free_space_summary=False)
    outs = {}
    for name, flag, summary in (("regs", capi.LIK_LANES, False), ("regs+summary", capi.LIK_LANES, True),
                                ("lds", capi.LIK_LDS_STATE, False), ("lds+summary", capi.LIK_LDS_STATE, True)):
        sc.sdf.brick_min = capi.ptr(sc.brick_min) if summary else None
        pl, nz = _batch(pb, sc, S, N, M, B)
        pl.extra_flags = flag
        pl.fuse = False                     # one launch per kernel: the likelihood launch is the form under test
        loss, grads = pl.loss_and_grad(generate=False)
        torch.cuda.synchronize()
        outs[name] = [pl.logp.clone(), pl.view("G"), pl.lik.clone()] + [g.clone() for g in grads]
    for k in range(2):
        # (`pl` is the last planner of the loop above: same inputs, same one-lane form, hence the same sphere centres)
        fw = orc.elbo_forward(pb["params"][k], pb["scene"], pb["X"], pb["Zy"], pb["ys"][k], nz[k], pb["alpha"],
                              lookup_pos=device_centres(pl, k))
        og, Gw = orc.elbo_backward(pb["params"][k], pb["scene"], pb["X"], pb["Zy"], nz[k], pb["alpha"], fw)
        assert (fw["logp"] < 0).any()
        np.testing.assert_allclose(outs["regs"][0][k].cpu().numpy(), fw["logp"], rtol=0, atol=TOL_LOGP * np.abs(fw["logp"]).max())
        np.testing.assert_allclose(float(outs["regs"][2][k]), fw["lik"], rtol=TOL_LIK)
        got = outs["regs"][3][k].cpu().numpy().T
        assert np.abs(got - og.q_mu).max() <= TOL_GRAD * np.abs(og.q_mu).max()
    # every form keeps the force / moment sums per frame (the prefix-scalar form of up to 8 joints, whose gradients agreed with these only
    # to float32 rounding, was retired in round 6: profiles/r06/flake.md): bit for bit
    for other in ("regs+summary", "lds", "lds+summary"):
        for a, b in zip(outs["regs"], outs[other]):
            assert torch.equal(a, b), other


@pytest.mark.parametrize("n,delta,origin,offset", [(512, 2.0 / 512, (-1.0, -1.0, -1.0), (0.0, 0.0, 0.0)),        # config 5: a power of two
                                                   (130, 0.0125, (-0.9125, -0.96, -0.49), (-0.2, 0.0, -0.2)),       # the industrial grid
                                                   (96, 0.025, (-1.2, -1.2, -0.6), (0.62, -0.15, 0.834)),           # bookshelves offset
                                                   (40, 0.03, (-0.4, 0.1, -1.0), (0.05, -0.03, 0.02))])
def test_kernel_index_path_is_the_reference_index_on_cell_boundaries(n, delta, origin, offset):
    """The ELBO kernels' own voxel index (float32 quotient; near a cell boundary the reference's float64 index WITHOUT the
    division, csrc/fk_sdf.hip::voxel_axis_near) against utils/sdf_utils.py:62-66 evaluated in float64 NumPy on the same
    float32 sphere centres: bit-exact on random points, on lattice points, and on the float32 neighbours (+- 1..3 ulps) of
    every kind of boundary -- where the quotient rounds up to an integer, where it stays just below one, at the clamps."""
    rng = np.random.default_rng(7)
    origin, offset = np.array(origin, dtype=np.float64), np.array(offset, dtype=np.float64)
    data = np.zeros((n, 8, 8))
    spec = rb.load_robot("franka")
    shape3 = np.array([n, 8, 8])
    sc = _engine().DeviceScene(spec, (data, origin, delta), offset, layout="brick")
    ext = delta * shape3
    rnd = (offset + origin + rng.uniform(-0.1, 1.1, (200000, 3)) * ext).astype(np.float32)
    k = rng.integers(-2, shape3 + 3, (60000, 3))
    lat = (offset + origin + delta * k).astype(np.float32)                         # the float32 nearest a boundary
    near = [np.nextafter(lat, np.float32(np.inf) * s) for s in (1, -1)]
    near2 = [np.nextafter(np.nextafter(lat, np.float32(np.inf) * s), np.float32(np.inf) * s) for s in (1, -1)]
    tiny = (lat.astype(np.float64) * (1 + rng.uniform(-3e-7, 3e-7, lat.shape))).astype(np.float32)
    pos = np.concatenate([rnd, lat] + near + near2 + [tiny])
    p64 = pos.astype(np.float64)
    q = ((p64 - offset) - origin) / delta                                          # sdf_utils.py:62-66, float64
    want = np.clip(np.trunc(q), 0, shape3 - 1).astype(np.int64)
    got = sc.sdf_index_f32(torch.tensor(pos)).cpu().numpy().astype(np.int64)
    assert (got[:, 0] >= 0).all(), "the two index forms of the kernels disagree"
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], pos[bad[:5]], got[bad[:5]], want[bad[:5]], q[bad[:5]])
    # the boundary cases were really there: quotients within 1e-9 of an integer on both sides, and exact integers
    frac = np.abs(q - np.rint(q))
    assert (frac == 0).sum() > 100 or delta != 2.0 / 512
    assert ((frac > 0) & (frac < 1e-6)).sum() > 1000
