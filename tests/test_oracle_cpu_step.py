"""The compiled float64 restatement (oracle/cpu_step.cpp: C++ / OpenMP, written from SURVEY Appendix A and the reference files it
cites) against the NumPy oracle (oracle/vgpmp_oracle.py) -- two independent restatements of the same optimisation step: loss,
every gradient and three-step Adam trajectories on small problems of three robots (Craig and classic DH, 6 and 7 joints).  Both
are test infrastructure; the GP half of either is unpinned against GPflow / TF (no such stack in the image), so their
agreement is what stands in for a second opinion there.  CPU only."""
import numpy as np
import pytest

from oracle import cpu_step
from oracle import vgpmp_oracle as orc
from helpers import small_problem


@pytest.mark.parametrize("robot,S,N,M,B", [("franka", 6, 9, 5, 64), ("wam", 10, 14, 7, 64), ("ur10", 16, 20, 6, 128),
                                           ("kuka", 8, 12, 7, 64)])
def test_compiled_step_against_numpy_oracle(robot, S, N, M, B):
    pb = small_problem(robot=robot, S=S, N=N, M=M, B=B, seed=11, n_grid=24)
    D = pb["spec"].dof
    p = pb["params"].copy()
    st = orc.adam_init(p)
    rng = np.random.default_rng(1)
    p.q_sqrt = np.tril(p.q_sqrt + 0.05 * rng.standard_normal(p.q_sqrt.shape))
    pr = cpu_step.Problem(pb["scene"], pb["X"], pb["Zy"], pb["y"], p, st)
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-300)
    active = False
    for step in range(3):
        nz = orc.draw_noise(rng, S, D, D, B, M + 2)
        fw = orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], pb["y"], nz, pb["alpha"])
        og, _ = orc.elbo_backward(p, pb["scene"], pb["X"], pb["Zy"], nz, pb["alpha"], fw)
        active = active or bool((fw["logp"] < 0).any())
        loss, g = pr.step(nz, pb["alpha"], pb["lr"], want_grad=True, threads=1 + step)      # (1, 2, 3 threads: the same numbers)
        orc.adam_step(p, og, st, pb["lr"], orc.DEFAULT_TRAINABLE)
        # measured: loss <= 3e-11, gradients <= 1.1e-10 of their largest entry (two float64 factorisations of a matrix of condition ~1e7)
        assert abs(loss + fw["elbo"]) <= 1e-9 * abs(fw["elbo"])
        for got, want in zip(g, (og.q_mu, og.q_sqrt, og.raw_ell, og.raw_var)):
            assert rel(got, want) < 1e-8
        # ... and the state after the Adam update (entries whose gradient is rounding noise may move by lr either way)
        for name in ("q_mu", "q_sqrt", "raw_ell", "raw_var"):
            gw = getattr(og, name)
            big = np.abs(gw) >= 1e-6 * np.abs(gw).max()
            assert np.abs(getattr(pr.p, name) - getattr(p, name))[big].max() < 1e-7 * pb["lr"] + 1e-12
    assert active, "the scene must put spheres inside the hinge band"
    assert pr.t == 3 == st.t


def test_gradient_only_call_leaves_the_state_alone_and_flags_gate_the_update():
    pb = small_problem(robot="franka", S=5, N=8, M=4, B=32, seed=3, n_grid=24)
    p = pb["params"].copy()
    rng = np.random.default_rng(2)
    nz = orc.draw_noise(rng, 5, 7, 7, 32, 6)
    pr = cpu_step.Problem(pb["scene"], pb["X"], pb["Zy"], pb["y"], p)
    before = [a.copy() for a in (pr.p.q_mu, pr.p.q_sqrt, pr.p.raw_ell, pr.p.raw_var)]
    pr.step(nz, pb["alpha"], pb["lr"], do_adam=False)
    assert all(np.array_equal(a, b) for a, b in zip(before, (pr.p.q_mu, pr.p.q_sqrt, pr.p.raw_ell, pr.p.raw_var))) and pr.t == 0
    pr.step(nz, pb["alpha"], pb["lr"], trainable=0b0101)                  # q_mu and lengthscales only
    assert not np.array_equal(before[0], pr.p.q_mu) and not np.array_equal(before[2], pr.p.raw_ell)
    assert np.array_equal(before[1], pr.p.q_sqrt) and np.array_equal(before[3], pr.p.raw_var)


def test_compiled_noise_draw_has_the_moments_of_the_oracles():
    nb = cpu_step.NoiseBuffers(64, 7, 7, 256, 12).draw(5, threads=2)
    a = nb.w.copy()
    assert np.array_equal(a, nb.draw(5, threads=1).w)                        # the same numbers whatever the thread count
    assert abs(nb.w.mean()) < 0.02 and abs(nb.w.std() - 1.0) < 0.02 and abs(nb.eps.std() - 1.0) < 0.05
    assert abs(nb.omega.var() - 5.0 / 3.0) < 0.35 and 0.0 <= nb.beta.min() and nb.beta.max() < 2 * np.pi      # Student-t(5): nu / (nu - 2)
