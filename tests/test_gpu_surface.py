"""GPU tests of the reference-shaped surface (gpflow_vgpmp.*), the driver flow of benchmarking.py,
sample-axis sharding, plan extraction and edge shapes.  All compute goes through the C ABI."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vgpmp_oracle as orc
from helpers import oracle_robot, oracle_scene, small_problem
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _env():
    from gpflow_vgpmp.utils.simulation_manager import SimulationManager
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return SimulationManager(file_path=ROOT / "parameters.yaml")


def test_driver_flow_like_benchmarking_py():
    """The call sequence of the reference's benchmarking.py:14-93 on two queries."""
    ns = {}
    exec("from gpflow_vgpmp.utils.miscellaneous import *", ns)
    gpflow, np_, p, solve = ns["gpflow"], ns["np"], ns["p"], ns["solve_planning_problem"]
    gpflow.config.set_default_float(np_.float64)
    env = _env()
    assert env.config["robot_params"]["robot_name"] == "franka"
    env.config["planner_params"].update(num_steps=60, num_samples=16, num_inducing=12, time_spacing_X=40, time_spacing_Xnew=60)
    queries = env.config["scene_params"]["queries"][:2]
    solved_total = 0
    for start, end in queries:
        start = np_.array(start, dtype=np_.float64).reshape(1, env.robot.dof)
        end = np_.array(end, dtype=np_.float64).reshape(1, env.robot.dof)
        env.robot.set_current_joint_config(np_.squeeze(start))
        env.robot.set_joint_motor_control(np_.squeeze(start), 300, 0.5)
        p.stepSimulation()
        solved, traj = solve(env=env, start_joints=start, end_joints=end)
        assert env.simulation.check_simulation_thread_health() is False      # as in the reference
        assert traj.shape == (60, 7) and np_.isfinite(traj).all()
        # pinned end points (1e-6 conditioning): path starts/ends at the query states
        assert np_.abs(traj[0] - start[0]).max() < 5e-2 and np_.abs(traj[-1] - end[0]).max() < 5e-2
        solved_total += bool(solved)
        p.removeAllUserDebugItems()
    env.simulation.stop_simulation_thread()
    assert 0 <= solved_total <= 2


def test_model_surface_against_oracle():
    from gpflow_vgpmp.models.vgpmp import VGPMP
    env = _env()
    ps = rb.load_problemset("franka", "industrial")
    y = np.array([ps.states[0], ps.states[1]])
    pp = dict(ps.planner_params, num_samples=8, num_inducing=6, num_bases=64)
    model = VGPMP.initialize(sdf=env.sdf, robot=env.robot, sampler=env.sampler, query_states=y,
                             scene_offset=env.scene.position, **pp)
    X = orc.init_trainset(12, 7)
    e = model.elbo(X)
    assert np.isfinite(e)
    # q_mu / q_sqrt properties (models/vgpmp.py:200-218) against the oracle
    osc = oracle_scene(env.robot.spec, env.sdf.grid, env.scene.position, sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    params = orc.init_params(osc.robot, y, 6, pp["lengthscales"], pp["variance"])
    cv = orc.cov_forward(params, X, orc.inducing_Zy(6, 7), orc.joint_sigmoid_inverse(osc.robot, y))
    np.testing.assert_allclose(model.q_sqrt, cv["C"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(model.q_mu, np.concatenate([orc.joint_sigmoid_inverse(osc.robot, y), params.q_mu]), rtol=1e-12)
    assert model._q_sqrt.shape == (7, 6, 6) and len(model.trainable_variables) == 4
    # likelihood.log_prob [S, N, D] -> [S, N]; sampler FK; SDF lookups (likelihood.py:57, sampler.py:103,216, sdf_utils.py:73)
    rng = np.random.default_rng(0)
    g = rng.uniform(osc.robot.low, osc.robot.high, (5, 9, 7)).astype(np.float32)
    lp = model.likelihood.log_prob(g).cpu().numpy()
    want = orc.log_prob(osc, g.astype(np.float64))
    assert lp.shape == (5, 9) and np.isclose(lp, want, rtol=2e-4, atol=1e-5).mean() > 0.95
    q = g[0, 0].astype(np.float64)
    np.testing.assert_allclose(env.sampler.forward_kinematics(q.reshape(7, 1)), orc.forward_kinematics(osc.robot, q), atol=5e-6)
    np.testing.assert_allclose(env.sampler.forward_kinematics_cost(q.reshape(7, 1)).cpu().numpy(),
                               orc.sphere_positions(osc.robot, q), atol=5e-6)
    pos = rng.uniform(-0.5, 0.5, (33, 3))
    assert np.array_equal(env.sdf.get_distance_tf(pos).cpu().numpy(), orc.sdf_distance(osc.sdf, pos).astype(np.float32))
    assert np.array_equal(env.sdf.get_distance_grad_tf(pos).cpu().numpy(), orc.sdf_gradient(osc.sdf, pos).astype(np.float32))
    # training and plan extraction run and keep the paths inside the joint limits
    from gpflow_vgpmp.utils.miscellaneous import training_loop, disable_param_opt
    disable_param_opt(model, env.config["trainable_params"])
    training_loop(model, X, 25)
    mu, best, samples, unc = model.sample_from_posterior(orc.init_trainset(20, 7), env.robot)
    assert mu.shape == (20, 7) and best.shape == (20, 7) and samples.shape == (7, 20, 7) and unc == 2.0
    assert (best >= osc.robot.low - 1e-6).all() and (best <= osc.robot.high + 1e-6).all()
    assert model.get_best_sample(samples) in range(7)
    # inducing locations as variables (utils/miscellaneous.py:338; reference default False): the device batch is rebuilt with
    # them, the trained values show up in the model's inducing variable, inside the Sigmoid(0.09, 0.91) bounds
    z0 = np.array(model.inducing_variable.inducing_variable.Zy[2:], copy=True)
    disable_param_opt(model, dict(env.config["trainable_params"], inducing_variable=True))
    training_loop(model, X, 10)
    z1 = model.inducing_variable.inducing_variable.Zy[2:]
    assert z1.shape == z0.shape and np.abs(z1 - z0).max() > 1e-3 and (z1 > 0.09).all() and (z1 < 0.91).all()
    assert np.array_equal(model.inducing_variable.inducing_variable.Zy[:2], np.stack([np.zeros(7), np.ones(7)]))
    mu_u, _, _, unc_u = model.sample_from_posterior(orc.init_trainset(20, 7), env.robot, compute_uncertainty=True)
    assert unc_u.shape == (20, 3) and (unc_u >= 0).all() and mu_u.shape == (20, 7)
    # sigma_obs / alpha as variables (reference default False): the device batch is rebuilt with them and the host
    # parameters follow the trained values
    disable_param_opt(model, dict(env.config["trainable_params"], alpha=True, sigma_obs=True))
    assert model.alpha.trainable and model.likelihood.variance.trainable
    a0, s0 = float(model.alpha), np.array(model.likelihood.variance.numpy(), copy=True)
    training_loop(model, X, 10)
    assert model._planner.lik_variables and float(model.alpha) != a0
    assert model.likelihood.variance.numpy().shape == s0.shape and not np.allclose(model.likelihood.variance.numpy(), s0)
    mu, best, samples, unc = model.sample_from_posterior(orc.init_trainset(20, 7), env.robot)
    assert np.isfinite(best).all()


def test_trained_inducing_locations_survive_a_change_of_time_stamps():
    """VGPMP._ensure rebuilds the device batch when the number of time stamps changes; the trained inducing
    locations (raw_Z), their Adam moments and the step count must move over with the other variables (the reference's
    variables do not depend on the data, models/vgpmp.py:29-42), and trainable_variables lists raw_Z."""
    from gpflow_vgpmp.models.vgpmp import VGPMP
    from gpflow_vgpmp.utils.miscellaneous import training_loop, disable_param_opt
    env = _env()
    ps = rb.load_problemset("franka", "industrial")
    y = np.array([ps.states[0], ps.states[1]])
    pp = dict(ps.planner_params, num_samples=8, num_inducing=6, num_bases=64)
    model = VGPMP.initialize(sdf=env.sdf, robot=env.robot, sampler=env.sampler, query_states=y,
                             scene_offset=env.scene.position, **pp)
    disable_param_opt(model, dict(env.config["trainable_params"], inducing_variable=True))
    training_loop(model, orc.init_trainset(12, 7), 8)
    old = model._planner
    assert old.z_variables and len(model.trainable_variables) == 5 and model.trainable_variables[-1] is old.raw_Z
    z_trained = old.inducing_locations().clone()
    assert float((z_trained[0, :, 0].cpu() - torch.linspace(0.1, 0.9, 6, dtype=torch.float64)).abs().max()) > 1e-3
    m, v, t = old.z_adam_m.clone(), old.z_adam_v.clone(), old.t
    assert np.isfinite(model.elbo(orc.init_trainset(15, 7)))            # another N: the batch is rebuilt
    new = model._planner
    assert new is not old and new.N == 15 and new.t == t
    assert torch.equal(new.inducing_locations(), z_trained)
    assert torch.equal(new.z_adam_m, m) and torch.equal(new.z_adam_v, v) and float(v.abs().max()) > 0
    assert torch.equal(new.q_mu, old.q_mu) and torch.equal(new.raw_ell, old.raw_ell)


def test_sample_sharding_two_ranks_equal_full_batch():
    """Two emulated ranks (sample_offset 0 / 8, KL on rank 0) sum to the 16-sample gradient: the device
    generator hands each rank its slice of the same global Philox sample stream."""
    from vgpmp_amd import engine
    S, N, M, B = 16, 11, 6, 64
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=4, n_grid=48)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    kw = dict(num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, alpha=pb["alpha"], seed=77,
              split_k=1)
    full = engine.PlannerBatch(sc, pb["y"][None], num_samples=S, **kw)
    r0 = engine.PlannerBatch(sc, pb["y"][None], num_samples=8, samples_total=S, sample_offset=0, kl_scale=1.0, **kw)
    r1 = engine.PlannerBatch(sc, pb["y"][None], num_samples=8, samples_total=S, sample_offset=8, kl_scale=0.0, **kw)
    lf, gf = full.loss_and_grad(step=3)
    l0, g0 = r0.loss_and_grad(step=3)
    l1, g1 = r1.loss_and_grad(step=3)
    assert torch.equal(full.w[0, :8], r0.w[0]) and torch.equal(full.w[0, 8:], r1.w[0])      # same global stream
    assert torch.equal(full.eps[0, 8:], r1.eps[0]) and torch.equal(full.omega, r1.omega)
    np.testing.assert_allclose(float(l0[0] + l1[0]), float(lf[0]), rtol=1e-5)
    for a, b, c in zip(gf, g0, g1):
        want, got = a[0].cpu().numpy(), (b[0] + c[0]).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-9
    assert float(r1.kl[0]) == 0.0 and float(r0.kl[0]) == float(full.kl[0])


def test_sample_slices_that_start_inside_a_philox_counter():
    """A rank's eps / eps' range need not start on a counter of the global stream (four normals per counter): Mz = 7, L = 7 and
    a sample offset of 3 put its first element at 147.  The rank draws whole counters and keeps its own elements."""
    from vgpmp_amd import engine
    S, N, M, B = 11, 9, 5, 32
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=5, n_grid=32)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    kw = dict(num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, alpha=pb["alpha"], seed=91, split_k=1)
    full = engine.PlannerBatch(sc, pb["y"][None], num_samples=S, **kw)
    r1 = engine.PlannerBatch(sc, pb["y"][None], num_samples=S - 3, samples_total=S, sample_offset=3, kl_scale=0.0, **kw)
    full.loss_and_grad(step=2); r1.loss_and_grad(step=2)
    torch.cuda.synchronize()
    assert (3 * (M + 2) * 7) % 4 != 0
    assert torch.equal(full.eps[0, 3:], r1.eps[0]) and torch.equal(full.eps2[0, 3:], r1.eps2[0])
    assert torch.equal(full.w[0, 3:], r1.w[0])


def test_latent_major_noise_copies_and_register_path_kernels_on_sample_slices():
    """Large batches draw eps / eps' in two layouts (rng_eps_t_body: [P,S,Mz,L] and, for the per-latent path kernels,
    [P,L,S,Mz]) and assemble the paths with a latent's operands in registers (paths_fwd_regs / paths_bwd_regs).  Both on a
    rank's slice of the samples: the copies are transposes of each other, the slices are the full stream's, and two ranks
    sum to the full batch."""
    from vgpmp_amd import engine
    P, S, N, M, B = 224, 32, 20, 30, 64
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=32, delta=0.08, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    kw = dict(num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, seed=13, split_k=1)
    full = engine.PlannerBatch(sc, qs, num_samples=S, **kw)
    r0 = engine.PlannerBatch(sc, qs, num_samples=S // 2, samples_total=S, sample_offset=0, kl_scale=1.0, **kw)
    r1 = engine.PlannerBatch(sc, qs, num_samples=S // 2, samples_total=S, sample_offset=S // 2, kl_scale=0.0, **kw)
    for pl in (full, r0, r1):
        pl.fuse = False
    lf, gf = full.loss_and_grad(step=3)
    l0, g0 = r0.loss_and_grad(step=3)
    l1, g1 = r1.loss_and_grad(step=3)
    torch.cuda.synchronize()
    Mz, L = M + 2, 7
    for pl in (full, r0, r1):
        for name, t in (("epsT", pl.eps), ("eps2T", pl.eps2)):
            tt = pl.view(name).reshape(P, L, pl.S, Mz)
            assert torch.equal(tt, t.reshape(P, pl.S, Mz, L).permute(0, 3, 1, 2))
    assert torch.equal(full.eps[:, S // 2:], r1.eps) and torch.equal(full.eps2[:, :S // 2], r0.eps2)
    assert torch.equal(full.f[:, S // 2:], r1.f) and torch.equal(full.f[:, :S // 2], r0.f)      # rows of a tile are independent
    np.testing.assert_allclose((l0 + l1).cpu().numpy(), lf.cpu().numpy(), rtol=1e-5)
    for a, b, c in zip(gf, g0, g1):
        want, got = a.cpu().numpy(), (b + c).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-9


@pytest.mark.parametrize("S,N,M,B", [(1, 1, 1, 16), (13, 37, 3, 48), (9, 150, 46, 32), (150, 100, 30, 64)])
def test_edge_shapes_against_oracle(S, N, M, B):
    """Ragged / extreme shapes: one sample, one time point, one inducing point, S and N that are not
    multiples of any tile, the largest supported inducing set (Mz = 48), the 150-path posterior draw."""
    from vgpmp_amd import engine
    pb = small_problem(robot="franka", S=S, N=N, M=M, B=B, seed=21, n_grid=32)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    pl = engine.PlannerBatch(sc, pb["y"][None], num_samples=S, num_inducing=M, num_data=N, num_bases=B,
                             lengthscales=[2.0] * 7, variance=0.2, alpha=pb["alpha"], split_k=1)
    p = pb["params"]
    pl.q_mu.copy_(torch.tensor(p.q_mu.T[None])); pl.q_sqrt.copy_(torch.tensor(p.q_sqrt[None]))
    pl.raw_ell.copy_(torch.tensor(p.raw_ell[None])); pl.raw_var.copy_(torch.tensor(p.raw_var[None]))
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    nz = pb["noise"]
    nz = orc.Noise(r32(nz.omega), r32(nz.beta), r32(nz.w), r32(nz.eps), r32(nz.eps2))
    pl.set_noise(nz.omega[None], nz.beta[None], nz.w[None], nz.eps[None], nz.eps2[None])
    loss, grads = pl.loss_and_grad(generate=False)
    fw = orc.elbo_forward(p, pb["scene"], pb["X"], pb["Zy"], pb["y"], nz, pb["alpha"])
    og, _ = orc.elbo_backward(p, pb["scene"], pb["X"], pb["Zy"], nz, pb["alpha"], fw)
    np.testing.assert_allclose(pl.f[0].cpu().numpy(), fw["f"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(float(pl.kl[0]), fw["cv"]["kl"], rtol=1e-8)
    ok = np.isclose(pl.logp[0].cpu().numpy(), fw["logp"], rtol=2e-3, atol=1e-4)
    assert ok.mean() >= 0.97
    if ok.all():
        np.testing.assert_allclose(float(loss[0]), -fw["elbo"], rtol=5e-4)
        for got, name in zip(grads, ("q_mu", "q_sqrt", "raw_ell", "raw_var")):
            want = getattr(og, name)
            got = got[0].cpu().numpy().T if name == "q_mu" else got[0].cpu().numpy()
            assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max() + 1e-9, name
    # forward-only entry reproduces the same ELBO pieces
    e = pl.elbo(generate=False)
    assert float(e[0]) == float(-loss[0])


def test_trainable_flags_freeze_parameters():
    from vgpmp_amd import engine
    pb = small_problem(robot="franka", S=8, N=10, M=5, B=32, seed=2, n_grid=32)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    pl = engine.PlannerBatch(sc, pb["y"][None], num_samples=8, num_inducing=5, num_data=10, num_bases=32,
                             lengthscales=[2.0] * 7, variance=0.2, trainable=dict(q_mu=True, q_sqrt=False,
                             lengthscales=False, kernel_variance=True))
    q0, s0, e0, v0 = pl.q_mu.clone(), pl.q_sqrt.clone(), pl.raw_ell.clone(), pl.raw_var.clone()
    for _ in range(3):
        pl.step()
    assert not torch.equal(pl.q_mu, q0) and not torch.equal(pl.raw_var, v0)
    assert torch.equal(pl.q_sqrt, s0) and torch.equal(pl.raw_ell, e0)
    # likelihood constants as variables: sigma_obs trained, alpha present but frozen
    pl = engine.PlannerBatch(sc, pb["y"][None], num_samples=8, num_inducing=5, num_data=10, num_bases=32,
                             lengthscales=[2.0] * 7, variance=0.2, trainable=dict(q_mu=True, q_sqrt=True, lengthscales=True,
                             kernel_variance=True, sigma_obs=True, alpha=False))
    a0, g0 = pl.raw_alpha.clone(), pl.raw_sigma.clone()
    for _ in range(3):
        pl.step()
    n = pb["spec"].num_spheres
    assert torch.equal(pl.raw_alpha, a0) and not torch.equal(pl.raw_sigma[:, :n], g0[:, :n])
    assert torch.equal(pl.raw_sigma[:, n:], g0[:, n:])
    assert float((pl.sigma_obs() - 0.005).abs().max()) > 1e-5 and float(pl.alphas()[0]) == pytest.approx(100.0)


def test_graph_replay_matches_eager_steps():
    from vgpmp_amd import engine
    pb = small_problem(robot="franka", S=8, N=10, M=5, B=32, seed=2, n_grid=32)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    kw = dict(num_samples=8, num_inducing=5, num_data=10, num_bases=32, lengthscales=[2.0] * 7, variance=0.2, seed=5)
    a, b = engine.PlannerBatch(sc, pb["y"][None], **kw), engine.PlannerBatch(sc, pb["y"][None], **kw)
    a.capture(unroll=3)                    # one eager step + capture
    a.run_steps(7)                         # two graph replays + one eager step
    b.run_steps(8)
    torch.cuda.synchronize()
    assert a.t == b.t == 8
    assert torch.allclose(a.q_mu, b.q_mu, rtol=0, atol=1e-12) and torch.allclose(a.raw_ell, b.raw_ell, rtol=0, atol=1e-12)


def test_graph_replay_of_the_merged_launch_schedule():
    """Six problems run the medium-batch schedule (cov_a | noise, cov_b | tiled GEMM, update | final): a captured
    hipGraph of its steps replays to the same parameters as plain launches."""
    from vgpmp_amd import engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i] for i in range(6)])
    kw = dict(num_samples=64, num_inducing=12, num_data=40, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=9)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    assert a.dims.split_k == 1
    a.capture(unroll=3)
    a.run_steps(7)
    b.run_steps(8)
    torch.cuda.synchronize()
    assert a.t == b.t == 8
    for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var)):
        assert torch.allclose(x, y, rtol=0, atol=1e-12), float((x - y).abs().max())


@pytest.mark.parametrize("P,M", [(1, 12), (3, 12), (1, 30), (2, 30), (1, 20), (6, 12), (30, 12), (40, 30), (112, 30), (110, 20)])
def test_pipelined_steps_equal_single_step_calls(P, M):
    """vgpmp_elbo_steps shares launches between independent kernels and runs the q_mu / q_sqrt update of step t
    next to the covariance / feature kernels of step t+1; the result must equal the same steps issued one call
    at a time with one launch per kernel."""
    from vgpmp_amd import engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    # 6 problems: the merged launches of the medium batches (cov_a | noise, cov_b | tiled GEMM with the tiles first, hyper |
    # final); 30: stage B of the covariance path inside the fused prior launch; 40: the large-batch schedule with its small
    # launches merged and the counter tick inside paths_fwd -- each against one launch per kernel; 112 / 110 problems (784 / 770
    # latents, Mz = 32 and the zero-padded Mz = 22): more latents than workgroup slots of the covariance launches.
    # M = 30 (Mz = 32): the update role on four column strips, the two-panel elimination and the whole-wave hyper-parameter
    # update of the shared launches against final_kernel / hyper_kernel; M = 20 (Mz = 22): the zero-padded forms
    kw = dict(num_samples=32, num_inducing=M, num_data=40, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=3)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.fuse = False
    few = P * spec.dof <= 64      # `a` runs the few-problem schedule: features stored, products by the stage-2 GEMM role
    if few:
        # one launch per kernel would form the prior draws of this many latents by the few-sample kernel (features inside the
        # GEMM, projections on the matrix cores, constant factors applied to the accumulators: other float32 roundings than
        # features_kernel + GEMM).  For the bitwise comparison `b` keeps the stored-feature form; the few-sample kernel (here its
        # two-tile form, 32 samples) is held against it to float32 trajectory tolerance below, and against the oracle in
        # test_gpu_parity / test_gpu_plans.
        from vgpmp_amd import capi
        b.extra_flags |= capi.GEMM_DIRECT
        c = engine.PlannerBatch(sc, qs, **kw)
        c.fuse = False
        c.run_steps(1)
    a.run_steps(25)
    for i in range(25):
        b.run_steps(1)
        if few and i == 0:      # the first step's paths: the same noise, the two forms of the prior draws
            d = float((c.f - b.f).abs().max())
            assert 0.0 < d < 2e-5 * float(b.f.abs().max()) + 1e-6, d
    torch.cuda.synchronize()
    assert a.t == b.t == 25
    for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var),
                 (a.adam_v[1], b.adam_v[1])):
        assert torch.allclose(x, y, rtol=0, atol=1e-11), float((x - y).abs().max())
    assert torch.equal(a.eps, b.eps) and torch.equal(a.w, b.w)
    assert torch.equal(a.f, b.f) and torch.equal(a.logp, b.logp)
    # (the reported likelihood is a float32 sum of logp by workgroup: the form that assembles its own paths -- one problem,
    #  Mz = 32 -- walks a sample's time points in tiles of sixteen, the other forms sixteen consecutive configurations)
    assert torch.allclose(a.lik, b.lik, rtol=1e-7) and torch.allclose(a.kl, b.kl, rtol=1e-12)


@pytest.mark.parametrize("P", [4, 8])
def test_shared_launches_without_k_slices_equal_one_launch_per_kernel(P):
    """128 samples and four or eight problems: whole-K GEMM tiles at the front of stage 2 (four problems) / the merged
    launches of the large-batch schedule (eight), the rows role with two tiles per workgroup.  The one-launch-per-kernel schedule
    forms its prior draws by the f16-split kernel at this size; with its float32 form (VGPMP_PRIOR_F32) the two schedules
    agree bit for bit."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    kw = dict(num_samples=128, num_inducing=30, num_data=40, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=3)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    assert a.dims.split_k == 1
    b.fuse = False
    b.extra_flags |= capi.PRIOR_F32
    if P > 4:
        a.extra_flags |= capi.PRIOR_F32      # (eight problems: the large-batch schedule, merged small launches -- float32 form too)
    a.run_steps(7); a.run_steps(5)
    b.run_steps(12)
    torch.cuda.synchronize()
    for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var), (a.adam_v[1], b.adam_v[1])):
        assert torch.equal(x, y), float((x - y).abs().max())
    assert torch.equal(a.f, b.f) and torch.equal(a.logp, b.logp)


def test_pipelined_steps_with_trainable_likelihood_constants():
    """sigma_obs / alpha among the variables: the chained schedule (their update rides between the reverse pass of
    step t and stage 1 of step t+1) equals one call per step with one launch per kernel."""
    from vgpmp_amd import engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i] for i in range(2)])
    tr = dict(q_mu=True, q_sqrt=True, lengthscales=True, kernel_variance=True, sigma_obs=True, alpha=True)
    kw = dict(num_samples=32, num_inducing=12, num_data=40, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=3,
              alpha=4.0, trainable=tr)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.fuse = False
    from vgpmp_amd import capi
    b.extra_flags |= capi.GEMM_DIRECT      # (features_kernel + GEMM like `a`: see test_pipelined_steps_equal_single_step_calls)
    a.run_steps(12)
    for _ in range(12):
        b.run_steps(1)
    torch.cuda.synchronize()
    n = spec.num_spheres
    for x, y in ((a.q_mu, b.q_mu), (a.raw_ell, b.raw_ell), (a.raw_alpha, b.raw_alpha), (a.raw_sigma[:, :n], b.raw_sigma[:, :n]),
                 (a.lik_adam_v[1][:, :n], b.lik_adam_v[1][:, :n])):
        assert torch.allclose(x, y, rtol=0, atol=1e-11), float((x - y).abs().max())
    assert torch.allclose(a.lik, b.lik, rtol=1e-12)
    assert float((a.raw_alpha - a.raw_alpha[0]).abs().max()) > 0 or float(a.lik.abs().max()) == 0      # problems differ
    moved = (a.raw_sigma[:, :n] - engine.torch.tensor(engine.softplus_inverse(np.full(n, 0.005) - engine.SIGMA_FLOOR),
                                                      device=a.raw_sigma.device)).abs()
    assert float(moved.min()) > 0.05


@pytest.mark.parametrize("extra", [dict(sigma_obs=True, alpha=True), dict(inducing_variable=True), dict()])
def test_reset_is_a_freshly_built_planner(extra):
    """PlannerBatch.reset() (bench.py calls it before every timed block) must leave what a NEW planner holds: every variable
    the planner trains -- likelihood constants and inducing locations included --, their Adam moments and the step count.
    k steps after reset() == k steps of a new planner, bit for bit (raw_alpha / raw_sigma / raw_Z were kept)."""
    from vgpmp_amd import engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i] for i in range(2)])
    tr = dict(engine.DEFAULT_TRAINABLE, **extra)
    kw = dict(num_samples=16, num_inducing=10, num_data=24, num_bases=64, lengthscales=[2.0] * 7, variance=0.2, seed=3,
              alpha=4.0, trainable=tr)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    a.run_steps(7)                      # trains everything, moments non-zero
    torch.cuda.synchronize()
    assert not torch.equal(a.q_mu, b.q_mu)
    a.reset()
    assert a.t == 0
    for x, y in zip(a._variables() + a._moments(), b._variables() + b._moments()):
        assert torch.equal(x, y)
    a.run_steps(5)
    b.run_steps(5)
    torch.cuda.synchronize()
    for x, y in zip(a._variables() + a._moments(), b._variables() + b._moments()):
        assert torch.equal(x, y), float((x - y).abs().max())
    assert torch.equal(a.f, b.f) and torch.equal(a.lik, b.lik) and torch.equal(a.kl, b.kl)



@pytest.mark.parametrize("S,M,N,P", [(64, 30, 40, 12), (7, 24, 70, 20)])
def test_merged_launches_of_the_batch_schedule_change_nothing(S, M, N, P):
    """Batches that draw their own noise merge what a dependency level allows into one launch: the previous step's updates beside
    stage A and the draws (mid_stage1), the rows of A as stage A's tail, stage B behind the tiles of the prior kernel
    (prior_split_cov_b_kernel; with few samples mid_cov_b_prior16_kernel).  VGPMP_NO_FUSE runs one launch per kernel in sequence: the
    same kernels' arithmetic on the same inputs -- every variable, moment and the paths bit for bit, over a run long enough for a
    missing dependency inside a merged launch to show."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % 36] for i in range(P)])          # 84 / 140 (problem, latent) pairs: the large-batch schedule
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * 7, variance=0.2, seed=4)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.extra_flags |= capi.NO_FUSE
    for _ in range(3):
        a.run_steps(40); b.run_steps(40)
        a.step(); b.step()
    torch.cuda.synchronize()
    a.step()
    merged = [k for k in capi.last_schedule(a.lib) if "prior_split_cov_b_kernel" in k or "mid_cov_b_prior16_kernel" in k]
    b.step()
    apart = [k for k in capi.last_schedule(b.lib) if "cov_b_kernel<" in k and "prior" not in k]
    assert merged and apart, (capi.last_schedule(a.lib), capi.last_schedule(b.lib))      # (the two schedules this test is about)
    torch.cuda.synchronize()
    for x, y in zip(a._variables() + a._moments() + [a.f, a.lik, a.kl], b._variables() + b._moments() + [b.f, b.lik, b.kl]):
        assert torch.equal(x, y), float((x - y).abs().max())

def test_noise_drawn_ahead_is_never_paired_with_another_step():
    """The sample-sharded step lets step t draw the prior noise of step t + 1 (VGPMP_NOISE_AHEAD / _READY).  Which step's
    draws the buffers hold is tracked by the planner: a direct call in between that redraws them (elbo at another step)
    must make the next sharded step draw its own -- the trajectory equals the uninterrupted one."""
    from vgpmp_amd import engine, sharding
    ps = rb.load_problemset("ur10", "industrial")
    spec = rb.load_robot("ur10", *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    kw = dict(num_samples=32, num_inducing=12, num_data=40, num_bases=128, lengthscales=[2.0] * 6, variance=0.2, seed=9)
    runs = []
    for interrupt in (False, True):
        pl = engine.PlannerBatch(sc, np.array([ps.queries[0]]), **kw)
        sp = sharding.SampleShardedPlanner(pl)
        sp._allreduce = lambda buf=None: None
        for k in range(6):
            if interrupt and k == 3:
                pl.elbo(generate=True, step=40)        # leaves step 40's omega / beta / w in the shared buffers
                assert pl.noise_ahead_step is None
            sp.step()
            assert pl.noise_ahead_step == pl.t
        torch.cuda.synchronize()
        runs.append([t.clone() for t in (pl.q_mu, pl.q_sqrt, pl.raw_ell, pl.raw_var)])
    for x, y in zip(*runs):
        assert torch.equal(x, y)


def test_sharded_loop_in_one_call_equals_the_python_loop():
    """vgpmp_elbo_steps_reduced (step, all-reduce, Adam enqueued from C; here one rank, nothing to exchange) against
    SampleShardedPlanner.step() called from Python: bit for bit, also when the two are mixed."""
    from vgpmp_amd import engine, sharding
    ps = rb.load_problemset("ur10", "industrial")
    spec = rb.load_robot("ur10", *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    kw = dict(num_samples=64, num_inducing=12, num_data=40, num_bases=128, lengthscales=[2.0] * 6, variance=0.2, seed=9)
    out = []
    for c_loop in (False, True):
        pl = engine.PlannerBatch(sc, np.array([ps.queries[0]]), **kw)
        sp = sharding.SampleShardedPlanner(pl)
        sp._allreduce = lambda buf=None: None
        sp._single_rank = c_loop
        sp.run_steps(5)
        sp.step()                       # (a Python step in between: the noise-ahead chain carries over)
        sp.run_steps(4)
        torch.cuda.synchronize()
        assert pl.t == 10
        out.append([t.clone() for t in (pl.q_mu, pl.q_sqrt, pl.raw_ell, pl.raw_var, pl.adam_m[0], pl.adam_v[1], pl.f)])
    for x, y in zip(*out):
        assert torch.equal(x, y), float((x - y).abs().max())


def test_elimination_forms_agree():
    """The factorisation of Kuu + jI runs on one wave with the augmented matrix in registers (Mz <= 32); the
    workgroup-wide form through LDS applies the same multipliers with the operands associated differently."""
    from vgpmp_amd import capi, engine
    pb = small_problem(robot="franka", S=8, N=12, M=30, B=64, seed=2, n_grid=32)
    sc = engine.DeviceScene(pb["spec"], pb["grid"], pb["offset"])
    kw = dict(num_samples=8, num_inducing=30, num_data=12, num_bases=64, lengthscales=[2.0] * 7, variance=0.2, seed=5)
    a, b = engine.PlannerBatch(sc, pb["y"][None], **kw), engine.PlannerBatch(sc, pb["y"][None], **kw)
    b.extra_flags = capi.ELIM_BLOCK
    la, ga = a.loss_and_grad(step=1)
    lb, gb = b.loss_and_grad(step=1)
    torch.cuda.synchronize()
    # Kuu + 1e-6 I has condition number ~1e7: the two forms agree to ~1e-9 of the factor's scale
    for name in ("Kinv", "A4"):
        x, y = a.view(name).double(), b.view(name).double()
        assert float((x - y).abs().max()) <= 1e-6 * float(y.abs().max()), name
    np.testing.assert_allclose(float(la[0]), float(lb[0]), rtol=1e-6)
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 1e-4 * (float(y.abs().max()) + 1e-30)


def test_split_path_kernels_equal_one_workgroup_form():
    """Small launches run the path assembly and its reverse on two workgroups per (sample chunk, latent)
    (paths_fwd_split_body: halves of the time axis; paths_bwd_split: halves of the inducing axis).  The forward
    form keeps the arithmetic order, so f is bit-identical; the reverse form changes only float32 summation order."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[0]])
    # Mz = 16 (multiple of 8), N = 40 (multiple of 4), split-K 4: both split kernels are eligible
    kw = dict(num_samples=32, num_inducing=14, num_data=40, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=5)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    assert a.dims.split_k == 4
    b.extra_flags = capi.NO_SPLIT
    la, ga = a.loss_and_grad(step=2)
    lb, gb = b.loss_and_grad(step=2)
    torch.cuda.synchronize()
    assert torch.equal(a.f, b.f)
    np.testing.assert_allclose(float(la[0]), float(lb[0]), rtol=1e-12)
    for x, y in zip(ga, gb):
        scale = float(y.abs().max()) + 1e-30
        assert float((x - y).abs().max()) <= 2e-5 * scale, (float((x - y).abs().max()), scale)
    a.run_steps(10)
    b.run_steps(10)
    torch.cuda.synchronize()
    # Adam normalises every gradient entry by its own running magnitude: entries whose gradient is ~0 amplify the
    # float32 summation-order differences, hence the loose bound on the largest entry and the tight one on the mean
    for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var)):
        d = (x - y).abs()
        assert float(d.max()) < 2e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))


def test_large_batch_kernels_match_small_launch_kernels():
    """Large batches (five problems up) use the fused f16-split prior kernel (split_k = 1) and one lane per configuration in the
    likelihood; each problem must still equal the same problem evaluated alone (split-K GEMM, 8 lanes per
    configuration) up to float32 summation order."""
    from vgpmp_amd import engine
    S, N, M, B = 128, 60, 7, 64
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i] for i in range(10)])
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, seed=11)
    batch = engine.PlannerBatch(sc, qs, split_k=1, **kw)          # 10 * 128 * 60 configurations -> 1 lane each
    lb, gb = batch.loss_and_grad(step=5)
    for p in (0, 4, 9):
        solo = engine.PlannerBatch(sc, qs[p:p + 1], problem_base=p, **kw)
        assert solo.dims.split_k > 1
        ls, gs = solo.loss_and_grad(step=5)
        # (the same noise streams; the batch's w is never stored -- its prior kernel draws the weights where it uses them)
        assert torch.equal(solo.eps[0], batch.eps[p]) and torch.equal(solo.omega[0], batch.omega[p])
        np.testing.assert_allclose(float(ls[0]), float(lb[p]), rtol=2e-5)
        for a, b in zip(gs, gb):
            a, b = a[0].cpu().numpy(), b[p].cpu().numpy()
            assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max() + 1e-9


def test_batched_solve_of_a_problem_set():
    """All C(9,2) = 36 Franka/industrial queries as one device batch; every path is pinned to its query."""
    from gpflow_vgpmp.utils.miscellaneous import solve_planning_problems_batched
    env = _env()
    env.config["planner_params"].update(num_steps=40, num_samples=8, num_inducing=10, time_spacing_X=30, time_spacing_Xnew=40)
    queries = env.config["scene_params"]["queries"]
    info = {}
    out = solve_planning_problems_batched(env, queries, report=info)
    assert len(out) == 36
    low, high = env.robot.spec.low, env.robot.spec.high
    for k, ((solved, traj), (a, b)) in enumerate(zip(out, queries)):
        assert traj.shape == (40, 7) and np.isfinite(traj).all()
        assert np.abs(traj[0] - np.array(a)).max() < 5e-2 and np.abs(traj[-1] - np.array(b)).max() < 5e-2
        # the flag means something: the best sample clears every obstacle at every time point and respects the joint limits
        inside = bool(((traj >= low - 1e-9) & (traj <= high + 1e-9)).all())
        assert solved == (info["best_sample"][k] > 0.0 and inside)
        # ... and a query whose own start or goal state touches the obstacles (sphere model) cannot be solved
        if min(info["start"][k], info["goal"][k]) < -1e-3:
            assert not solved
        # 40 steps with 8 samples (a fifth of a plan, reduced sizes) do not leave a path much worse than the straight line it started
        # from: 15 mm of slack -- which query is how far along after 40 steps goes with the noise stream (tests/test_gpu_plans.py
        # holds "never worse than it began" at the reference's own planner parameters over whole plans)
        assert info["best_sample"][k] >= info["initial_path"][k] - 1.5e-2, (k, info["best_sample"][k], info["initial_path"][k])
    # (tests/test_gpu_plans.py runs the set at the reference's own planner parameters and counts)


@pytest.mark.parametrize("robot,problem,S,M,N,P", [("wam", "industrial", 50, 10, 70, 1),        # BASELINE config 1 shape
                                                    ("franka", "bookshelves", 7, 24, 70, 55),     # config 3: full C(11,2) batch
                                                    ("ur10", "industrial", 128, 18, 70, 1)])      # config 4: one rank's sample shard
def test_baseline_config_shapes_properties(robot, problem, S, M, N, P):
    """BASELINE configs at full size: size-independent properties (bitwise replay, finite, KL >= 0,
    paths inside the joint limits and pinned to start/goal, loss decreases)."""
    from vgpmp_amd import engine
    ps = rb.load_problemset(robot, problem)
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=96, delta=2.4 / 96, origin=(-1.2, -1.2, -0.6), seed=1)
    pp = ps.planner_params
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    qs = np.array(ps.queries[:P])
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=1024, lengthscales=pp["lengthscales"],
              variance=pp["variance"], alpha=pp["alpha"], learning_rate=pp["learning_rate"], seed=3)
    if robot == "ur10":
        kw.update(samples_total=1024, sample_offset=256, kl_scale=0.0)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    l0 = -a.elbo(step=10**6).clone()
    for _ in range(25):
        a.step(); b.step()
    assert torch.equal(a.q_mu, b.q_mu) and torch.equal(a.q_sqrt, b.q_sqrt) and torch.equal(a.raw_var, b.raw_var)
    l1 = -a.elbo(step=10**6)
    assert torch.isfinite(l0).all() and torch.isfinite(l1).all() and bool((a.kl >= 0).all())
    assert float(l1.mean()) < float(l0.mean())
    g = a.samples()
    lo, hi = torch.tensor(spec.low, device=g.device), torch.tensor(spec.high, device=g.device)
    assert bool(((g >= lo) & (g <= hi)).all())
    if max(pp["lengthscales"]) <= 4.0:
        # the 1e-6 "conditioning" pins the end points only while Kuu >> jitter I; with UR10's lengthscale 6 the
        # reference's own formula gives k(0, Z)(Kuu + jI)^-1 = (0.59, 0.02, 0.30, ...) rather than e_0
        y = torch.tensor(qs, device=g.device, dtype=g.dtype)
        assert float((g[:, :, 0, :] - y[:, None, 0, :]).abs().max()) < 1e-1
        assert float((g[:, :, -1, :] - y[:, None, 1, :]).abs().max()) < 1e-1


def test_mesh_sdf_matches_oracle_and_analytic_shapes(tmp_path):
    """vgpmp_mesh_sdf (SURVEY f-2): exact against the analytic SDF of a cube mesh, float64-close to the NumPy
    restatement on a real scene mesh, and the text file round-trips through the reference-format parser."""
    V = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)])
    F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6],
                  [0, 6, 4], [1, 5, 7], [1, 7, 3]])
    tri = V[F].reshape(-1, 9)
    data, origin, delta = scenes.mesh_sdf(tri, np.zeros(12, dtype=np.int32), delta=0.11, padding=4)
    pts = scenes.lattice(data.shape, origin, delta)
    q = np.abs(pts) - 0.5
    want = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)
    np.testing.assert_allclose(data, want, atol=1e-12)
    assert (data < 0).any() and (data > 0).any()
    tri, part = scenes.load_scene_mesh("boxes")
    data, origin, delta = scenes.mesh_sdf(tri, part, delta=0.05, padding=3)
    pts = scenes.lattice(data.shape, origin, delta)
    Vb, Fb = tri.reshape(-1, 3), np.arange(tri.shape[0] * 3).reshape(-1, 3)
    want = orc.mesh_signed_distance(Vb, Fb, part, pts)
    np.testing.assert_allclose(data, want, rtol=1e-9, atol=1e-12)
    scenes.write_sdf(str(tmp_path / "boxes.sdf"), (data, origin, delta))
    back = orc.parse_sdf_text(str(tmp_path / "boxes.sdf"))
    assert np.array_equal(back.data, data) and back.delta == delta


def test_industrial_scene_from_mesh_plans():
    """End to end on the reference's industrial scene geometry (grid generated from its collision mesh)."""
    from vgpmp_amd import engine
    grid = scenes.scene_sdf("industrial", delta=0.02, padding=15)
    assert grid[0].min() < 0 < grid[0].max()
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka", *ps.robot_pos_and_orn)
    pp = ps.planner_params
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0], sigma_obs=pp["sigma_obs"], epsilon=pp["epsilon"])
    pl = engine.PlannerBatch(sc, np.array(ps.queries[:6]), num_samples=16, num_inducing=10, num_data=50,
                             lengthscales=pp["lengthscales"], variance=pp["variance"], alpha=pp["alpha"],
                             learning_rate=pp["learning_rate"], seed=0)
    l0 = -pl.elbo(step=10**6).clone()
    pl.run_steps(60)
    l1 = -pl.elbo(step=10**6)
    assert torch.isfinite(l1).all() and float(l1.mean()) < float(l0.mean())
    _, best, _, _ = pl.sample_from_posterior(50, None)
    assert torch.isfinite(pl.path_clearance(best)).all()


def test_dispatchers_on_the_device_match_oracle():
    """gpflow_vgpmp.covariances.Kuu / Kuf / Kfu, kernel_conditioning.K_conditioned, kernels' __call__ and
    kullback_leiblers.prior_kl: every number from libvgpmp_hip.so (vgpmp_cov_matrices; the covariance stage of the ELBO step
    for the KL), against the oracle."""
    from gpflow_vgpmp.kernel_conditioning import K_conditioned
    from gpflow_vgpmp.covariances import Kuu, Kuf, Kfu
    from gpflow_vgpmp.kullback_leiblers.prior_kl import prior_kl
    from gpflow_vgpmp.inducing_variables.inducing_variables import (ConditionedVariableInducingPoints,
                                                                    SharedIndependentInducingVariables)
    from gpflow_vgpmp.kernels.kernels import Matern52, VanillaConditioningSeparateIndependent
    from oracle import vgpmp_oracle as orc
    L, M, N = 3, 5, 6
    Z = np.tile(np.linspace(0.1, 0.9, M)[:, None], (1, L))
    iv = SharedIndependentInducingVariables(ConditionedVariableInducingPoints(Z, np.stack([np.zeros(L), np.ones(L)])))
    ell, var = [2.0, 3.0, 0.7], 0.3
    kern = VanillaConditioningSeparateIndependent([Matern52(e, var) for e in ell])
    X = orc.init_trainset(N, L)
    Zy = orc.inducing_Zy(M, L)
    K = Kuu(iv, kern, jitter=1e-6).numpy()
    for l in range(L):
        np.testing.assert_allclose(K[l], orc.matern52(Zy[:, l], Zy[:, l], ell[l], var) + 1e-6 * np.eye(M + 2), rtol=1e-12)
        np.testing.assert_allclose(Kuf(iv, kern, X)[l].numpy(), orc.matern52(Zy[:, l], X[:, l], ell[l], var), rtol=1e-12)
    assert Kfu(iv, kern, X).shape == (L, N, M + 2)
    rng = np.random.default_rng(0)
    p = orc.Params(q_mu=rng.standard_normal((M, L)), q_sqrt=np.tril(rng.standard_normal((L, M, M))) + 2 * np.eye(M),
                   raw_ell=orc.softplus_inverse(np.array(ell)), raw_var=np.full(L, orc.softplus_inverse(var - 0.1)))
    y_u = rng.standard_normal((2, L))
    cv = orc.cov_forward(p, X, Zy, y_u)
    np.testing.assert_allclose(float(prior_kl(iv, kern, p.q_mu, p.q_sqrt, y_u)), cv["kl"], rtol=1e-9)
    np.testing.assert_allclose(K_conditioned(iv, X, kern).numpy(), Kuf(iv, kern, X).numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(kern.kernels[1](Zy[:, 1], X[:, 1]).numpy(), orc.matern52(Zy[:, 1], X[:, 1], ell[1], var), rtol=1e-12)
    # a second geometry through the same path: more inducing points, other hyper-parameters
    M2 = 30
    Z2 = np.tile(np.linspace(0.1, 0.9, M2)[:, None], (1, L))
    iv2 = SharedIndependentInducingVariables(ConditionedVariableInducingPoints(Z2, np.stack([np.zeros(L), np.ones(L)])))
    p2 = orc.Params(q_mu=rng.standard_normal((M2, L)), q_sqrt=np.tril(0.1 * rng.standard_normal((L, M2, M2))) + np.eye(M2),
                    raw_ell=p.raw_ell, raw_var=p.raw_var)
    cv2 = orc.cov_forward(p2, X, orc.inducing_Zy(M2, L), y_u)
    np.testing.assert_allclose(float(prior_kl(iv2, kern, p2.q_mu, p2.q_sqrt, y_u)), cv2["kl"], rtol=1e-8)


@pytest.mark.parametrize("S,N,M,B,lengthscales,P", [(37, 50, 10, 256, True, 3), (128, 150, 30, 1024, True, 3), (70, 20, 5, 64, False, 3),
                                                      (70, 20, 5, 64, True, 63), (128, 12, 5, 128, False, 64)])   # 128-row tiles (ragged / full)
def test_fused_prior_kernel_equals_generator_features_and_gemm(S, N, M, B, lengthscales, P):
    """Large-batch schedule with device-generated noise: W and Phi / dPhi formed inside the GEMM (prior_fused_batch_kernel,
    float32 MFMAs: flag PRIOR_F32) against the three launches it replaces (Philox generator -> w, features_kernel -> Phi / dPhi,
    tiled GEMM): the same expressions in the same order, so the prior draws, the paths, the likelihood and every gradient agree
    bit for bit.  Ragged sample count, one and two column tiles, with and without the lengthscale tangent; 63 / 64 problems: the
    128-row tiles of the full-chip regime."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    tr = dict(q_mu=True, q_sqrt=True, lengthscales=lengthscales, kernel_variance=True)
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * 7, variance=0.2, seed=9, split_k=1,
              trainable=tr, problem_base=5)
    outs = []
    for flag in (capi.PRIOR_F32, capi.NO_FUSE_PRIOR):
        pl = engine.PlannerBatch(sc, qs, **kw)
        pl.fuse = False                       # one launch per kernel: the large-batch schedule
        pl.extra_flags = flag
        pl.step(); pl.step()
        loss, grads = pl.loss_and_grad(generate=True, step=7)
        torch.cuda.synchronize()
        keep = [g.clone() for k, g in enumerate(grads) if lengthscales or k != 2]      # no lengthscale tangent: that gradient is not formed
        outs.append([pl.view("F0"), pl.view("H") if lengthscales else pl.view("F0"), pl.f.clone(), pl.logp.clone(), loss.clone(),
                     pl.q_mu.clone(), pl.raw_ell.clone()] + keep)
    assert float(outs[0][0].abs().max()) > 0
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("robot,S,N,M,B,lengthscales,P", [("franka", 37, 50, 10, 256, True, 3), ("franka", 128, 100, 30, 1024, True, 5),
                                                            ("franka", 70, 20, 5, 64, False, 3), ("franka", 64, 33, 6, 128, True, 2),
                                                            ("synthetic14", 128, 100, 30, 1024, True, 2), ("synthetic14", 50, 12, 5, 64, True, 3)])
def test_f16_split_prior_kernel_against_float64_and_the_float32_kernels(robot, S, N, M, B, lengthscales, P):
    """prior_fused_split_kernel (csrc/gp_prior_split.h, the default of the large-batch schedule): F0 = W Phi^T and
    H = W (dPhi / d ell)^T with every float32 operand split into two f16 halves and three f16 MFMAs per product, float32
    accumulators (restates the `temporary_paths` draw, models/vgpmp.py:281-282).  Gate: its distance from
    a float64 evaluation of the SAME W, omega, beta (the generator's own values, read back from the three-launch form) must
    not exceed that of the float32-MFMA kernels by more than rounding noise, and stays inside the 2e-5 max|.| the oracle
    parity tests allow; paths, likelihood and gradients follow.  7 joints (8-wide padding) and 14 joints (16-wide), 64- and
    128-row tiles, ragged S, with and without the lengthscale tangent."""
    from vgpmp_amd import capi, engine
    if robot == "franka":
        ps = rb.load_problemset("franka", "industrial")
        spec = rb.load_robot("franka")
        qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
        off = ps.object_positions[0]
    else:
        spec = rb.synthetic_arm(14)
        qs = np.random.default_rng(3).uniform(-2.0, 2.0, (P, 2, 14))
        off = (0.05, -0.03, 0.02)
    L = spec.dof
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, off)
    tr = dict(q_mu=True, q_sqrt=True, lengthscales=lengthscales, kernel_variance=True)
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=B, lengthscales=[2.0] * L, variance=0.2, seed=9, split_k=1,
              trainable=tr, problem_base=5)
    J = N + M + 2
    res = {}
    for name, flag in (("split", 0), ("f32", capi.PRIOR_F32), ("three", capi.NO_FUSE_PRIOR)):
        pl = engine.PlannerBatch(sc, qs, **kw)
        pl.fuse = False                       # one launch per kernel: the large-batch schedule
        pl.extra_flags = flag
        loss, grads = pl.loss_and_grad(generate=True, step=7)          # fresh variables: identical inputs in the three runs
        torch.cuda.synchronize()
        res[name] = dict(F0=pl.view("F0").reshape(P, S, L, J).cpu().numpy().astype(np.float64),
                         H=pl.view("H").reshape(P, S, L, J).cpu().numpy().astype(np.float64) if lengthscales else None,
                         f=pl.f.cpu().numpy(), logp=pl.logp.cpu().numpy(), loss=loss.cpu().numpy().copy(),
                         grads=[g.cpu().numpy().copy() for g in grads], pl=pl)
    three = res["three"]["pl"]
    assert torch.equal(res["f32"]["pl"].omega, three.omega) and torch.equal(res["split"]["pl"].omega, three.omega)
    X, Zy = orc.init_trainset(N, L), orc.inducing_Zy(M, L)
    pts = np.concatenate([X, Zy], axis=0)
    ell, var = three.lengthscales().cpu().numpy(), three.variances().cpu().numpy()
    worst = {}
    for k in range(P):
        nz = orc.Noise(three.omega[k].cpu().numpy().astype(np.float64), three.beta[k].cpu().numpy().astype(np.float64),
                       three.w[k].cpu().numpy().astype(np.float64), None, None)
        Phi, dPhi = orc.rff_features(nz, pts, ell[k], var[k], True)
        F0 = np.matmul(nz.w.transpose(1, 0, 2), Phi.transpose(0, 2, 1)).transpose(1, 0, 2)
        H = np.matmul(nz.w.transpose(1, 0, 2), dPhi.transpose(0, 2, 1)).transpose(1, 0, 2)
        for name in ("split", "f32", "three"):
            e = np.abs(res[name]["F0"][k] - F0).max() / np.abs(F0).max()
            worst[name, "F0"] = max(worst.get((name, "F0"), 0.0), e)
            if lengthscales:
                e = np.abs(res[name]["H"][k] - H).max() / np.abs(H).max()
                worst[name, "H"] = max(worst.get((name, "H"), 0.0), e)
    print("max |device - float64| / max|.|:", {f"{a} {b}": f"{v:.2e}" for (a, b), v in sorted(worst.items())})
    for mat in ("F0", "H") if lengthscales else ("F0",):
        assert worst["f32", mat] < 2e-5 and worst["split", mat] < 2e-5
        assert worst["split", mat] <= 1.5 * worst["f32", mat] + 2e-7, (mat, worst["split", mat], worst["f32", mat])
    # downstream of the draws: paths, likelihood, loss, gradients (a float32 sphere centre may change voxel: fractions)
    a, b = res["split"], res["f32"]
    np.testing.assert_allclose(a["f"], b["f"], rtol=0, atol=2e-5)
    ok = np.isclose(a["logp"], b["logp"], rtol=2e-3, atol=1e-4)
    assert ok.mean() > 0.99
    # two device forms whose paths differ by <= 2e-5: a few sphere centres resolve to neighbouring voxels.  The share is measured,
    # printed and bounded, and the allowance it buys is capped (a wrong gradient cannot hide behind it)
    flips = 1.0 - ok.mean()
    worst_g = max(np.abs(ga - gb).max() / (np.abs(gb).max() + 1e-12) for k, (ga, gb) in enumerate(zip(a["grads"], b["grads"]))
                  if not (k == 2 and not lengthscales))
    print(f"PARITY split-vs-f32 flipped_share={flips:.5f} loss={np.abs(a['loss'] - b['loss']).max() / np.abs(b['loss']).max():.2e} grad={worst_g:.2e}")
    assert flips <= 5e-3
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=min(10 * flips, 2e-2) + 1e-4)
    for k, (ga, gb) in enumerate(zip(a["grads"], b["grads"])):
        if k == 2 and not lengthscales:
            continue                                    # no lengthscale tangent: that gradient is not formed
        assert np.abs(ga - gb).max() <= (min(10 * flips, 2e-2) + 1e-3) * np.abs(gb).max() + 1e-12
    # ... and three optimisation steps stay together (Adam normalises: float32-level gradient differences move a variable by
    # far less than lr per step)
    sp, fp = res["split"]["pl"], res["f32"]["pl"]
    for _ in range(3):
        sp.step(); fp.step()
    torch.cuda.synchronize()
    assert torch.isfinite(sp.q_mu).all()
    assert float((sp.q_mu - fp.q_mu).abs().max()) < 3 * sp.lr * 2e-2


@pytest.mark.parametrize("robot,S,M,N,P,ell", [("franka", 7, 24, 70, 12, True), ("franka", 64, 30, 100, 6, True), ("ur10", 32, 18, 70, 13, True),
                                               ("franka", 20, 15, 37, 11, True), ("franka", 7, 24, 70, 12, False)])
def test_rows_role_in_registers_gives_the_bits_of_its_lds_form(robot, S, M, N, P, ell):
    """Batches run stage B's rows role -- A = Kfu (Kuu + jI)^-1 and its two tangents -- on one wave per 16 time points with the
    products chained in registers (csrc/gp_cov.h::cov_rows_wave_body), the inverse formed once per latent by stage A;
    VGPMP_COV_LDS_ROWS keeps the LDS form (cov_rows_body / cov_rows_padded_body, four waves and four barriers per 16 time points, the
    inverse by every row-tile workgroup).  Same products in the same order: A4 (three planes), Kinv and three optimisation steps
    agree bit for bit.  Mz = 26, 32, 20, 17; N a multiple of 16 or not."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset(robot, "industrial")
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    assert P * spec.dof > (64 if S <= 32 else 32)             # a batch schedule
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=256, lengthscales=[2.0] * spec.dof, variance=0.2, seed=3,
              trainable=dict(q_mu=True, q_sqrt=True, lengthscales=ell, kernel_variance=True))      # (ell False: no d/d ell plane)
    a, b = engine.PlannerBatch(sc, qs, **kw), engine.PlannerBatch(sc, qs, **kw)
    b.extra_flags |= capi.COV_LDS_ROWS
    a.run_steps(3); b.run_steps(3)
    torch.cuda.synchronize()
    for v in ("A4", "Kinv", "C"):
        assert torch.equal(a.view(v), b.view(v)), v
    for x, y in ((a.q_mu, b.q_mu), (a.q_sqrt, b.q_sqrt), (a.raw_ell, b.raw_ell), (a.raw_var, b.raw_var)):
        assert torch.equal(x, y)
    assert float(a.view("A4").abs().max()) > 0.0


@pytest.mark.parametrize("robot,S", [("ur10", 1024), ("franka", 512)])
def test_many_sample_gemm_role_on_the_f16_pipe_against_its_float32_form(robot, S):
    """A few problems of 512 samples or more (BASELINE config 4 on one rank: 1024) run stage 2's GEMM role with f16-split
    products (csrc/gp_prior.h::prior_gemm_lds_body<true>: the float32 LDS tiles split in registers, three MFMAs per product);
    VGPMP_PRIOR_F32 keeps the float32 MFMAs.  Same noise, same variables: paths within 2e-5, loss and gradients to float32
    accuracy, and not the same bits (the f16 form did run)."""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset(robot, "industrial")
    spec = rb.load_robot(robot, *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[3]])
    kw = dict(num_samples=S, num_inducing=18, num_data=70, num_bases=1024, lengthscales=[2.0] * spec.dof, variance=0.2, seed=9)
    res = {}
    for name, flag in (("f16", 0), ("f32", capi.PRIOR_F32)):
        pl = engine.PlannerBatch(sc, qs, **kw)
        pl.extra_flags |= flag
        loss, grads = pl.loss_and_grad(generate=True, step=7)
        torch.cuda.synchronize()
        res[name] = dict(f=pl.f.cpu().numpy(), logp=pl.logp.cpu().numpy(), loss=loss.cpu().numpy().copy(),
                         grads=[g.cpu().numpy().copy() for g in grads], pl=pl)
    a, b = res["f16"], res["f32"]
    assert np.abs(a["f"] - b["f"]).max() > 0.0
    np.testing.assert_allclose(a["f"], b["f"], rtol=0, atol=2e-5)
    ok = np.isclose(a["logp"], b["logp"], rtol=2e-3, atol=1e-4)
    assert ok.mean() > 0.99
    flips = 1.0 - ok.mean()
    worst_g = max(np.abs(ga - gb).max() / (np.abs(gb).max() + 1e-12) for ga, gb in zip(a["grads"], b["grads"]))
    print(f"PARITY gemm-f16-vs-f32 flipped_share={flips:.5f} loss={np.abs(a['loss'] - b['loss']).max() / np.abs(b['loss']).max():.2e} grad={worst_g:.2e}")
    assert flips <= 5e-3
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=min(10 * flips, 2e-2) + 1e-4)
    for ga, gb in zip(a["grads"], b["grads"]):
        assert np.abs(ga - gb).max() <= (min(10 * flips, 2e-2) + 1e-3) * np.abs(gb).max() + 1e-12
    sp, fp = a["pl"], b["pl"]
    for _ in range(3):
        sp.step(); fp.step()
    torch.cuda.synchronize()
    assert torch.isfinite(sp.q_mu).all()
    assert float((sp.q_mu - fp.q_mu).abs().max()) < 3 * sp.lr * 2e-2


@pytest.mark.parametrize("S,N,M,P", [(128, 100, 30, 64), (70, 20, 5, 63), (37, 50, 10, 40)])
def test_reverse_path_pass_over_several_chunks_per_workgroup_is_bitwise_the_same(S, N, M, P):
    """paths_bwd_sc8 stages a latent's A / C tangents once and walks several 8-sample chunks (the
    per-chunk re-staging moved 3.6x the operand bytes); the per-chunk partial sums it leaves are the same numbers in the
    same order as with one chunk per workgroup (flag BWD_ONE_CHUNK), so losses, gradients and updates agree bit for bit.
    (Mz = 32, the first case: paths_bwd_regs -- one set of sums per WORKGROUP; agreement to float32 rounding of the chunk sum.)"""
    from vgpmp_amd import capi, engine
    ps = rb.load_problemset("franka", "industrial")
    spec = rb.load_robot("franka")
    grid = scenes.synthetic_boxes_sdf(n=48, delta=0.05, origin=(-1.2, -1.2, -0.6), seed=0)
    sc = engine.DeviceScene(spec, grid, ps.object_positions[0])
    qs = np.array([ps.queries[i % len(ps.queries)] for i in range(P)])
    kw = dict(num_samples=S, num_inducing=M, num_data=N, num_bases=128, lengthscales=[2.0] * 7, variance=0.2, seed=4, split_k=1)
    outs = []
    for flag in (0, capi.BWD_ONE_CHUNK):
        pl = engine.PlannerBatch(sc, qs, **kw)
        pl.fuse = False
        pl.extra_flags = flag
        regs = M + 2 == 32 and N <= 100 and N % 4 == 0      # (paths_bwd_regs: see the docstring)
        if not regs:
            pl.step(); pl.step()      # (not for the rounding comparison: Adam turns 1e-7 of a near-zero gradient into 1e-5 of a variable)
        loss, grads = pl.loss_and_grad(generate=True, step=5)
        torch.cuda.synchronize()
        # (the flag also keeps the forward assembly on paths_fwd_sc8; without it paths_fwd_regs forms the paths: f and R as well)
        outs.append([loss.clone(), pl.q_mu.clone(), pl.q_sqrt.clone(), pl.raw_ell.clone(), pl.raw_var.clone(), pl.f.clone(), pl.view("R")]
                    + [g.clone() for g in grads])
    assert float(outs[0][7].abs().max()) > 0
    if regs:
        # the register-resident kernel (paths_bwd_regs) adds its chunks' sums up itself and leaves ONE set per workgroup (round 5): per
        # chunk the same numbers, their sum in float32 chunk order instead of the assembly's float64 -- rounding of a float32 sum
        for a, b in zip(*outs):
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30, float((a - b).abs().max()) / float(b.abs().max())
        return
    for a, b in zip(*outs):
        assert torch.equal(a, b)
