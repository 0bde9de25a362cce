"""One rank of the two-process sample-sharding test (tests/test_gpu_sharded.py).  Started by tests/conftest.py at session
start -- BEFORE the pytest process has touched the GPU -- as a fresh interpreter:

    python tests/shard_worker.py <rank> <world> <port> <out dir>

Both ranks share the one GPU of the box, so the collective runs over gloo (RCCL refuses two ranks on one device); the
planner, its sample slice, the in-place reduction of its contiguous gradient buffer and the replicated Adam update are
exactly what a multi-GPU run executes (vgpmp_amd.sharding.SampleShardedPlanner)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

S_TOTAL, N, M, B, STEPS, SEED = 16, 11, 6, 64, 3, 77


def problem():
    """BASELINE config 4's robot (UR10, classic DH, twist, variance on the positive(0.1) floor) on a small obstacle grid."""
    import numpy as np

    from vgpmp_amd import robots, scenes
    ps = robots.load_problemset("ur10", "industrial")
    spec = robots.load_robot("ur10", *ps.robot_pos_and_orn)
    grid = scenes.synthetic_boxes_sdf(n=40, delta=0.06, origin=(-1.2, -1.2, -0.6), seed=4)
    q = np.array([[ps.states[0], ps.states[1]]])
    kw = dict(num_inducing=M, num_data=N, num_bases=B, lengthscales=ps.planner_params["lengthscales"],
              variance=ps.planner_params["variance"], alpha=float(ps.planner_params["alpha"]),
              learning_rate=float(ps.planner_params["learning_rate"]), seed=SEED, split_k=1)
    return spec, grid, ps.object_positions[0], q, kw


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist

    from vgpmp_amd import engine, sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    spec, grid, off, q, kw = problem()
    sc = engine.DeviceScene(spec, grid, off)
    s_loc, s_off = sharding.shard_samples(S_TOTAL, world, rank)
    pl = engine.PlannerBatch(sc, q, num_samples=s_loc, samples_total=S_TOTAL, sample_offset=s_off,
                             kl_scale=1.0 if rank == 0 else 0.0, **kw)
    sp = sharding.SampleShardedPlanner(pl)
    elbos = []
    for _ in range(STEPS):
        sp.step()
        elbos.append(float((pl.lik - pl.kl)[0]))          # the reduced buffer: whole-job ELBO pieces of this step
    torch.cuda.synchronize()
    np.savez(os.path.join(out, f"rank{rank}.npz"), q_mu=pl.q_mu.cpu().numpy(), q_sqrt=pl.q_sqrt.cpu().numpy(),
             raw_ell=pl.raw_ell.cpu().numpy(), raw_var=pl.raw_var.cpu().numpy(), elbo=np.array(elbos), t=pl.t)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
