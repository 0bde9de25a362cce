"""Measurement aid: would another lane <-> (sample, time) map let the lanes of a gather instruction share 64-byte sectors?
("SDF pass beyond 0.53 only if a measured query-pattern change says so before a kernel is written".)  Config-5 share.
For the paths of one optimisation step, early and late in a plan: per gather instruction of the batch likelihood (one sphere index,
64 lanes) the number of DISTINCT 64-byte sectors / 128-byte lines among the lanes that read the table, under
  A  the kernel's map: 64 consecutive (sample, time) configurations, time fastest;
  B  sample-major: 64 samples of ONE time step (two waves per time step at S = 128);
  C  8 samples x 8 consecutive time steps.
    python tools/sdf_lane_maps.py [--steps-list 3,60,190]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=16)
    ap.add_argument("--steps-list", default="3,60,190")
    a = ap.parse_args()
    args = bench.resolve(bench.parse_args(["--workload", "stress", "--problems", str(a.problems)]))
    ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
    P, S, L, N = pl.P, pl.S, pl.L, pl.N
    nsph = spec.num_spheres
    eps = scene.epsilon
    radii = torch.as_tensor(spec.sphere_radii, dtype=torch.float32, device=pl.device)
    nx, ny, nz = scene.shape
    nby, nbz = (ny + 3) // 4, (nz + 3) // 4
    done = 0
    for target in [int(v) for v in a.steps_list.split(",")]:
        pl.run_steps(target - done)
        done = target
        pl.elbo(generate=True)
        torch.cuda.synchronize()
        tot = {k: np.zeros(3) for k in "ABC"}          # instructions, active lanes, distinct sectors
        lines = {k: 0 for k in "ABC"}
        for p in range(P):
            g = scene.joint_sigmoid(pl.f[p].permute(0, 2, 1)).reshape(S * N, L)
            pos = scene.fk_spheres(g).to(torch.float64)
            rel = pos - torch.as_tensor(scene.scene_offset, dtype=torch.float64, device=pos.device)
            idx, _, _ = scene.sdf_query(rel.reshape(-1, 3))
            idx = idx.to(torch.int64).reshape(S, N, nsph, 3)
            ix, iy, iz = idx[..., 0], idx[..., 1], idx[..., 2]
            brick = ((ix >> 2) * nby + (iy >> 2)) * nbz + (iz >> 2)
            morton = (iz & 1) | ((iy & 1) << 1) | ((ix & 1) << 2) | ((iz & 2) << 2) | ((iy & 2) << 3) | ((ix & 2) << 4)
            off = brick * 64 + morton
            near = (eps - (scene.brick_min[brick] - radii[None, None, :])) > 0.0
            sect = torch.where(near, off >> 2, torch.full_like(off, -1))            # [S, N, Q]
            n8 = (N // 8) * 8
            maps = {"A": sect.reshape(S * N, nsph)[: (S * N // 64) * 64].reshape(-1, 64, nsph),
                    "B": sect.permute(1, 0, 2).reshape(N, S // 64, 64, nsph).reshape(-1, 64, nsph),
                    "C": sect[:, :n8].reshape(S // 8, 8, n8 // 8, 8, nsph).permute(0, 2, 1, 3, 4).reshape(-1, 64, nsph)}
            for k, m in maps.items():
                m = m.permute(0, 2, 1).reshape(-1, 64)                              # one row per gather instruction
                srt, _ = torch.sort(m, dim=1)
                active = (srt >= 0).sum(1)
                distinct = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(1) + (srt[:, 0] >= 0).long()
                ln = torch.sort(torch.where(m >= 0, m >> 1, m), dim=1)[0]
                dl = ((ln[:, 1:] != ln[:, :-1]) & (ln[:, 1:] >= 0)).sum(1) + (ln[:, 0] >= 0).long()
                tot[k] += np.array([m.shape[0], float(active.sum()), float(distinct.sum())])
                lines[k] += float(dl.sum())
        print(f"after {target} steps:")
        for k in "ABC":
            ins, act, dis = tot[k]
            print(f"  map {k}: {act / ins:5.1f} of 64 lanes read the table per gather instruction; distinct sectors / reading lanes = {dis / act:.3f}, "
                  f"distinct 128-B lines / reading lanes = {lines[k] / act:.3f}")


if __name__ == "__main__":
    main()
