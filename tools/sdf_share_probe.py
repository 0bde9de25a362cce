"""Measurement aid: is the SDF pass bound by the number of LANES that gather or by the number of distinct SECTORS / LINES a gather
instruction touches?  The config-5 share with time stamps repeated in blocks of k (X[n] = t[n // k]): the k configurations of a
block coincide, so the lanes of a wave share sectors k-fold while the number of reading lanes stays what it is.  Prints, per k, the
likelihood kernel's event time next to reading lanes and distinct sectors per gather instruction (map A of tools/sdf_lane_maps.py).
    python tools/sdf_share_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def sectors(scene, spec, pl):
    P, S, L, N = pl.P, pl.S, pl.L, pl.N
    nsph = spec.num_spheres
    radii = torch.as_tensor(spec.sphere_radii, dtype=torch.float32, device=pl.device)
    nx, ny, nz = scene.shape
    nby, nbz = (ny + 3) // 4, (nz + 3) // 4
    ins = act = dis = lin = 0.0
    for p in range(0, P, 4):
        g = scene.joint_sigmoid(pl.f[p].permute(0, 2, 1)).reshape(S * N, L)
        pos = scene.fk_spheres(g).to(torch.float64)
        rel = pos - torch.as_tensor(scene.scene_offset, dtype=torch.float64, device=pos.device)
        idx, _, _ = scene.sdf_query(rel.reshape(-1, 3))
        idx = idx.to(torch.int64).reshape(S * N, nsph, 3)
        ix, iy, iz = idx[..., 0], idx[..., 1], idx[..., 2]
        brick = ((ix >> 2) * nby + (iy >> 2)) * nbz + (iz >> 2)
        morton = (iz & 1) | ((iy & 1) << 1) | ((ix & 1) << 2) | ((iz & 2) << 2) | ((iy & 2) << 3) | ((ix & 2) << 4)
        off = brick * 64 + morton
        near = (scene.epsilon - (scene.brick_min[brick] - radii[None, :])) > 0.0
        sect = torch.where(near, off >> 2, torch.full_like(off, -1))
        m = sect[: (S * N // 64) * 64].reshape(-1, 64, nsph).permute(0, 2, 1).reshape(-1, 64)
        srt, _ = torch.sort(m, dim=1)
        ln, _ = torch.sort(torch.where(m >= 0, m >> 1, m), dim=1)
        ins += m.shape[0]
        act += float((srt >= 0).sum())
        dis += float((((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(1) + (srt[:, 0] >= 0).long()).sum())
        lin += float((((ln[:, 1:] != ln[:, :-1]) & (ln[:, 1:] >= 0)).sum(1) + (ln[:, 0] >= 0).long()).sum())
    return act / ins, dis / ins, lin / ins


def main():
    args = bench.resolve(bench.parse_args(["--workload", "stress"]))
    ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
    N, L = pl.N, pl.L
    t = np.linspace(0.0, 1.0, N)
    for k in (1, 2, 4, 8, 16, 64):
        X = np.tile(t[(np.arange(N) // k) * k][:, None], (1, L))
        pl.reset()
        pl.set_time_stamps(X)
        pl.run_steps(60)
        times = pl.profile_steps(5)
        pl.elbo(generate=True)
        torch.cuda.synchronize()
        a, d, l = sectors(scene, spec, pl)
        print(f"stamps in blocks of {k:2d}: likelihood kernel {1e3 * times['loglik_kernel']:7.1f} us | per gather instruction: {a:5.1f} reading lanes, "
              f"{d:5.1f} distinct 64-B sectors, {l:5.1f} distinct 128-B lines", flush=True)


if __name__ == "__main__":
    main()
