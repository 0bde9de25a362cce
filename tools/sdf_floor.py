"""Measurement aid: the floor of the SDF pass's table traffic for one launch of the config-5 share (64 problems of the
14-DoF arm, 512^3 voxels = 2 GiB table of 16-byte records in 4x4x4 Morton bricks).

For the paths of one optimisation step it recomputes every sphere query on the device (vgpmp_fk_spheres,
vgpmp_sdf_query: the kernels' own index arithmetic), applies the free-space test of the batch likelihood kernel
(brick minimum vs epsilon + radius) and counts the DISTINCT 64-byte sectors / 128-byte lines the remaining queries touch:
  * over the whole launch            -- what a cache of unbounded size in front of HBM would have to fetch;
  * per workgroup (64 configurations) -- what the launch fetches if nothing is shared between workgroups;
  * per XCD under round-robin placement of the workgroups (each XCD's L2 unbounded).
profiles/r03/ keeps the output next to the FETCH_SIZE of the same launch: the kernel moves ~1.0-1.1x the per-XCD floor.

    python tools/sdf_floor.py [--problems 64] [--grid 512]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    args = bench.resolve(bench.parse_args(["--workload", "stress", "--problems", str(a.problems), "--grid", str(a.grid)]))
    ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
    for _ in range(a.steps):
        pl.step()
    pl.elbo(generate=True)                       # forward only: pl.f holds the step's paths
    torch.cuda.synchronize()
    P, S, L, N = pl.P, pl.S, pl.L, pl.N
    nsph = spec.num_spheres
    eps = scene.epsilon
    radii = torch.as_tensor(spec.sphere_radii, dtype=torch.float32, device=pl.device)
    nx, ny, nz = scene.shape
    nby, nbz = (ny + 3) // 4, (nz + 3) // 4
    tot_q = tot_near = 0
    sect_all, line_all = [], []
    per_wg_sect = per_wg_line = 0
    xcd_sets = [[] for _ in range(8)]
    nblk = (S * N + 63) // 64
    for p in range(P):
        g = scene.joint_sigmoid(pl.f[p].permute(0, 2, 1)).reshape(S * N, L)          # [S N, L] joint angles
        pos = scene.fk_spheres(g).to(torch.float64)                                    # [S N, nsph, 3]
        rel = pos - torch.as_tensor(scene.scene_offset, dtype=torch.float64, device=pos.device)
        idx, _, _ = scene.sdf_query(rel.reshape(-1, 3))
        idx = idx.to(torch.int64).reshape(S * N, nsph, 3)
        ix, iy, iz = idx[..., 0], idx[..., 1], idx[..., 2]
        brick = ((ix >> 2) * nby + (iy >> 2)) * nbz + (iz >> 2)
        morton = (iz & 1) | ((iy & 1) << 1) | ((ix & 1) << 2) | ((iz & 2) << 2) | ((iy & 2) << 3) | ((ix & 2) << 4)
        off = brick * 64 + morton                                                        # 16-byte records
        near = (eps - (scene.brick_min[brick] - radii[None, :])) > 0.0                   # the kernel's free-space test
        tot_q += off.numel(); tot_near += int(near.sum())
        sect = torch.where(near, off >> 2, torch.full_like(off, -1))                     # 64-byte sector = 4 records
        line = torch.where(near, off >> 3, torch.full_like(off, -1))
        sect_all.append(torch.unique(sect[near])); line_all.append(torch.unique(line[near]))
        # per workgroup: 64 consecutive configurations
        pad = nblk * 64 - S * N
        sp = torch.cat([sect, torch.full((pad, nsph), -1, dtype=sect.dtype, device=sect.device)]).reshape(nblk, 64 * nsph)
        lp = torch.cat([line, torch.full((pad, nsph), -1, dtype=line.dtype, device=line.device)]).reshape(nblk, 64 * nsph)
        for b in range(nblk):
            us = torch.unique(sp[b]); ul = torch.unique(lp[b])
            us = us[us >= 0]; ul = ul[ul >= 0]
            per_wg_sect += us.numel(); per_wg_line += ul.numel()
            xcd_sets[(p * nblk + b) % 8].append(us)                                      # grid (nblk, P): linear index x + nblk y
    glob_s = torch.unique(torch.cat(sect_all)).numel()
    glob_l = torch.unique(torch.cat(line_all)).numel()
    xcd_s = sum(torch.unique(torch.cat(v)).numel() for v in xcd_sets)
    q = float(tot_q)
    print(f"sphere queries per launch          {tot_q}  ({P} problems x {S * N} configurations x {nsph} spheres)")
    print(f"queries that read the table        {tot_near}  ({tot_near / q:.3f} of all; the rest lie in free space by the brick summary)")
    print(f"distinct 64-B sectors, whole launch {glob_s}  = {glob_s * 64 / 1e6:.1f} MB   ({glob_s / q:.3f} per query)")
    print(f"distinct 128-B lines, whole launch  {glob_l}  = {glob_l * 128 / 1e6:.1f} MB")
    print(f"distinct sectors summed over XCDs   {xcd_s}  = {xcd_s * 64 / 1e6:.1f} MB   (round-robin workgroups, each L2 unbounded)")
    print(f"distinct sectors summed over workgroups {per_wg_sect}  = {per_wg_sect * 64 / 1e6:.1f} MB   ({per_wg_sect / q:.3f} per query)")
    print(f"distinct lines summed over workgroups   {per_wg_line}  = {per_wg_line * 128 / 1e6:.1f} MB")
    print(f"brick summary reads: {tot_q} x 4 B = {tot_q * 4 / 1e6:.1f} MB requested, table of {scene.brick_min.numel() * 4 / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
