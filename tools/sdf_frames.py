"""Measurement aid: how much of the SDF pass of the config-5 share a COARSE free-space level could skip.

For the paths of one optimisation step (early and late in a plan) it recomputes every sphere query on the device and
reports, per coarse block size b (voxels per axis: 4 = the brick summary, 8, 16, 32):
  * the fraction of sphere queries whose block minimum clears epsilon + r_max (a per-scene bit mask could hold that test);
  * the fraction of (configuration, frame) pairs whose spheres are ALL free at that level, and the fraction of the
    queries that sit on such frames;
  * the same per (wave of 64 consecutive configurations, frame): what a wave-uniform branch could skip;
  * a conservative ball test per frame: block minima over the blocks the ball (frame origin, max |offset| of its
    spheres) touches.

    python tools/sdf_frames.py [--problems 64] [--grid 512] [--steps 5,200]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def block_min(scene, b):
    """min distance per b^3-voxel block from the 4^3 brick summary (b a multiple of 4)."""
    nx, ny, nz = scene.shape
    nb = [(n + 3) // 4 for n in (nx, ny, nz)]
    bm = scene.brick_min.reshape(nb)
    k = b // 4
    if k == 1:
        return bm
    pad = [(-n) % k for n in nb]
    bm = torch.nn.functional.pad(bm, (0, pad[2], 0, pad[1], 0, pad[0]), value=float("inf"))
    s = bm.shape
    return bm.reshape(s[0] // k, k, s[1] // k, k, s[2] // k, k).amin(dim=(1, 3, 5))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", default="5,200")
    ap.add_argument("--workload", default="stress")
    a = ap.parse_args()
    args = bench.resolve(bench.parse_args(["--workload", a.workload, "--problems", str(a.problems), "--grid", str(a.grid),
                                           "--summary", "on"]))
    ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
    P, S, L, N = pl.P, pl.S, pl.L, pl.N
    nsph = spec.num_spheres
    eps = scene.epsilon
    dev = pl.device
    radii = torch.as_tensor(spec.sphere_radii, dtype=torch.float32, device=dev)
    rmax = float(radii.max())
    frame_of = torch.as_tensor(np.asarray(spec.sphere_frame if hasattr(spec, "sphere_frame") else spec.sphere_link),
                               dtype=torch.int64, device=dev) if (hasattr(spec, "sphere_frame") or hasattr(spec, "sphere_link")) else None
    offs = torch.as_tensor(np.asarray(spec.sphere_offsets), dtype=torch.float32, device=dev)
    print(f"{P} problems x {S * N} configurations x {nsph} spheres; epsilon {eps}, r_max {rmax}, delta {scene.delta}")
    if frame_of is None:
        print("no sphere -> frame table on the spec; frame statistics skipped")
    done = 0
    for target in [int(v) for v in a.steps.split(",")]:
        while done < target:
            pl.step(); done += 1
        pl.elbo(generate=True)
        torch.cuda.synchronize()
        levels = {b: block_min(scene, b) for b in (4, 8, 16, 32)}
        tot = 0
        free = {b: 0 for b in levels}
        free_exact4 = 0
        fr_pairs = 0
        fr_free = {b: 0 for b in levels}; frq_free = {b: 0 for b in levels}
        wv_pairs = 0
        wv_free = {b: 0 for b in levels}; wvq_free = {b: 0 for b in levels}
        for p in range(P):
            g = scene.joint_sigmoid(pl.f[p].permute(0, 2, 1)).reshape(S * N, L)
            pos = scene.fk_spheres(g).to(torch.float64)
            rel = pos - torch.as_tensor(scene.scene_offset, dtype=torch.float64, device=dev)
            idx, _, _ = scene.sdf_query(rel.reshape(-1, 3))
            idx = idx.to(torch.int64).reshape(S * N, nsph, 3)
            tot += idx.shape[0] * nsph
            b4 = levels[4]
            bm4 = b4[idx[..., 0] >> 2, idx[..., 1] >> 2, idx[..., 2] >> 2]
            free_exact4 += int(((eps - (bm4 - radii[None, :])) <= 0).sum())
            for b, tab in levels.items():
                sh = {4: 2, 8: 3, 16: 4, 32: 5}[b]
                v = tab[idx[..., 0] >> sh, idx[..., 1] >> sh, idx[..., 2] >> sh]
                fq = v >= (eps + rmax)                                   # [S N, nsph]
                free[b] += int(fq.sum())
                if frame_of is not None:
                    nfr = int(frame_of.max()) + 1
                    onehot = torch.nn.functional.one_hot(frame_of, nfr).to(torch.float32)     # [nsph, nfr]
                    cnt = onehot.sum(0)                                                        # spheres per frame
                    nfree = fq.to(torch.float32) @ onehot                                      # [S N, nfr]
                    allfree = (nfree == cnt[None, :]) & (cnt[None, :] > 0)
                    if b == 4:
                        fr_pairs += int((cnt > 0).sum()) * idx.shape[0]
                    fr_free[b] += int(allfree.sum())
                    frq_free[b] += int((allfree.to(torch.float32) * cnt[None, :]).sum())
                    nw = (S * N) // 64
                    aw = allfree[:nw * 64].reshape(nw, 64, nfr).all(dim=1)
                    if b == 4:
                        wv_pairs += int((cnt > 0).sum()) * nw
                    wv_free[b] += int(aw.sum())
                    wvq_free[b] += int((aw.to(torch.float32) * cnt[None, :]).sum()) * 64
        print(f"== after {done} steps: {tot} sphere queries")
        print(f"   brick summary with the sphere's own radius (what the kernel skips today): {free_exact4 / tot:.3f}")
        for b in levels:
            line = f"   block {b:2d}^3, r_max: free queries {free[b] / tot:.3f}"
            if frame_of is not None:
                line += (f" | (config, frame) pairs all-free {fr_free[b] / max(fr_pairs, 1):.3f} holding {frq_free[b] / tot:.3f} of the queries"
                         f" | (wave, frame) all-free {wv_free[b] / max(wv_pairs, 1):.3f} holding {wvq_free[b] / tot:.3f}")
            print(line)


if __name__ == "__main__":
    main()
