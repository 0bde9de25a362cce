"""Measurement aid: how many table gathers of the SDF pass a TIME-COHERENCE test could skip, exactly.

The grid of the config-5 share is a sampled 1-Lipschitz distance field (vgpmp_mesh_sdf output and the analytic scenes are), and the
lookup is nearest-voxel (utils/sdf_utils.py:62-76): with c(x) the centre of the voxel x falls in, |c(x) - x| <= sqrt(3)/2 delta, so
for two time points of the same sphere

    grid[c(x_n)] >= grid[c(x_a)] - |x_n - x_a| - sqrt(3) delta.

If an ANCHOR time point a was gathered and its value clears epsilon + r by more than the displacement + sqrt(3) delta, the hinge at n
is exactly 0 (likelihoods/likelihood.py:131-143) without a gather.  With a time-major lane map the anchor is a neighbouring lane.
This script counts, on the paths of a config-5 plan (early, middle, late), among the queries the kernel's free-space masks do NOT
answer (the ones that gather today), how many such a test would skip for anchors every 2 / 4 / 8 time points (anchors always gather
unless masked; a dependent uses its nearest anchor on either side that was gathered).  No kernel before this count.

    python tools/sdf_time_coherence.py [--problems 64] [--grid 512] [--steps 5,100,200]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
from sdf_frames import block_min  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", default="5,100,200")
    a = ap.parse_args()
    args = bench.resolve(bench.parse_args(["--workload", "stress", "--problems", str(a.problems), "--grid", str(a.grid), "--summary", "on"]))
    ps, spec, grid, scene, pl = bench.build_problem(0, args, 1)
    P, S, L, N = pl.P, pl.S, pl.L, pl.N
    Q = spec.num_spheres
    dev = pl.device
    eps, delta = float(scene.epsilon), float(scene.delta)
    radii = torch.as_tensor(spec.sphere_radii, dtype=torch.float32, device=dev)
    shift = getattr(scene, "mask_shift", None)
    clr = getattr(scene, "mask_clearances", None)
    assert scene.free_space_mask and shift is not None, "the scene must carry the free-space masks of the batch likelihood kernel"
    assert len(clr) == 1, "one radius class expected for the config-5 arm"
    bm = block_min(scene, 1 << shift)
    slack = math.sqrt(3.0) * delta
    print(f"{P} problems x {S} samples x {N} times x {Q} spheres; epsilon {eps}, delta {delta:.5f}, mask blocks {1 << shift}^3 voxels, "
          f"clearance {clr[0]:.5f}, slack sqrt(3) delta = {slack:.5f}")
    off = torch.as_tensor(scene.scene_offset, dtype=torch.float64, device=dev)
    done = 0
    for target in [int(v) for v in a.steps.split(",")]:
        while done < target:
            pl.step(); done += 1
        pl.elbo(generate=True)
        torch.cuda.synchronize()
        tot = unmasked = 0
        skip = {2: 0, 4: 0, 8: 0}
        anchors_gathering = {2: 0, 4: 0, 8: 0}
        disp_sum, disp_n = 0.0, 0
        for p in range(P):
            g = scene.joint_sigmoid(pl.f[p].permute(0, 2, 1)).reshape(S * N, L)
            pos = scene.fk_spheres(g).to(torch.float64)                                  # [S N, Q, 3]
            idx, dist, _ = scene.sdf_query((pos - off).reshape(-1, 3))
            idx = idx.to(torch.int64).reshape(S, N, Q, 3)
            dv = dist.reshape(S, N, Q).to(torch.float32)
            x = pos.reshape(S, N, Q, 3).to(torch.float32)
            masked = bm[idx[..., 0] >> shift, idx[..., 1] >> shift, idx[..., 2] >> shift] >= clr[0]          # [S, N, Q]: no gather today
            tot += masked.numel()
            unmasked += int((~masked).sum())
            d1 = (x[:, 1:] - x[:, :-1]).norm(dim=-1)
            disp_sum += float(d1.sum()); disp_n += d1.numel()
            margin = dv - radii[None, None, :] - eps                                      # what an anchor's value clears the hinge by
            n_idx = torch.arange(N, device=dev)
            for step in skip:
                is_anchor = (n_idx % step) == 0
                anchors_gathering[step] += int((~masked[:, is_anchor]).sum())
                ok = torch.zeros_like(masked)
                for side in (0, 1):                                                       # nearest anchor below / above
                    an = (n_idx // step) * step + (step if side else 0)
                    valid = an < N
                    an = an.clamp(max=N - 1)
                    xa, ma, ga = x[:, an], margin[:, an], ~masked[:, an]
                    d = (x - xa).norm(dim=-1)
                    ok |= ga & valid[None, :, None] & (ma - d - slack >= 0)
                dep = (~is_anchor)[None, :, None] & (~masked)
                skip[step] += int((ok & dep).sum())
        print(f"== after {done} steps: {tot} sphere queries, {unmasked} ({unmasked / tot:.3f}) gather today (not answered by the masks); "
              f"mean displacement between adjacent time points {disp_sum / disp_n:.4f} m = {disp_sum / disp_n / delta:.1f} voxels")
        for step in skip:
            print(f"   anchors every {step} time points: {skip[step]} of today's gathers skipped = {skip[step] / max(unmasked, 1):.3f} "
                  f"(anchors that gather: {anchors_gathering[step] / max(unmasked, 1):.3f} of them)")


if __name__ == "__main__":
    main()
