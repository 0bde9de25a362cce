"""Signed-distance grids for the planner: text-format I/O and synthetic scenes.

The reference's `.sdf` blobs are missing from its checkout (.MISSING_LARGE_BLOBS:3-7), so
benchmark scenes are analytic unions of boxes and spheres sampled at the lattice points the
reference indexes (value stored for voxel (i,j,k) = SDF at origin + delta*(i,j,k)).
File format: gpflow_vgpmp/utils/sdf_utils.py:195-210 (header, then one value per line, x fastest).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

Grid = Tuple[np.ndarray, np.ndarray, float]   # data[x, y, z] float64, origin[3], delta


def read_sdf(path: str) -> Grid:
    with open(path, "r") as fh:
        nx, ny, nz = (int(v) for v in fh.readline().split())
        origin = np.array([float(v) for v in fh.readline().split()], dtype=np.float64)
        delta = float(fh.readline().strip())
        vals = np.loadtxt(fh, dtype=np.float64).reshape(-1)
    if vals.size != nx * ny * nz:
        raise ValueError(f"{path}: expected {nx * ny * nz} values, found {vals.size}")
    # line i holds data[i % nx, (i // nx) % ny, i // (nx * ny)]
    return vals.reshape(nz, ny, nx).transpose(2, 1, 0).copy(), origin, delta


def write_sdf(path: str, grid: Grid) -> None:
    data, origin, delta = grid
    nx, ny, nz = data.shape
    with open(path, "w") as fh:
        fh.write(f"{nx} {ny} {nz}\n")
        fh.write(" ".join(repr(float(v)) for v in origin) + "\n")
        fh.write(repr(float(delta)) + "\n")
        np.savetxt(fh, np.asarray(data).transpose(2, 1, 0).reshape(-1), fmt="%.17g")


def _box_sdf(p: np.ndarray, centre, half) -> np.ndarray:
    q = np.abs(p - np.asarray(centre)) - np.asarray(half)
    outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
    inside = np.minimum(np.max(q, axis=-1), 0.0)
    return outside + inside


def lattice(shape: Sequence[int], origin, delta: float, dtype=np.float64) -> np.ndarray:
    ax = [np.asarray(origin[i], dtype) + dtype(delta) * np.arange(shape[i], dtype=dtype) for i in range(3)]
    return np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1)


def _synthetic_shapes(n: int, delta: float, origin, seed: int, n_boxes: int, n_spheres: int):
    rng = np.random.default_rng(seed)
    origin = np.asarray(origin, dtype=np.float64)
    ext = delta * n
    boxes = [(origin + rng.uniform(0.15, 0.85, 3) * ext, rng.uniform(0.04, 0.14, 3) * ext) for _ in range(n_boxes)]
    balls = [(origin + rng.uniform(0.15, 0.85, 3) * ext, rng.uniform(0.04, 0.10) * ext) for _ in range(n_spheres)]
    return origin, boxes, balls


def synthetic_boxes_sdf(n: int = 128, delta: float = 0.0125, origin=(-0.8, -0.8, -0.2), seed: int = 0,
                        n_boxes: int = 6, n_spheres: int = 4, dtype=np.float64) -> Grid:
    """Union of `n_boxes` axis-aligned boxes and `n_spheres` spheres, placed by `seed` inside the
    grid extent (SURVEY 8d defaults: 128^3, delta 0.0125, origin (-0.8,-0.8,-0.2))."""
    origin, boxes, balls = _synthetic_shapes(n, delta, origin, seed, n_boxes, n_spheres)
    shape = (n, n, n)
    out = np.full(shape, np.inf, dtype=dtype)
    # evaluate slab by slab so large grids stay within memory
    step = max(1, min(n, (1 << 22) // (n * n)))
    for x0 in range(0, n, step):
        x1 = min(n, x0 + step)
        p = lattice((x1 - x0, n, n), origin + np.array([x0 * delta, 0.0, 0.0]), delta)
        d = np.full(p.shape[:-1], np.inf)
        for c, h in boxes:
            d = np.minimum(d, _box_sdf(p, c, h))
        for c, r in balls:
            d = np.minimum(d, np.linalg.norm(p - c, axis=-1) - r)
        out[x0:x1] = d.astype(dtype)
    return out, origin, float(delta)


class AnalyticSceneRows:
    """The same synthetic scene as `synthetic_boxes_sdf`, evaluated lazily on a device, rows [x_lo, x_hi) at a
    time (float64 torch tensor [x_hi - x_lo, n, n]): a 512^3 grid (1 GiB in float64) is never held whole.
    `engine.DeviceScene` accepts it in place of the array of a Grid.  Synthetic-data plumbing, not product code."""

    def __init__(self, n: int, delta: float, origin, seed: int = 0, n_boxes: int = 6, n_spheres: int = 4,
                 round_to=None):
        self.shape = (int(n), int(n), int(n))
        self.delta = float(delta)
        self.origin, self.boxes, self.balls = _synthetic_shapes(n, delta, origin, seed, n_boxes, n_spheres)
        self.round_to = round_to        # e.g. torch.float32: values rounded as a float32 grid file would hold them

    def rows(self, x_lo: int, x_hi: int, device):
        import torch
        f64 = torch.float64
        n = self.shape[0]
        ax = [torch.as_tensor(self.origin[i], dtype=f64, device=device)
              + self.delta * torch.arange(n if i else x_hi - x_lo, dtype=f64, device=device) for i in range(3)]
        ax[0] = ax[0] + x_lo * self.delta
        X, Y, Z = torch.meshgrid(*ax, indexing="ij")
        d = torch.full(X.shape, float("inf"), dtype=f64, device=device)
        zero = torch.zeros((), dtype=f64, device=device)
        for c, h in self.boxes:
            qx, qy, qz = (X - c[0]).abs() - h[0], (Y - c[1]).abs() - h[1], (Z - c[2]).abs() - h[2]
            outside = torch.sqrt(torch.maximum(qx, zero) ** 2 + torch.maximum(qy, zero) ** 2 + torch.maximum(qz, zero) ** 2)
            inside = torch.minimum(torch.maximum(torch.maximum(qx, qy), qz), zero)
            d = torch.minimum(d, outside + inside)
        for c, r in self.balls:
            d = torch.minimum(d, torch.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r)
        if self.round_to is not None:
            d = d.to(self.round_to).to(f64)
        return d.contiguous()


# ---- mesh -> SDF (device) -------------------------------------------------------------------------------
def load_scene_mesh(name: str):
    """Triangles [T, 9] (float64) and part ids [T] (int32) of a scene collision mesh
    (vgpmp_amd/data/scene_meshes.json: vertices/faces of the reference's data/scenes/*/*.obj)."""
    import json
    from pathlib import Path
    tables = json.load(open(Path(__file__).resolve().parent / "data" / "scene_meshes.json"))
    if name not in tables:
        raise KeyError(f"no collision mesh for scene {name!r} (have {sorted(tables)})")
    t = tables[name]
    V = np.asarray(t["vertices"], dtype=np.float64)
    F = np.asarray(t["faces"], dtype=np.int64)
    part = np.asarray(t["part"], dtype=np.int32)
    order = np.argsort(part, kind="stable")
    return V[F[order]].reshape(-1, 9), part[order]


def mesh_grid_extent(triangles: np.ndarray, delta: float, padding: int):
    """Grid of spacing `delta` covering the mesh bounding box plus `padding` cells on every side
    (the layout SDFGen produces for gen_sdf.py: origin = bbox_min - padding * delta)."""
    pts = np.asarray(triangles, dtype=np.float64).reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    origin = lo - padding * delta
    shape = tuple(int(v) for v in np.ceil((hi - lo) / delta).astype(int) + 2 * padding + 1)
    return origin, shape


def mesh_sdf(triangles: np.ndarray, part: np.ndarray, delta: float, padding: int = 20, origin=None, shape=None) -> Grid:
    """Signed distance grid of a triangle mesh, computed on the GPU (vgpmp_mesh_sdf)."""
    import ctypes as C

    import torch

    from . import capi
    lib = capi.load(require=True)
    if not torch.cuda.is_available():
        raise capi.VgpmpError("mesh_sdf needs a HIP device (no CPU fallback)")
    if origin is None or shape is None:
        origin, shape = mesh_grid_extent(triangles, delta, padding)
    origin = np.asarray(origin, dtype=np.float64)
    tri = torch.as_tensor(np.ascontiguousarray(triangles, dtype=np.float64)).cuda()
    prt = torch.as_tensor(np.ascontiguousarray(part, dtype=np.int32)).cuda()
    grid = torch.empty(shape, dtype=torch.float64, device="cuda")
    org = (C.c_double * 3)(*origin)
    capi.check(lib.vgpmp_mesh_sdf(capi.ptr(tri), capi.ptr(prt), int(tri.shape[0]), *shape, org, float(delta),
                                  capi.ptr(grid), capi.stream_ptr()), "vgpmp_mesh_sdf")
    return grid.cpu().numpy(), origin, float(delta)


def scene_sdf(name: str, delta: float = 0.0125, padding: int = 20) -> Grid:
    """SDF of one of the reference's scenes (industrial, bookshelves, boxes, lab) from its collision mesh."""
    tri, part = load_scene_mesh(name)
    return mesh_sdf(tri, part, delta, padding)
