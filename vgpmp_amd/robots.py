"""Robot and problem-set tables of the planner (host side, no GPU work).

The tables in ``vgpmp_amd/data/*.json`` hold the numbers of the reference's
``data/robots/<name>/config.yaml``, the sphere visuals of its URDFs and
``data/problemsets/<name>.py`` (re-entered by tools/extract_reference_data.py).
``RobotSpec`` carries what ``gpflow_vgpmp/utils/sampler.py:28-56`` keeps of a robot:
DH table, twist, convention flag, base pose, frame slice, per-sphere offsets/radii.
"""
from __future__ import annotations

import dataclasses
import itertools
import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_DATA = Path(__file__).resolve().parent / "data"
AVAILABLE_ROBOTS = ("franka", "ur10", "wam", "kuka")   # data/problemsets/config.py


def _load(name: str) -> dict:
    with open(_DATA / name, "r") as fh:
        return json.load(fh)


def quat_to_rotmat(q: Sequence[float]) -> np.ndarray:
    """(x, y, z, w) quaternion -> 3x3 rotation (utils/robot.py:17-19 uses scipy's from_quat)."""
    x, y, z, w = (float(v) for v in q)
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


def base_pose_matrix(position: Sequence[float], orientation: Sequence[float]) -> np.ndarray:
    """utils/robot.py:196-203 get_base_pose: homogeneous matrix of the base position/orientation."""
    T = np.eye(4)
    T[:3, :3] = quat_to_rotmat(orientation)
    T[:3, 3] = position
    return T


@dataclasses.dataclass
class RobotSpec:
    name: str
    dof: int
    craig: bool
    dh: np.ndarray                 # [D, 3]  d, a, alpha
    twist: np.ndarray              # [D]
    base_pose: np.ndarray          # [4, 4]
    fk_slice: np.ndarray           # [F] int
    num_spheres_per_link: np.ndarray  # [F] int
    sphere_offsets: np.ndarray     # [P, 3]
    sphere_radii: np.ndarray       # [P]
    joint_limits: np.ndarray       # [D, 2] (high, low)
    velocity_limits: np.ndarray    # [D, 2]
    raw: dict = dataclasses.field(default_factory=dict, repr=False)

    @property
    def num_spheres(self) -> int:
        return int(self.sphere_offsets.shape[0])

    @property
    def num_frames_for_spheres(self) -> int:
        return int(self.fk_slice.shape[0])

    @property
    def sphere_frame(self) -> np.ndarray:
        return np.repeat(self.fk_slice, self.num_spheres_per_link).astype(np.int32)

    @property
    def low(self) -> np.ndarray:
        return self.joint_limits[:, 1]

    @property
    def high(self) -> np.ndarray:
        return self.joint_limits[:, 0]


def load_robot(name: str, position: Sequence[float] = (0.0, 0.0, 0.0),
               orientation: Sequence[float] = (0.0, 0.0, 0.0, 1.0)) -> RobotSpec:
    tables = _load("robots.json")
    if name not in tables:
        raise KeyError(f"Robot not available: {name!r} (have {sorted(tables)})")
    t = tables[name]
    return RobotSpec(name=name, dof=int(t["dof"]), craig=bool(t["craig_dh_convention"]),
                     dh=np.array(t["dh_parameters"], dtype=np.float64).reshape(-1, 3),
                     twist=np.array(t["twist"], dtype=np.float64),
                     base_pose=base_pose_matrix(position, orientation),
                     fk_slice=np.array(t["fk_slice"], dtype=np.int32),
                     num_spheres_per_link=np.array(t["num_spheres_per_link"], dtype=np.int32),
                     sphere_offsets=np.array(t["sphere_offsets"], dtype=np.float64).reshape(-1, 3),
                     sphere_radii=np.array(t["radius"], dtype=np.float64),
                     joint_limits=np.array(t["joint_limits"], dtype=np.float64).reshape(-1, 2),
                     velocity_limits=np.array(t["velocity_limits"], dtype=np.float64).reshape(-1, 2),
                     raw=t)


def synthetic_arm(dof: int = 14, spheres_per_link: int = 3, radius: float = 0.05,
                  link_length: float = 0.15) -> RobotSpec:
    """BASELINE config 5: classic-DH chain, d = link_length, a = 0, alpha = +-pi/2 alternating,
    `spheres_per_link` spheres on each of frames 1..dof plus the base frame."""
    dh = np.zeros((dof, 3))
    dh[:, 0] = link_length
    dh[:, 2] = np.where(np.arange(dof) % 2 == 0, np.pi / 2, -np.pi / 2)
    fk_slice = np.arange(dof + 1, dtype=np.int32)
    per = np.full(dof + 1, spheres_per_link, dtype=np.int32)
    offs = np.tile(np.linspace(-link_length, 0.0, spheres_per_link, endpoint=False)[:, None]
                   * np.array([[0.0, 0.0, 1.0]]), (dof + 1, 1))
    lim = np.tile(np.array([[np.pi, -np.pi]]), (dof, 1))
    return RobotSpec(name=f"synthetic{dof}", dof=dof, craig=False, dh=dh, twist=np.zeros(dof),
                     base_pose=np.eye(4), fk_slice=fk_slice, num_spheres_per_link=per,
                     sphere_offsets=offs, sphere_radii=np.full(offs.shape[0], radius),
                     joint_limits=lim, velocity_limits=lim.copy())


@dataclasses.dataclass
class ProblemSet:
    robot: str
    name: str
    states: List[List[float]]
    planner_params: Dict
    robot_pos_and_orn: Tuple[List[float], List[float]]
    object_positions: List[List[float]]

    @property
    def queries(self) -> List[Tuple[List[float], List[float]]]:
        """utils/parameter_loader.py:138: all C(n, 2) start-goal pairs."""
        return list(itertools.combinations(self.states, 2))


def load_problemset(robot: str, name: str) -> ProblemSet:
    tables = _load("problemsets.json")
    if robot not in tables:
        raise KeyError(f"Robot not available: {robot!r}")
    if name not in tables[robot]:
        raise ValueError("Unknown problem set: {}".format(name))
    e = tables[robot][name]
    pos_orn = e.get("pos_and_orn", [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    return ProblemSet(robot, name, e["states"], dict(e.get("planner_params", {})),
                      (list(pos_orn[0]), list(pos_orn[1])), e.get("object_positions", [[0.0, 0.0, 0.0]]))
