#!/usr/bin/env python3
"""Re-enter the reference's robot / problem-set tables as this repo's own data files.

Run once in the build container (needs a reference checkout, default /root/reference):

    python tools/extract_reference_data.py [/path/to/vgpmp]

Writes vgpmp_amd/data/robots.json, problemsets.json and scene_meshes.json (vertices / triangles of the
scene collision meshes data/scenes/*/*.obj, input of the mesh -> SDF generator).  Nothing of the
reference's code is copied: only the numbers of data/robots/*/config.yaml, the sphere
visuals of the URDFs and the state lists / planner parameters of data/problemsets/*.py.

Sphere offsets replace what the reference obtains from pybullet
(gpflow_vgpmp/utils/robot.py:482-499: getVisualShapeData()[5] = the visual origin expressed
in the link's INERTIAL frame; confirmed for UR10 by tests/test_robot.py:62-67) followed by the
per-robot hand corrections of gpflow_vgpmp/utils/sampler.py:68-101.  Only Franka (all inertial
origins zero) is exact from the URDF alone; the other robots follow the documented pybullet
convention and are UNVERIFIED against pybullet itself (not installable here).
"""
import importlib.util
import json
import math
import sys
import types
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np
import yaml

ROBOT_URDF = {"franka": "franka_spheres.urdf", "wam": "wam.urdf", "ur10": "ur10.urdf", "kuka": "kuka.urdf"}


def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def floats(s, default):
    return [float(v) for v in s.split()] if s else list(default)


def urdf_spheres(path):
    """[(link_name, [xyz in inertial frame, ...])] in URDF link order, links with spheres only."""
    root = ET.parse(path).getroot()
    out = []
    for link in root.findall("link"):
        ino = link.find("inertial/origin")
        ixyz = np.array(floats(ino.get("xyz") if ino is not None else None, (0, 0, 0)))
        irot = rpy_matrix(floats(ino.get("rpy") if ino is not None else None, (0, 0, 0)))
        sph = []
        for vis in link.findall("visual"):
            if vis.find("geometry/sphere") is None:
                continue
            o = vis.find("origin")
            vxyz = np.array(floats(o.get("xyz") if o is not None else None, (0, 0, 0)))
            sph.append((irot.T @ (vxyz - ixyz)).tolist())
        if sph:
            out.append((link.get("name"), sph))
    return out


def urdf_inertial_origins(path):
    """{link name: xyz of <inertial><origin>} for every link (zeros when absent): what pybullet reports as the link's
    localInertialFramePosition (utils/robot.py:366-373 joint_link_offsets; KAT tests/test_robot.py:62-67)."""
    root = ET.parse(path).getroot()
    out = {}
    for link in root.findall("link"):
        ino = link.find("inertial/origin")
        out[link.get("name")] = floats(ino.get("xyz") if ino is not None else None, (0, 0, 0))
    return out


def offset_branch(robot, index):
    """Which branch of gpflow_vgpmp/utils/sampler.py:68-101 sphere `index` takes (documentation of the table below)."""
    if robot == "wam":
        return ("index<8" if index < 8 else "8<index<=12" if 8 < index <= 12 else "index>14" if index > 14
                else "index==8" if index == 8 else "else(13,14)")
    if robot == "ur10":
        return "0<index<7" if 0 < index < 7 else "else"
    if robot == "kuka":
        for lo, hi in ((2, 5), (5, 8), (8, 11), (11, 15), (15, 17), (17, 20)):
            if lo <= index < hi:
                return f"{lo}<=index<{hi}"
        return "else"
    return "identity"


def corrected_offset(robot, index, off):
    """Numbers of gpflow_vgpmp/utils/sampler.py:68-101 (per-index sphere offset fixes)."""
    x, y, z = off
    if robot == "wam":
        if index < 8:
            return [x - 0.045, -y, z]
        if 8 < index <= 12:
            return [x + 0.045, -y - 0.05, z]
        if index > 14:
            return [x, y, z]
        if index == 8:
            return [0.0, 0.0, 0.0]
        return [x, -y, z]
    if robot == "ur10":
        if 0 < index < 7:
            return [z, x, y + 0.163941 + 0.05]
        return [z, x, y]
    if robot == "kuka":
        if 1 < index < 5:
            return [x, -z + 0.18, y]
        if 5 <= index < 8:
            return [x, z, y]
        if 8 <= index < 11:
            return [x, z - 0.18, -y]
        if 11 <= index < 15:
            return [x, -z, y]
        if 15 <= index < 17:
            return [x, z + 0.1, y - 0.06]
        if 17 <= index < 20:
            return [x, z - 0.07, y]
        return [x, y, z]
    return [x, y, z]


SCENE_MESH = {"industrial": "industrial/industrial-acd.obj", "bookshelves": "bookshelves/bookshelves_center.obj",
              "boxes": "boxes/boxes-acd.obj", "lab": "lab/lab.obj"}


def read_obj(path):
    """Vertices, triangles (0-based) and the index of the `o` part each triangle belongs to."""
    verts, faces, part, cur = [], [], [], -1
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "o":
            cur += 1
        elif t[0] == "v":
            verts.append([float(v) for v in t[1:4]])
        elif t[0] == "f":
            idx = [int(v.split("/")[0]) - 1 for v in t[1:]]
            for k in range(1, len(idx) - 1):                      # fan-triangulate polygons
                faces.append([idx[0], idx[k], idx[k + 1]])
                part.append(max(cur, 0))
    return verts, faces, part


def load_problemset(ref, robot):
    stub = types.ModuleType("problemset")
    stub.AbstractProblemset = type("AbstractProblemset", (), {})
    sys.modules["problemset"] = stub
    spec = importlib.util.spec_from_file_location(f"_ps_{robot}", ref / "data/problemsets" / f"{robot}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Problemset


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    out_dir = Path(__file__).resolve().parent.parent / "vgpmp_amd" / "data"
    out_dir.mkdir(parents=True, exist_ok=True)
    robots, psets = {}, {}
    for robot, urdf in ROBOT_URDF.items():
        cfg = yaml.safe_load(open(ref / "data/robots" / robot / "config.yaml"))
        links = urdf_spheres(ref / "data/robots" / robot / urdf)
        raw = [o for _, sph in links for o in sph]
        assert len(raw) == len(cfg["radius"]), (robot, len(raw), len(cfg["radius"]))
        robots[robot] = dict(
            dof=cfg["dof"], craig_dh_convention=bool(cfg["craig_dh_convention"]),
            dh_parameters=np.array(cfg["dh_parameters"], dtype=float).reshape(-1, 3).tolist(),
            twist=[float(v) for v in cfg["twist"]], fk_slice=list(cfg["fk_slice"]),
            num_frames_for_spheres=cfg["num_frames_for_spheres"],
            joint_limits=np.array(cfg["joint_limits"], dtype=float).reshape(-1, 2).tolist(),
            velocity_limits=np.array(cfg["velocity_limits"], dtype=float).reshape(-1, 2).tolist(),
            radius=[float(v) for v in cfg["radius"]],
            sphere_links=[name for name, _ in links],
            num_spheres_per_link=[len(s) for _, s in links],
            sphere_offsets_urdf=raw,
            sphere_offsets=[corrected_offset(robot, i, o) for i, o in enumerate(raw)],
            sphere_offset_branch=[offset_branch(robot, i) for i in range(len(raw))],
            num_spheres_config=int(cfg["num_spheres"]),
            inertial_origins=urdf_inertial_origins(ref / "data/robots" / robot / urdf),
            joint_names=cfg["joint_names"], default_pose=[float(v) for v in cfg["default_pose"]],
            active_joints=cfg["active_joints"], active_links=cfg["active_links"],
            link_name_base=cfg["link_name_base"], link_name_wrist=cfg["link_name_wrist"])
        ps = load_problemset(ref, robot)
        psets[robot] = {}
        for name in ("industrial", "bookshelves", "lab", "boxes"):
            try:
                n, states = ps.states(name)
            except ValueError:
                continue
            entry = dict(states=states)
            for key, fn in (("planner_params", ps.planner_params), ("pos_and_orn", ps.pos_and_orn),
                            ("object_positions", ps.object_positions)):
                try:
                    entry[key] = fn(name)
                except ValueError:
                    pass
            psets[robot][name] = entry
    meshes = {}
    for name, rel in SCENE_MESH.items():
        v, f, part = read_obj(ref / "data/scenes" / rel)
        meshes[name] = dict(source=rel, vertices=v, faces=f, part=part)
    json.dump(meshes, open(out_dir / "scene_meshes.json", "w"))
    json.dump(robots, open(out_dir / "robots.json", "w"), indent=1)
    json.dump(psets, open(out_dir / "problemsets.json", "w"), indent=1)
    print("wrote", out_dir / "robots.json", out_dir / "problemsets.json")


if __name__ == "__main__":
    main()
