"""CPU tests of the host-side mirror of the reference's boundary (no GPU, no compute calls)."""
import os
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_driver_star_import_surface():
    """Names the reference's benchmarking.py takes from `from gpflow_vgpmp.utils.miscellaneous import *`
    (benchmarking.py:3,9,11,14,17,60-65,81,83,91,95)."""
    ns = {}
    exec("from gpflow_vgpmp.utils.miscellaneous import *", ns)
    for name in ("gpflow", "np", "p", "time", "os", "get_root_package_path", "solve_planning_problem"):
        assert name in ns, name
    ns["gpflow"].config.set_default_float(np.float64)
    ns["gpflow"].config.Config(jitter=1e-6)
    pb = ns["p"]
    body = pb.register_body((0, 0, 0.346), (0, 0, 0, 1))
    pos, orn = pb.getBasePositionAndOrientation(body)
    pb.resetBasePositionAndOrientation(body, (pos[0], pos[1], pos[2] - 0.346), orn)
    assert pb.getBasePositionAndOrientation(body)[0] == (0.0, 0.0, 0.0)
    pb.resetDebugVisualizerCamera(cameraDistance=2, cameraYaw=-75, cameraPitch=-45, cameraTargetPosition=[0, 0, 0])
    pb.stepSimulation(); pb.removeAllUserDebugItems()
    assert Path(ns["get_root_package_path"]()) == ROOT


def test_parameter_loader_schema_and_queries():
    from gpflow_vgpmp.utils.parameter_loader import ParameterLoader
    pl = ParameterLoader()
    with pytest.warns(UserWarning):                      # the .sdf blob is absent -> synthetic scene
        pl.initialize(file_path=ROOT / "parameters.yaml")
    cfg = pl.params
    assert set(cfg) == {"robot_params", "scene_params", "planner_params", "trainable_params", "graphics_params"}
    assert cfg["robot_params"]["robot_name"] == "franka" and cfg["robot_params"]["dof"] == 7
    assert len(cfg["scene_params"]["queries"]) == 36      # C(9, 2), utils/parameter_loader.py:138
    assert cfg["planner_params"]["num_steps"] == 200 and cfg["trainable_params"]["lengthscales"] is True
    assert "benchmark_attributes" not in cfg["scene_params"]


def test_parameter_loader_non_benchmark_and_errors(tmp_path):
    import yaml
    from gpflow_vgpmp.utils.parameter_loader import ParameterLoader
    params = yaml.safe_load(open(ROOT / "parameters.yaml"))
    params[1]["scene"]["benchmark"] = False
    pl = ParameterLoader()
    with pytest.warns(UserWarning):
        pl.initialize(params=params)
    assert len(pl.params["scene_params"]["queries"]) == 1 and pl.params["planner_params"]["num_inducing"] == 10
    params[0]["robot"]["robot_name"] = "pr2"
    with pytest.raises(SystemExit):
        ParameterLoader().initialize(params=params)
    with pytest.raises(SystemExit):
        ParameterLoader().initialize(file_path=tmp_path / "nope.yaml")


def test_init_trainset_matches_reference_grid():
    from gpflow_vgpmp.utils.miscellaneous import init_trainset
    X, y, Xnew = init_trainset(50, 100, 7, 7, np.arange(7.0), -np.arange(7.0), scale=1)
    assert X.shape == (50, 7) and Xnew.shape == (100, 7) and y.shape == (2, 7)
    np.testing.assert_allclose(X[:, 3], np.linspace(0, 1, 50))
    assert (X == X[:, :1]).all() and np.array_equal(y[1], -np.arange(7.0))


def test_fill_order_independent_variables_and_problemsets():
    from vgpmp_amd import robots
    ps = robots.load_problemset("wam", "industrial")
    assert len(ps.queries) == 36 and ps.planner_params["num_inducing"] == 24
    assert ps.robot_pos_and_orn[0] == [0.0, 0.0, 0.346]
    with pytest.raises(ValueError):
        robots.load_problemset("franka", "kitchen")
    arm = robots.synthetic_arm(14)
    assert arm.dof == 14 and arm.num_spheres == 45 and list(arm.sphere_frame) == sorted(arm.sphere_frame)


def test_robot_tables_pinned_on_the_reference_known_answers():
    """pybullet-free robot tables.  What the reference's own tests hold for them:
    tests/test_robot.py:62-67 -- pybullet's inertial-frame offsets of UR10's active links -- must equal the URDF inertial
    origins the sphere offsets here are expressed against (utils/robot.py:482-499 reports a sphere visual's origin in the
    link's INERTIAL frame); :70-73 the base pose of the benchmark orientation.  Counts against the robots' config.yaml."""
    import json
    from pathlib import Path
    from vgpmp_amd import robots as rb
    kat = json.load(open(Path(__file__).resolve().parent / "golden" / "ur10_inertial_kat.json"))
    tab = json.load(open(Path(rb.__file__).resolve().parent / "data" / "robots.json"))
    ur = tab["ur10"]
    got = [ur["inertial_origins"][name] for name in ur["active_links"]]
    assert got == kat["ur10_joint_link_offsets"]
    np.testing.assert_allclose(rb.base_pose_matrix((0, 0, 0), (0, 0, -1, 0)), np.array(kat["base_pose_benchmark"]), atol=1e-15)
    counts = {"franka": [2, 3, 3, 4, 4, 7, 3, 11], "wam": [9, 4, 1, 11], "ur10": [1, 6, 7, 1, 2], "kuka": [2, 3, 3, 3, 4, 2, 3, 1]}
    for name, t in tab.items():
        assert t["num_spheres_per_link"] == counts[name]
        assert sum(t["num_spheres_per_link"]) == t["num_spheres_config"] == len(t["radius"]) == len(t["sphere_offsets"])
        assert len(t["num_spheres_per_link"]) == t["num_frames_for_spheres"] == len(t["fk_slice"])
        # a sphere's stored offset = R_inertial^T (visual origin - inertial origin), then the per-index correction of
        # utils/sampler.py:68-101; UR10's sphere links have identity inertial rotations, so the first step is a subtraction
        spec = rb.load_robot(name)
        assert spec.num_spheres == t["num_spheres_config"] and list(spec.sphere_frame) == sorted(spec.sphere_frame)
    # which branch of sampler.get_mat every sphere index takes (utils/sampler.py:68-101), robot by robot
    branches = {
        "franka": {"identity": list(range(37))},
        "ur10": {"else": [0] + list(range(7, 17)), "0<index<7": list(range(1, 7))},
        "wam": {"index<8": list(range(8)), "index==8": [8], "8<index<=12": [9, 10, 11, 12], "else(13,14)": [13, 14],
                "index>14": list(range(15, 25))},
        "kuka": {"else": [0, 1, 20], "2<=index<5": [2, 3, 4], "5<=index<8": [5, 6, 7], "8<=index<11": [8, 9, 10],
                 "11<=index<15": [11, 12, 13, 14], "15<=index<17": [15, 16], "17<=index<20": [17, 18, 19]},
    }
    for name, want in branches.items():
        got = {}
        for i, b in enumerate(tab[name]["sphere_offset_branch"]):
            got.setdefault(b, []).append(i)
        assert got == want, name
    # UR10 spheres 1..6 (the upper arm: link with inertial origin z = 0.306) carry the +0.163941 + 0.05 shift of :83-84
    raw, cor = np.array(ur["sphere_offsets_urdf"]), np.array(ur["sphere_offsets"])
    np.testing.assert_allclose(cor[1:7], np.stack([raw[1:7, 2], raw[1:7, 0], raw[1:7, 1] + 0.163941 + 0.05], 1), atol=1e-15)
    np.testing.assert_allclose(cor[7:], np.stack([raw[7:, 2], raw[7:, 0], raw[7:, 1]], 1), atol=1e-15)
