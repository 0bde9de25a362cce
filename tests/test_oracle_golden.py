"""Pin the oracle against vectors produced by the reference's own code (tests/golden/make_golden.py)
and the known answers held by the reference's tests."""
import json
from pathlib import Path

import numpy as np

from oracle import vgpmp_oracle as orc
from vgpmp_amd import robots as rb
from vgpmp_amd import scenes
from helpers import oracle_robot

GOLD = Path(__file__).resolve().parent / "golden"


def test_ur10_dh_known_answers():
    """reference tests/test_robot.py:14-42: six literal classic-DH matrices at theta = 0."""
    kat = json.load(open(GOLD / "ur10_dh_kat.json"))["matrices"]
    ur = rb.load_robot("ur10")
    got = orc.dh_matrix_classic(np.zeros(6), ur.dh[:, 0], ur.dh[:, 1], ur.dh[:, 2])
    assert got.shape == (6, 4, 4)
    np.testing.assert_allclose(got, np.array(kat), atol=5e-8)


def test_fk_matches_reference_numpy_fk():
    """reference robot_mixin.py:32-58 executed by make_golden.py (franka=Craig, others classic)."""
    z = np.load(GOLD / "fk_reference.npz")
    poses = {"franka": ((0, 0, 0), (0, 0, 0, 1)), "wam": ((0, 0, 0.346), (0, 0, 0, 1)),
             "ur10": ((0, 0, 0), (0, 0, -1, 0)), "kuka": ((0.1, -0.2, 0.3), (0, 0, 0.38268343, 0.92387953))}
    for name, (pos, orn) in poses.items():
        spec = rb.load_robot(name, pos, orn)
        np.testing.assert_allclose(spec.base_pose, z[f"{name}_base"], atol=1e-15)
        frames = orc.forward_kinematics(oracle_robot(spec), z[f"{name}_q"])
        np.testing.assert_allclose(frames, z[f"{name}_frames"], rtol=0, atol=1e-13)


def test_ur10_base_pose_for_flipped_quaternion():
    """reference tests/test_robot.py:70-73: quaternion (0,0,-1,0) -> diag(-1,-1,1)."""
    T = rb.base_pose_matrix((0, 0, 0), (0, 0, -1, 0))
    np.testing.assert_allclose(T, np.diag([-1.0, -1.0, 1.0, 1.0]), atol=1e-15)


def test_franka_frame_origins_at_zero():
    """SURVEY 8c KAT (4): Craig DH of data/robots/franka/config.yaml:76-90 at q = 0."""
    spec = rb.load_robot("franka")
    o = orc.forward_kinematics(oracle_robot(spec), np.zeros(7))[:, :3, 3]
    want = np.array([[0, 0, 0], [0, 0, .333], [0, 0, .333], [0, 0, .649], [.0825, 0, .649],
                     [0, 0, 1.033], [0, 0, 1.033], [.088, 0, 1.033]])
    np.testing.assert_allclose(o, want, atol=2e-6)
    assert spec.num_spheres == 37 and list(spec.num_spheres_per_link) == [2, 3, 3, 4, 4, 7, 3, 11]


def test_sdf_lookup_matches_reference_numpy_twins():
    """reference sdf_utils.py:56-60,68-71,78-98 executed by make_golden.py: bit-exact indices and values."""
    z = np.load(GOLD / "sdf_reference.npz")
    grid = orc.SDFGrid(z["data"], z["origin"], float(z["delta"]))
    idx = orc.sdf_index(grid, z["pos"])
    assert np.array_equal(idx, z["idx"])
    assert np.array_equal(orc.sdf_distance(grid, z["pos"]), z["dist"])
    g = orc.sdf_gradient(grid, z["pos"], replace_zero=False)
    assert np.array_equal(g, z["grad"])
    # TF path of the likelihood: exact zeros become 0.1 (sdf_utils.py:121-135)
    g_tf = orc.sdf_gradient(grid, z["pos"], replace_zero=True)
    assert (z["grad"] == 0).any(), "fixture must exercise the zero-gradient replacement"
    assert np.array_equal(g_tf, np.where(z["grad"] == 0, 0.1, z["grad"]))
    # the precomputed per-voxel table answers every query identically
    tab = orc.sdf_gradient_table(grid)
    q = tab[idx[:, 0], idx[:, 1], idx[:, 2]]
    assert np.array_equal(q[:, 0], z["dist"]) and np.array_equal(q[:, 1:], g_tf)


def test_sdf_text_format_roundtrip():
    """The fixture file was written by the oracle and parsed by the reference's from_sdf
    (asserted equal in make_golden.py); here oracle and product parsers read it back."""
    z = np.load(GOLD / "sdf_reference.npz")
    g = orc.parse_sdf_text(str(GOLD / "sdf_small.sdf"))
    assert np.array_equal(g.data, z["data"])
    np.testing.assert_allclose(g.origin, z["parsed_origin"], atol=0)
    assert g.delta == float(z["parsed_delta"])
    data, origin, delta = scenes.read_sdf(str(GOLD / "sdf_small.sdf"))
    assert np.array_equal(data, z["data"]) and np.array_equal(origin, g.origin) and delta == g.delta


def test_sdf_writer_roundtrip(tmp_path):
    grid = scenes.synthetic_boxes_sdf(n=10, delta=0.1, origin=(-0.5, -0.5, -0.5), seed=3)
    scenes.write_sdf(str(tmp_path / "a.sdf"), grid)
    data, origin, delta = scenes.read_sdf(str(tmp_path / "a.sdf"))
    assert np.array_equal(data, grid[0]) and np.array_equal(origin, grid[1]) and delta == grid[2]
    g = orc.parse_sdf_text(str(tmp_path / "a.sdf"))
    assert np.array_equal(g.data, grid[0])
