#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own importable code in the build
container (it cannot travel to the GPU box; only the resulting numbers are committed).

    python tests/golden/make_golden.py [/root/reference]

What is executed from the reference:
  * gpflow_vgpmp/utils/robot_mixin.py  -- pure numpy; RobotMixin.forward_kinematics (:32-58)
    and the scalar DH builders (:60-112).  Loaded by file path so the package __init__
    (which imports TensorFlow/pybullet) is not triggered.
  * gpflow_vgpmp/utils/sdf_utils.py    -- its numpy twins `_rel_pos_to_idxes` (:56-60),
    `get_distance` (:68-71), `get_distance_grad` (:78-98) and the text parser `from_sdf`
    (:195-210).  The module imports tensorflow at the top and builds tf constants in
    __init__, so `tensorflow` is replaced by an inert placeholder module and `np.int`
    (removed from NumPy >= 1.24) is aliased to `int`; no TensorFlow arithmetic is executed or
    emulated -- only the numpy methods are called.
Written: fk_reference.npz, sdf_reference.npz (+ sdf_small.sdf, a grid file in the reference's
text format written by the oracle and parsed back by the reference's from_sdf).
ur10_dh_kat.json holds the six literal matrices of the reference's tests/test_robot.py:14-42
(data of the reference's own test; transcribed by hand, checked here against its DH builder).
"""
import importlib.util
import json
import sys
import types
from pathlib import Path
from unittest import mock

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import vgpmp_oracle as orc  # noqa: E402
from vgpmp_amd import robots as rb  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    rng = np.random.default_rng(20240607)

    # ---------------- FK from the reference's numpy RobotMixin ----------------
    mixin = load_by_path("_ref_robot_mixin", ref / "gpflow_vgpmp/utils/robot_mixin.py")
    fk = {}
    for name, pos, orn in (("franka", (0, 0, 0), (0, 0, 0, 1)), ("wam", (0, 0, 0.346), (0, 0, 0, 1)),
                           ("ur10", (0, 0, 0), (0, 0, -1, 0)), ("kuka", (0.1, -0.2, 0.3), (0, 0, 0.38268343, 0.92387953))):
        spec = rb.load_robot(name, pos, orn)
        r = mixin.RobotMixin(name, spec.dh.reshape(-1).tolist(), spec.dof, spec.twist.tolist(),
                             spec.fk_slice.tolist(), spec.craig, spec.joint_limits.reshape(-1).tolist(),
                             spec.velocity_limits.reshape(-1).tolist(), base_pose=spec.base_pose)
        qs = np.concatenate([np.full((1, spec.dof), 0.1), np.zeros((1, spec.dof)),
                             rng.uniform(spec.low, spec.high, (6, spec.dof))])
        frames = np.stack([r.forward_kinematics(q.reshape(-1, 1)) for q in qs])
        fk[f"{name}_q"] = qs
        fk[f"{name}_frames"] = frames
        fk[f"{name}_base"] = spec.base_pose
    np.savez_compressed(HERE / "fk_reference.npz", **fk)

    # UR10 DH known answers of tests/test_robot.py:14-42 vs the reference's own scalar builder
    kat = json.load(open(HERE / "ur10_dh_kat.json"))
    ur = rb.load_robot("ur10")
    r = mixin.RobotMixin("ur10", ur.dh.reshape(-1).tolist(), 6, [0.0] * 6, ur.fk_slice.tolist(), False,
                         ur.joint_limits.reshape(-1).tolist(), ur.velocity_limits.reshape(-1).tolist())
    for i, m in enumerate(kat["matrices"]):
        got = r.get_transform_matrix_scalar(0.0, *ur.dh[i])
        assert np.allclose(got, np.array(m), atol=5e-8), (i, got, m)

    # ---------------- SDF numpy twins of the reference ----------------
    np.int = int  # noqa: removed alias still used by sdf_utils.py:57-58,80-81
    with mock.patch.dict(sys.modules, {"tensorflow": mock.MagicMock(name="tensorflow-placeholder")}):
        sdfmod = load_by_path("_ref_sdf_utils", ref / "gpflow_vgpmp/utils/sdf_utils.py")
        data = rng.normal(0.0, 0.2, (9, 8, 7))
        data[2, 3, :] = data[4, 3, :]          # exact-zero x-gradient cells at i=3
        origin = np.array([-0.31, -0.27, 0.05])
        delta = 0.073
        ref_sdf = sdfmod.SignedDistanceField(data.copy(), origin.copy(), delta)
        pos = rng.uniform(origin - 0.2, origin + delta * np.array(data.shape) + 0.2, (400, 3))
        pos[:8] = origin + delta * np.array([[0, 0, 0], [1, 1, 1], [8, 7, 6], [9, 8, 7],
                                             [3, 3, 2], [3.999999, 3, 2], [-0.5, 2, 2], [2, 2, -1e-9]])
        idx = ref_sdf._rel_pos_to_idxes(pos)
        dist = ref_sdf.get_distance(pos)
        grad = ref_sdf.get_distance_grad(pos)
        grid = orc.SDFGrid(data, origin, delta)
        orc.write_sdf_text(str(HERE / "sdf_small.sdf"), grid)
        parsed = sdfmod.SignedDistanceField.from_sdf(str(HERE / "sdf_small.sdf"))
        assert np.array_equal(parsed.data, data), "oracle writer / reference parser axis order"
    np.savez_compressed(HERE / "sdf_reference.npz", data=data, origin=origin, delta=delta, pos=pos,
                        idx=idx.astype(np.int64), dist=dist, grad=grad,
                        parsed_origin=parsed.origin, parsed_delta=parsed.delta)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
