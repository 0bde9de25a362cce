"""Host-side environment objects of the reference's boundary (no GPU arithmetic here):

  ParameterLoader     utils/parameter_loader.py:18-172   parameters.yaml -> robot/scene/planner/trainable/graphics params
  Simulation          utils/simulation.py:95-136         connection holder (headless here)
  Scene               utils/scene.py:9-138               keeps the scene pose the planner reads (miscellaneous.py:166)
  Robot               utils/robot.py:54-564              robot tables + joint state; success check without physics
  SignedDistanceField utils/sdf_utils.py:24-215          grid container; lookups run on the device table
  Sampler             utils/sampler.py:19-244            FK interface (device kernel behind it)
  SimulationManager   utils/simulation_manager.py:25-157 wires the above together

`parameters.yaml` keeps the reference's schema (a list of four single-key dicts).  Robot tables and
problem sets come from vgpmp_amd/data (or from a `data/` directory next to parameters.yaml when the
reference's own checkout is used as the root).  Missing `.sdf` blobs are replaced by a synthetic scene.
"""
from __future__ import annotations

import copy
import itertools
import os
import sys
import warnings
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import yaml

from .. import robots as robot_tables
from .. import scenes
from .shims import p


def get_root_package_path() -> str:
    """utils/miscellaneous.py:366: directory that holds parameters.yaml (override: VGPMP_ROOT)."""
    return os.environ.get("VGPMP_ROOT") or str(Path(__file__).resolve().parents[2])


# --------------------------------------------------------------------------------------------------------
class ParameterLoader:
    def __init__(self):
        self.is_initialized = False
        self._params = None
        self.trainable_params = self.planner_params = self.graphics_params = None
        self.robot_params = self.scene_params = None
        self.root_path = Path(get_root_package_path())
        self.data_dir_path = self.root_path / "data"

    @property
    def params(self) -> dict:
        assert self._params is not None, "Parameter Loader must be initialized before it can be accessed"
        return self._params

    def initialize(self, file_path: Optional[Path] = None, params: Optional[list] = None):
        if file_path is not None:
            self.load_parameter_file(Path(file_path))
        else:
            assert params is not None, "Either parameter_file_path or params must be specified"
            self._params = self.set_params(params)

    def load_parameter_file(self, path: Path):
        try:
            with open(path, "r") as stream:
                params = yaml.safe_load(stream)
        except FileNotFoundError:
            print(f"[Error]: Parameters file {path} could not be found")
            sys.exit("[EXIT]: System will exit, please provide a parameter file and try again")
        self._params = self.set_params(params)

    def set_params(self, params):
        robot_params, scene_params, trainable_params, graphic_params = params      # positional, as the reference
        self.scene_params = copy.deepcopy(scene_params["scene"])
        self.robot_params = dict(robot_params["robot"])
        self.trainable_params = dict(trainable_params["trainable_params"])
        self.graphics_params = dict(graphic_params["graphics"])
        self.get_robot_config(self.robot_params)
        self.get_scene_config(self.scene_params)
        self.is_initialized = True
        return {"robot_params": self.robot_params, "scene_params": self.scene_params,
                "planner_params": self.planner_params, "trainable_params": self.trainable_params,
                "graphics_params": self.graphics_params}

    def get_robot_config(self, robot_params: dict):
        name = robot_params["robot_name"]
        if name not in robot_tables.AVAILABLE_ROBOTS:
            print("Robot not available. Check params file and try again... The simulator will now exit.")
            sys.exit(-1)
        local = self.data_dir_path / "robots" / name / "config.yaml"
        if local.exists():                       # a reference checkout is the root: read its own config
            with open(local, "r") as fh:
                cfg = yaml.safe_load(fh)
            cfg["urdf_path"] = local.parent / cfg["path"]
        else:
            t = robot_tables.load_robot(name).raw
            cfg = {"radius": t["radius"], "num_spheres": len(t["radius"]), "joint_names": t["joint_names"],
                   "default_pose": t["default_pose"], "active_links": t["active_links"],
                   "active_joints": t["active_joints"], "link_name_base": t["link_name_base"],
                   "link_name_wrist": t["link_name_wrist"], "path": None, "urdf_path": None,
                   "joint_limits": [v for row in t["joint_limits"] for v in row],
                   "velocity_limits": [v for row in t["velocity_limits"] for v in row],
                   "dh_parameters": [v for row in t["dh_parameters"] for v in row], "twist": t["twist"],
                   "dof": t["dof"], "craig_dh_convention": t["craig_dh_convention"],
                   "num_frames_for_spheres": t["num_frames_for_spheres"], "fk_slice": t["fk_slice"]}
        self.robot_params = {**cfg, **robot_params}           # main-file keys win (parameter_loader.py:99)

    def get_scene_config(self, scene_params: dict):
        assert scene_params.get("benchmark") is not None, "Benchmark attribute is not specified"
        assert type(scene_params["benchmark"]) is bool, "Benchmark attribute must be a boolean"
        if scene_params["benchmark"] is False:
            attrs = scene_params["non_benchmark_attributes"]
            states, planner_params = attrs["states"], attrs["planner_params"]
            robot_pos_and_orn = tuple(attrs["robot_pos_and_orn"])
            object_positions = [scene_params.get("position", [0.0, 0.0, 0.0])]
        else:
            ps = robot_tables.load_problemset(self.robot_params["robot_name"],
                                              scene_params["benchmark_attributes"]["problemset_name"])
            states, planner_params = ps.states, ps.planner_params
            robot_pos_and_orn, object_positions = ps.robot_pos_and_orn, ps.object_positions
        queries = list(itertools.combinations(states, 2))
        print(f"There are {len(states)} total robot positions and a total of {len(queries)} problems")
        scene_params["queries"] = queries
        scene_params["robot_pos_and_orn"] = robot_pos_and_orn
        scene_params["object_positions"] = object_positions
        scene_params["objects_path"] = []
        env_path, sdf_path = self.get_assets_path(scene_params["environment_name"],
                                                  scene_params["environment_file_name"], scene_params["sdf_file_name"])
        scene_params["sdf_path"], scene_params["environment_path"] = sdf_path, env_path
        scene_params.pop("benchmark_attributes", None)
        scene_params.pop("non_benchmark_attributes", None)
        self.planner_params = planner_params

    def get_assets_path(self, environment_name, environment_file_name, sdf_file_name):
        d = self.data_dir_path / "scenes" / environment_name
        sdf = d / (sdf_file_name + ".sdf")
        if not sdf.exists():
            warnings.warn(f"{sdf} is missing (the reference ships its .sdf grids as large blobs); the grid is "
                          "generated from the scene's collision mesh on the device (or a synthetic scene if unknown)")
            sdf = None
        return d / (environment_file_name + ".urdf"), sdf


# --------------------------------------------------------------------------------------------------------
class _Thread:
    def __init__(self):
        self.client = None
        self._alive = False

    def is_alive(self):            # the reference's thread is never start()ed: always False (simulation.py:134-136)
        return self._alive


class Simulation:
    def __init__(self, config: ParameterLoader):
        assert config.is_initialized, "ParameterLoader is not initialized"
        self.graphics_params = config.graphics_params
        self.simulation_thread = _Thread()
        self.is_initialized = None

    def initialize(self):
        self.simulation_thread.client = p.connect(p.DIRECT)
        self.is_initialized = True

    def check_simulation_thread_health(self):
        return self.simulation_thread.is_alive()

    def stop_simulation_thread(self):
        p.disconnect()
        self.simulation_thread.client = None


class Scene:
    def __init__(self, config: ParameterLoader = None, client: int = None):
        self.config = getattr(config, "scene_params", None)
        self.client, self.position, self.orientation, self.is_initialized = client, None, None, None
        self.objects: List[dict] = []

    def initialize(self, client):
        assert self.config is not None, "Scene config is not initialized"
        self.client = client
        self.objects = [{"name": "plane"}, {"name": "environment", "path": self.config.get("environment_path")}]
        self.position, self.orientation = self.config["position"], self.config["orientation"]
        self.is_initialized = True


class SignedDistanceField:
    """data[x, y, z] float64 grid + origin + delta (utils/sdf_utils.py:24-44)."""

    def __init__(self, data: np.ndarray, origin: np.ndarray, delta: float):
        self.data = np.asarray(data, dtype=np.float64)
        self.nx, self.ny, self.nz = self.data.shape
        self.origin = np.asarray(origin, dtype=np.float64)
        self.delta = float(delta)
        self.min_coords = self.origin
        self.max_coords = self.origin + self.delta * np.array(self.data.shape)
        self._device = None                      # DeviceScene bound by the likelihood

    @classmethod
    def from_sdf(cls, sdf_file):
        return cls(*scenes.read_sdf(str(sdf_file)))

    @classmethod
    def synthetic(cls, **kwargs):
        return cls(*scenes.synthetic_boxes_sdf(**kwargs))

    def dump_sdf(self, path):
        scenes.write_sdf(str(path), self.grid)

    @property
    def grid(self):
        return self.data, self.origin, self.delta

    def bind(self, device_scene):
        self._device = device_scene

    def _query(self, rel_pos):
        import torch
        if self._device is None:
            raise RuntimeError("SignedDistanceField is not bound to a DeviceScene yet (build the likelihood first)")
        rel = torch.as_tensor(np.asarray(rel_pos, dtype=np.float64))
        idx, dist, grad = self._device.sdf_query(rel.reshape(-1, 3))
        shape = rel.shape[:-1]
        return idx.reshape(shape + (3,)), dist.reshape(shape), grad.reshape(shape + (3,))

    def get_distance_tf(self, rel_pos):           # utils/sdf_utils.py:73-76
        return self._query(rel_pos)[1]

    def get_distance_grad_tf(self, rel_pos):      # utils/sdf_utils.py:100-136 (zero components -> 0.1)
        return self._query(rel_pos)[2]

    get_distance = get_distance_tf
    get_distance_grad = get_distance_grad_tf


class Robot:
    """Robot tables and joint state.  `move_to_ee_config` replaces the simulated execution of the
    reference (utils/robot.py:416-480) by a sphere-vs-SDF clearance check of the planned path."""

    def __init__(self, config: ParameterLoader, simulation: Simulation):
        prm = config.robot_params
        self.name = prm["robot_name"]
        pos, orn = config.scene_params["robot_pos_and_orn"]
        self.position, self.orientation = list(pos), list(orn)
        self.spec = robot_tables.load_robot(self.name, pos, orn)
        self.dof = self.spec.dof
        self.robot_model = p.register_body(pos, orn)
        self.sphere_radii = list(self.spec.sphere_radii)
        self.num_spheres = self.spec.num_spheres
        self.num_spheres_per_link = list(self.spec.num_spheres_per_link)
        self.num_frames_for_spheres = self.spec.num_frames_for_spheres
        self.sphere_offsets = self.spec.sphere_offsets
        self.joint_limits = [float(v) for v in self.spec.joint_limits.reshape(-1)]
        self.velocity_limits = [float(v) for v in self.spec.velocity_limits.reshape(-1)]
        self.base_pose = self.spec.base_pose
        self.fk_slice = list(self.spec.fk_slice)
        self.curr_joint_config = np.zeros(self.dof)
        self.clearance_fn = None                  # set by solve_planning_problem
        self.is_initialized = False

    def initialise(self, default_robot_pos_and_orn=None, joint_names=None, default_pose=None, benchmark=True):
        self.is_initialized = True

    def get_base_pose(self):
        pos, orn = p.getBasePositionAndOrientation(self.robot_model)
        return robot_tables.base_pose_matrix(pos, orn)

    def set_current_joint_config(self, config):
        self.curr_joint_config = np.asarray(config, dtype=np.float64).reshape(-1)

    def get_current_joint_config(self):
        return self.curr_joint_config

    def set_joint_motor_control(self, position, kp=300, kv=0.5):
        pass

    def enable_collision_active_links(self, mask: int = 0):
        pass

    def move_to_ee_config(self, joint_config, margin: float = 0.0) -> bool:
        path = np.asarray(joint_config, dtype=np.float64)
        low, high = self.spec.low, self.spec.high
        if (path < low - 1e-9).any() or (path > high + 1e-9).any():
            return False
        if self.clearance_fn is None:
            return True
        ok = bool(self.clearance_fn(path) > margin)
        if ok:
            self.curr_joint_config = path[-1].copy()
        return ok


class Sampler:
    """FK interface of the likelihood (utils/sampler.py).  The arithmetic runs in fk_spheres_kernel."""

    def __init__(self, config: ParameterLoader, robot: Robot):
        self.robot, self.spec = robot, robot.spec
        self.name, self.dof = robot.name, robot.dof
        self.fk_slice = robot.fk_slice
        self.num_spheres_per_link = robot.num_spheres_per_link
        offs = np.tile(np.eye(4), (robot.num_spheres, 1, 1))
        offs[:, :3, 3] = robot.sphere_offsets
        self.sphere_offsets = offs                 # [P, 4, 4] pure translations, as the reference stores them
        self.base_pose = robot.base_pose[None]
        self._device = None

    def bind(self, device_scene):
        self._device = device_scene

    def _fk(self, q, want_frames):
        import torch
        if self._device is None:
            raise RuntimeError("Sampler is not bound to a DeviceScene yet (build the likelihood first)")
        return self._device.fk_spheres(torch.as_tensor(np.asarray(q, dtype=np.float32).reshape(1, self.dof)), want_frames)

    def forward_kinematics(self, thetas):
        """thetas [D, 1] -> cumulative frames [D+1, 4, 4] (utils/sampler.py:103-120)."""
        _, frames = self._fk(thetas, True)
        f = frames[0].cpu().numpy().astype(np.float64)
        out = np.tile(np.eye(4), (self.dof + 1, 1, 1))
        out[:, :3, :] = f
        return out

    def forward_kinematics_cost(self, joint_config):
        """joint_config [D, 1] -> sphere centres [P, 3] (utils/sampler.py:216-235)."""
        return self._fk(joint_config, False)[0]


class SimulationManager:
    def __init__(self, file_path=None, parameter_loader: ParameterLoader = None, simulation=None, scene=None,
                 robot=None, sdf=None, sampler=None):
        if file_path is not None:
            self._config = ParameterLoader()
            self._config.initialize(file_path=file_path)
        else:
            assert parameter_loader is not None and parameter_loader.is_initialized, \
                "Parameter Loader must be initialized if no parameter file is passed"
            self._config = parameter_loader
        self.simulation = simulation or Simulation(self._config)
        if not self.simulation.is_initialized:
            self.simulation.initialize()
        self.client = self.simulation.simulation_thread.client
        self.scene = scene or Scene(self._config)
        if not self.scene.is_initialized:
            self.scene.initialize(self.client)
        self.robot = robot or Robot(self._config, self.simulation)
        if not self.robot.is_initialized:
            sp = self.config["scene_params"]
            self.robot.initialise(sp["robot_pos_and_orn"], self.config["robot_params"].get("joint_names"),
                                  self.config["robot_params"].get("default_pose"), sp["benchmark"])
        self.sampler = sampler or Sampler(self._config, self.robot)
        if sdf is not None:
            self.sdf = sdf
        elif self.config["scene_params"]["sdf_path"] is not None:
            self.sdf = SignedDistanceField.from_sdf(self.config["scene_params"]["sdf_path"])
        else:
            try:                   # mesh -> SDF on the device (replaces utils/gen_sdf.py + external SDFGen)
                self.sdf = SignedDistanceField(*scenes.scene_sdf(self.config["scene_params"]["environment_name"]))
            except KeyError:
                self.sdf = SignedDistanceField.synthetic()

    @property
    def config(self) -> dict:
        return self._config.params

    def loop(self, planner=None):           # interactive debug loop of the reference: nothing to do headless
        return None
