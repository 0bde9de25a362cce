"""Stand-ins for the third-party names the reference's driver touches through
`from gpflow_vgpmp.utils.miscellaneous import *` (benchmarking.py:3,9,11,17,60-65,81,91):

  gpflow   only `gpflow.config.set_default_float`, `gpflow.config.Config`, `default_float`, `default_jitter`
  p        the handful of pybullet calls of the driver and of Robot, answered by a headless kinematic
           body registry (no physics, no rendering)

Nothing here computes on the hot path.
"""
from __future__ import annotations

import types

import numpy as np


# ---------------------------------------------------------------- gpflow ---------------------------------
class _GpflowConfig:
    def __init__(self):
        self._float = np.float64
        self._jitter = 1e-6

    def set_default_float(self, dtype):
        self._float = dtype

    def default_float(self):
        return self._float

    def default_jitter(self):
        return self._jitter

    class Config:                      # benchmarking.py:11 constructs one and drops it (a no-op there too)
        def __init__(self, **kwargs):
            self.__dict__.update(kwargs)


gpflow = types.SimpleNamespace(config=_GpflowConfig())
gpflow.default_float = gpflow.config.default_float
gpflow.default_jitter = gpflow.config.default_jitter


class Parameter:
    """Minimal value holder with a `trainable` flag (gpflow.Parameter look-alike: numpy(), assign())."""

    def __init__(self, value, trainable: bool = True, name: str = ""):
        self._value = np.array(value, dtype=np.float64)
        self.trainable = trainable
        self.name = name
        self.prior = None

    def numpy(self):
        return self._value

    def assign(self, value):
        self._value = np.array(value, dtype=np.float64).reshape(self._value.shape)

    def __array__(self, dtype=None):
        return self._value.astype(dtype) if dtype else self._value

    def __float__(self):
        return float(self._value)

    def __repr__(self):
        return f"Parameter({self.name}, trainable={self.trainable}, value={self._value!r})"


def set_trainable(obj, flag: bool) -> None:
    """gpflow.set_trainable on a Parameter or on any object holding Parameters."""
    if isinstance(obj, Parameter):
        obj.trainable = bool(flag)
        return
    for v in vars(obj).values():
        if isinstance(v, Parameter):
            v.trainable = bool(flag)


# ---------------------------------------------------------------- pybullet --------------------------------
class HeadlessBullet:
    """Registry of bodies with a base pose; enough for benchmarking.py and the headless Robot."""
    GUI, DIRECT, POSITION_CONTROL = 1, 2, 2

    def __init__(self):
        self._bodies = {}
        self._next = 0

    def connect(self, mode=None, **kwargs):
        return 0

    def disconnect(self, *a, **k):
        self._bodies.clear()

    def register_body(self, position=(0.0, 0.0, 0.0), orientation=(0.0, 0.0, 0.0, 1.0)) -> int:
        uid = self._next
        self._next += 1
        self._bodies[uid] = [tuple(float(v) for v in position), tuple(float(v) for v in orientation)]
        return uid

    def getBasePositionAndOrientation(self, body, physicsClientId=0):
        pos, orn = self._bodies[body]
        return pos, orn

    def resetBasePositionAndOrientation(self, body, pos, orn, physicsClientId=0):
        self._bodies[body] = [tuple(float(v) for v in pos), tuple(float(v) for v in orn)]

    # calls that only matter with a renderer / physics engine
    def resetDebugVisualizerCamera(self, *a, **k): pass
    def stepSimulation(self, *a, **k): pass
    def removeAllUserDebugItems(self, *a, **k): pass
    def addUserDebugLine(self, *a, **k): return -1
    def setJointMotorControl2(self, *a, **k): pass
    def configureDebugVisualizer(self, *a, **k): pass
    def setGravity(self, *a, **k): pass


p = HeadlessBullet()
