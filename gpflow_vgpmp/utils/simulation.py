from vgpmp_amd.host.environment import Simulation  # noqa: F401
