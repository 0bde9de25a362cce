from vgpmp_amd.host.environment import SimulationManager  # noqa: F401
