from vgpmp_amd.host.environment import Scene  # noqa: F401
