from vgpmp_amd.host.environment import ParameterLoader  # noqa: F401
