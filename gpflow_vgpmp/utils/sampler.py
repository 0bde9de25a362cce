from vgpmp_amd.host.environment import Sampler  # noqa: F401
