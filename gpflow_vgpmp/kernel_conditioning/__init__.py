from vgpmp_amd.host.model import K_conditioned  # noqa: F401
