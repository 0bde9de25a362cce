from vgpmp_amd.host.model import VGPMP  # noqa: F401
