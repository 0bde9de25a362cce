from vgpmp_amd.host.environment import SignedDistanceField  # noqa: F401
