from vgpmp_amd.host.model import prior_kl  # noqa: F401
