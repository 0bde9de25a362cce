from vgpmp_amd.host.model import VariationalMonteCarloLikelihood  # noqa: F401
