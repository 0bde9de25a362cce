"""Kuu / Kuf / Kfu of the conditioned inducing set (covariances/ of the reference)."""
from vgpmp_amd.host.model import Kfu, Kuf, Kuu  # noqa: F401
