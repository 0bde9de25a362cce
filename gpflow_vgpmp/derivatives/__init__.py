"""derivatives/ of the reference (velocity-constrained kernel variant), computed on the device."""
from vgpmp_amd.host.derivatives import K_grad, K_grad_grad, velocity_kuu_kuf  # noqa: F401
