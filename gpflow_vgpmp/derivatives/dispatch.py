"""derivatives/dispatch.py of the reference: the K_grad / K_grad_grad dispatchers."""
from vgpmp_amd.host.derivatives import K_grad, K_grad_grad  # noqa: F401
