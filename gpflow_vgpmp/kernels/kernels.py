from vgpmp_amd.host.model import (FirstOrderKernelDerivativeSeparateIndependent, Matern52,  # noqa: F401
                                  VanillaConditioningSeparateIndependent, VanillaConditioningSharedIndependent)
