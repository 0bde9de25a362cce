from vgpmp_amd.host.model import (FirstOrderKernelDerivativeSeparateIndependent, Matern52, SquaredExponential,  # noqa: F401
                                  VanillaConditioningSeparateIndependent, VanillaConditioningSharedIndependent)
