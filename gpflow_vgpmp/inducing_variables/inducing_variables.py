from vgpmp_amd.host.model import ConditionedVariableInducingPoints, SharedIndependentInducingVariables  # noqa: F401
