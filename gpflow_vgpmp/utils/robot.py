from vgpmp_amd.host.environment import Robot  # noqa: F401
from vgpmp_amd.robots import base_pose_matrix, quat_to_rotmat  # noqa: F401
