from .robot_revolute import RobotRevolute  # noqa: F401
from .robot_planar import RobotPlanar  # noqa: F401
