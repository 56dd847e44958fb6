from .so2 import SO2Matrix as SO2  # noqa: F401
from .so3 import SO3Matrix as SO3  # noqa: F401
from .se2 import SE2Matrix as SE2  # noqa: F401
from .se3 import SE3Matrix as SE3  # noqa: F401
