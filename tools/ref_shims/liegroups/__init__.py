from .numpy import SO2, SO3, SE2, SE3  # noqa: F401
