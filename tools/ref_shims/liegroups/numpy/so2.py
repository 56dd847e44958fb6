from .._core import SO2Matrix  # noqa: F401
