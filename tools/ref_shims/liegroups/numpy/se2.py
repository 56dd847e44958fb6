from .._core import SE2Matrix  # noqa: F401
