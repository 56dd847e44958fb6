from .._core import SE3Matrix  # noqa: F401
