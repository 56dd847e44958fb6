from .._core import SO3Matrix  # noqa: F401
