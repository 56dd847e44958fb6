from .solver import Solver  # noqa: F401
from .conjugate_gradient import ConjugateGradient  # noqa: F401
from .linesearch import LineSearchAdaptive  # noqa: F401
