from .graph_base import DistanceGraph, ProblemGraph  # noqa: F401
from .graph_revolute import ProblemGraphRevolute  # noqa: F401
from .graph_planar import ProblemGraphPlanar  # noqa: F401
