from .solver import Solver  # noqa: F401


class ConjugateGradient(Solver):
    """Not on the captured path (reference default solver is TrustRegions)."""

    def __init__(self, *args, **kwargs):
        for k in ("beta_type", "orth_value", "linesearch"):
            kwargs.pop(k, None)
        super().__init__(*args, **kwargs)

    def solve(self, *a, **k):
        raise NotImplementedError("ConjugateGradient is outside the captured hot path")
