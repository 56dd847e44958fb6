"""Stand-in for pymanopt 0.2.5 (only what graphik.solvers.riemannian_solver touches)."""
from . import tools  # noqa: F401


class Problem:
    """pymanopt 0.2.5 Problem with user-supplied callables (no autodiff backend)."""

    def __init__(self, manifold, cost, egrad=None, ehess=None, grad=None, hess=None,
                 arg=None, precon=None, verbosity=2):
        self.manifold = manifold
        self.cost = cost
        self.egrad = egrad
        self.ehess = ehess
        self._grad = grad
        self._hess = hess
        self._precon = precon
        self.verbosity = verbosity

    @property
    def precon(self):
        if self._precon is None:
            def precon(x, d):
                return d
            return precon
        return self._precon

    @property
    def grad(self):
        if self._grad is None:
            egrad = self.egrad

            def grad(x):
                return self.manifold.egrad2rgrad(x, egrad(x))

            self._grad = grad
        return self._grad

    @property
    def hess(self):
        if self._hess is None:
            ehess = self.ehess

            def hess(x, a):
                # pymanopt 0.2.5 evaluates egrad(x) for every Hessian-vector product
                return self.manifold.ehess2rhess(x, self.egrad(x), ehess(x, a), a)

            self._hess = hess
        return self._hess
