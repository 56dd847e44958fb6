"""Stand-in for pymanopt 0.2.5 solvers/conjugate_gradient.py (restated from the package's
published algorithm -- a port of Manopt's conjugategradient.m; test infrastructure only, see
../../README.md).  The reference selects it with params["solver"] = "ConjugateGradient"
(graphik/solvers/riemannian_solver.py:51-59: beta_type = BetaTypes[3] = HagerZhang,
orth_value = 10e10, maxiter = 10e4, mingradnorm = 1e-9, minstepsize = 1e-10)."""
import time
from copy import deepcopy

import numpy as np

from .. import tools
from .linesearch import LineSearchAdaptive
from .solver import Solver

BetaTypes = tools.make_enum("BetaTypes", "FletcherReeves PolakRibiere HestenesStiefel HagerZhang".split())


class ConjugateGradient(Solver):
    def __init__(self, beta_type=BetaTypes.HestenesStiefel, orth_value=np.inf, linesearch=None,
                 *args, **kwargs):
        super(ConjugateGradient, self).__init__(*args, **kwargs)
        self._beta_type = beta_type
        self._orth_value = orth_value
        self._linesearch = LineSearchAdaptive() if linesearch is None else linesearch
        self.linesearch = None

    def solve(self, problem, x=None, reuselinesearch=False):
        man = problem.manifold
        objective = problem.cost
        gradient = problem.grad
        if not reuselinesearch or self.linesearch is None:
            self.linesearch = deepcopy(self._linesearch)
        linesearch = self.linesearch
        if x is None:
            x = man.rand()
        iter = 0
        stepsize = np.nan
        time0 = time.time()
        cost = objective(x)
        grad = gradient(x)
        gradnorm = man.norm(x, grad)
        Pgrad = problem.precon(x, grad)
        gradPgrad = man.inner(x, grad, Pgrad)
        desc_dir = -Pgrad          # initial descent direction: the negative gradient
        self._start_optlog(extraiterfields=["gradnorm"],
                           solverparams={"beta_type": self._beta_type, "orth_value": self._orth_value,
                                         "linesearcher": linesearch})
        while True:
            if self._logverbosity >= 2:
                self._append_optlog(iter, x, cost, gradnorm=gradnorm)
            stop_reason = self._check_stopping_criterion(time0, gradnorm=gradnorm, iter=iter + 1,
                                                         stepsize=stepsize)
            if stop_reason:
                break
            # directional derivative along the search direction; restart on an ascent direction
            df0 = man.inner(x, grad, desc_dir)
            if df0 >= 0:
                desc_dir = -Pgrad
                df0 = -gradPgrad
            stepsize, newx = linesearch.search(objective, man, x, desc_dir, cost, df0)
            newcost = objective(newx)
            newgrad = gradient(newx)
            newgradnorm = man.norm(newx, newgrad)
            Pnewgrad = problem.precon(newx, newgrad)
            newgradPnewgrad = man.inner(newx, newgrad, Pnewgrad)
            # CG scheme for the next search direction
            oldgrad = man.transp(x, newx, grad)
            orth_grads = man.inner(newx, oldgrad, Pnewgrad) / newgradPnewgrad
            if abs(orth_grads) >= self._orth_value:     # Powell's restart strategy
                beta = 0
                desc_dir = -Pnewgrad
            else:
                desc_dir = man.transp(x, newx, desc_dir)
                if self._beta_type == BetaTypes.FletcherReeves:
                    beta = newgradPnewgrad / gradPgrad
                elif self._beta_type == BetaTypes.PolakRibiere:
                    diff = newgrad - oldgrad
                    ip_diff = man.inner(newx, Pnewgrad, diff)
                    beta = max(0, ip_diff / gradPgrad)
                elif self._beta_type == BetaTypes.HestenesStiefel:
                    diff = newgrad - oldgrad
                    ip_diff = man.inner(newx, Pnewgrad, diff)
                    try:
                        beta = max(0, ip_diff / man.inner(newx, diff, desc_dir))
                    except ZeroDivisionError:
                        beta = 1
                elif self._beta_type == BetaTypes.HagerZhang:
                    diff = newgrad - oldgrad
                    Poldgrad = man.transp(x, newx, Pgrad)
                    Pdiff = Pnewgrad - Poldgrad
                    deno = man.inner(newx, diff, desc_dir)
                    numo = man.inner(newx, diff, Pnewgrad)
                    numo -= (2 * man.inner(newx, diff, Pdiff) *
                             man.inner(newx, desc_dir, newgrad) / deno)
                    beta = numo / deno
                    # robustness (Hager-Zhang): lower bound on beta
                    desc_dir_norm = man.norm(newx, desc_dir)
                    eta_HZ = -1 / (desc_dir_norm * min(0.01, gradnorm))
                    beta = max(beta, eta_HZ)
                else:
                    raise ValueError("Unknown beta_type %s" % self._beta_type)
                desc_dir = -Pnewgrad + beta * desc_dir
            x = newx
            cost = newcost
            grad = newgrad
            Pgrad = Pnewgrad
            gradnorm = newgradnorm
            gradPgrad = newgradPnewgrad
            iter += 1
        if self._logverbosity <= 0:
            return x
        self._stop_optlog(x, cost, stop_reason, time0, stepsize=stepsize, gradnorm=gradnorm, iter=iter)
        return x, self._optlog
