"""Stand-in for pymanopt 0.2.5 solvers/linesearch.py: the adaptive backtracking line search that
ConjugateGradient uses by default (restated from the package's published algorithm; test
infrastructure only, see ../../README.md)."""


class LineSearchAdaptive(object):
    def __init__(self, contraction_factor=.5, suff_decr=.5, maxiter=10, initial_stepsize=1):
        self._contraction_factor = contraction_factor
        self._suff_decr = suff_decr
        self._maxiter = maxiter
        self._initial_stepsize = initial_stepsize
        self._oldalpha = None

    def search(self, objective, man, x, d, f0, df0):
        norm_d = man.norm(x, d)
        if self._oldalpha is not None:
            alpha = self._oldalpha
        else:
            alpha = self._initial_stepsize / norm_d
        alpha = float(alpha)
        newx = man.retr(x, alpha * d)
        newf = objective(newx)
        cost_evaluations = 1
        while (newf > f0 + self._suff_decr * alpha * df0 and cost_evaluations <= self._maxiter):
            alpha *= self._contraction_factor          # reduce the step size,
            newx = man.retr(x, alpha * d)              # look closer down the line
            newf = objective(newx)
            cost_evaluations += 1
        if newf > f0:
            alpha = 0
            newx = x
        stepsize = alpha * norm_d
        # suggestion for the next initial trial (about alpha, not the step size): keep pace after
        # exactly one backtrack, otherwise speed up
        if cost_evaluations == 2:
            self._oldalpha = alpha
        else:
            self._oldalpha = 2 * alpha
        self.last_cost_evaluations = cost_evaluations      # (recording hook of the capture script)
        return stepsize, newx
