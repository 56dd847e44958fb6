"""Stand-in for pymanopt 0.2.5 solvers/solver.py: stopping criteria + optlog bookkeeping."""
import time


class Solver(object):
    def __init__(self, maxtime=1000, maxiter=1000, mingradnorm=1e-6, minstepsize=1e-10,
                 maxcostevals=5000, logverbosity=0):
        self._maxtime = maxtime
        self._maxiter = maxiter
        self._mingradnorm = mingradnorm
        self._minstepsize = minstepsize
        self._maxcostevals = maxcostevals
        self._logverbosity = logverbosity
        self._optlog = None

    def __str__(self):
        return type(self).__name__

    def _check_stopping_criterion(self, time0, iter=-1, gradnorm=float("inf"),
                                  stepsize=float("inf"), costevals=-1):
        reason = None
        if time.time() >= time0 + self._maxtime:
            reason = "Terminated - max time reached after %d iterations." % iter
        elif iter >= self._maxiter:
            reason = ("Terminated - max iterations reached after %.2f seconds."
                      % (time.time() - time0))
        elif gradnorm < self._mingradnorm:
            reason = ("Terminated - min grad norm reached after %d iterations, %.2f seconds."
                      % (iter, time.time() - time0))
        elif stepsize < self._minstepsize:
            reason = ("Terminated - min stepsize reached after %d iterations, %.2f seconds."
                      % (iter, time.time() - time0))
        elif costevals >= self._maxcostevals:
            reason = ("Terminated - max cost evals reached after %.2f seconds."
                      % (time.time() - time0))
        return reason

    def _start_optlog(self, solverparams=None, extraiterfields=None):
        if self._logverbosity <= 0:
            self._optlog = None
        else:
            self._optlog = {
                "solver": str(self),
                "stoppingcriteria": {
                    "maxtime": self._maxtime, "maxiter": self._maxiter,
                    "mingradnorm": self._mingradnorm, "minstepsize": self._minstepsize,
                    "maxcostevals": self._maxcostevals},
                "solverparams": solverparams,
            }
        if self._logverbosity >= 2:
            self._optlog["iterations"] = {"iteration": [], "time": [], "x": [], "f(x)": []}
            for field in (extraiterfields or []):
                self._optlog["iterations"][field] = []

    def _append_optlog(self, iteration, x, fx, **kwargs):
        it = self._optlog["iterations"]
        it["iteration"].append(iteration)
        it["time"].append(time.time())
        it["x"].append(x)
        it["f(x)"].append(fx)
        for key in kwargs:
            it[key].append(kwargs[key])

    def _stop_optlog(self, x, objective, stop_reason, time0, stepsize=float("inf"),
                     gradnorm=float("inf"), iter=-1, costevals=-1):
        self._optlog["stoppingreason"] = stop_reason
        self._optlog["final_values"] = {"x": x, "f(x)": objective, "time": time.time() - time0}
        if stepsize != float("inf"):
            self._optlog["final_values"]["stepsize"] = stepsize
        if gradnorm != float("inf"):
            self._optlog["final_values"]["gradnorm"] = gradnorm
        if iter != -1:
            self._optlog["final_values"]["iterations"] = iter
        if costevals != -1:
            self._optlog["final_values"]["costevals"] = costevals
