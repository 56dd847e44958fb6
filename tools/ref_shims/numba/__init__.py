"""Stand-in for numba: decorators are identities (reference loops run as plain Python)."""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco


jit = njit = _identity_decorator
