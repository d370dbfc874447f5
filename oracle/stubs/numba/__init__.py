"""Minimal stand-in for `numba`, used ONLY by oracle/make_goldens.py in the build container.

TEST INFRASTRUCTURE - never imported by the product path.

The reference decorates its RHS loop with ``@njit`` (marlpde/LHeureux_model.py:148,362).
Compiled numba code ignores ``np.seterr`` (LHeureux_model.py:6 sets "raise"), i.e. it
yields NaN/Inf silently.  The stand-in therefore runs the undecorated Python function
inside ``np.errstate(all="ignore")`` so the numerical behaviour (not the speed) matches.
"""
import functools

import numpy as np


def _wrap(func):
    @functools.wraps(func)
    def runner(*args, **kwargs):
        with np.errstate(all="ignore"):
            return func(*args, **kwargs)

    return runner


def njit(*args, **kwargs):
    # bare decorator:  @njit
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _wrap(args[0])

    # parametrised decorator:  @njit(cache=False)
    def deco(func):
        return _wrap(func)

    return deco


jit = njit
