"""Loads the UNMODIFIED reference (/root/reference/src/rri_nmf/*.py, Python 2 era
code) under Python 3 / numpy 2 so that golden vectors can be captured from it.

BUILD-CONTAINER TOOLING ONLY: /root/reference does not exist on the GPU box and
nothing in tests/, bench.py or the package imports this file at run time; only
oracle/make_golden.py does.  No reference source is copied: the modules are
imported from where they lie, with shims for names that newer numpy / scipy /
the stdlib removed (recipe: SURVEY.md Appendix A).
"""
import logging
import sys
import time
import types

import numpy as np
import scipy
import scipy.sparse  # noqa: F401

REF_SRC = '/root/reference/src/rri_nmf'
REF_DATA = '/root/reference/tests/data'


def load(quiet_objective=True):
    """Returns the reference modules as a namespace.  quiet_objective sets the
    reference logger to WARNING, which switches off its per-update objective
    evaluations (nmf.py:366,568-572,600-606); W,T per sweep are unaffected."""
    sys.dont_write_bytecode = True
    if 'numexpr' not in sys.modules:  # not installed; nmf.py:15-16,358,696,741 use 3 names
        ne = types.ModuleType('numexpr')

        def _evaluate(expr, local_dict=None, global_dict=None):
            f = sys._getframe(1)
            return eval(expr, dict(f.f_globals), dict(f.f_locals))
        ne.evaluate = _evaluate
        ne.set_num_threads = lambda n: None
        ne.detect_number_of_cores = lambda: 1
        sys.modules['numexpr'] = ne
    if not hasattr(time, 'clock'):
        time.clock = time.process_time
    if not hasattr(np, 'alltrue'):
        np.alltrue = np.all
    if not hasattr(np, 'int'):
        np.int = int
    for nm in ('sum', 'maximum', 'minimum', 'argmax'):
        if not hasattr(scipy, nm):
            setattr(scipy, nm, getattr(np, nm))
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import matrixops, optimization, initialization, nmf, sklearn_interface  # noqa: E401
    for wrapped in (nmf._project_and_check_reset_t, nmf._check_reset_W):
        f = wrapped.__closure__[0].cell_contents
        f.func_name = f.__name__
    nmf.logger.setLevel(logging.WARNING if quiet_objective else logging.NOTSET)
    return types.SimpleNamespace(matrixops=matrixops, optimization=optimization,
                                 initialization=initialization, nmf=nmf,
                                 sklearn_interface=sklearn_interface)
