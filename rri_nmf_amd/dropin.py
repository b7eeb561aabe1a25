"""`import rri_nmf_amd.dropin` makes this package answer to the reference's import name:

    import rri_nmf_amd.dropin                      # once, before the imports below
    from rri_nmf.nmf import nmf                    # unchanged user code (reference: src/rri_nmf/nmf.py)
    from rri_nmf.sklearn_interface import NMF_TM_Estimator, NMF_RS_Estimator

It registers `rri_nmf` and its five modules in sys.modules as aliases of the modules here; nothing is copied.  It
refuses to shadow a real `rri_nmf` that is already imported (the two would then be mixed up silently).
"""
import importlib
import sys

_MODULES = ('nmf', 'sklearn_interface', 'initialization', 'matrixops', 'optimization')


def install():
    mine = importlib.import_module('rri_nmf_amd')
    other = sys.modules.get('rri_nmf')
    if other is not None and other is not mine:
        raise ImportError('a different `rri_nmf` is already imported (%r): not aliasing over it'
                          % getattr(other, '__file__', other))
    sys.modules['rri_nmf'] = mine
    for name in _MODULES:
        sys.modules['rri_nmf.' + name] = importlib.import_module('rri_nmf_amd.' + name)
    return mine


install()
