"""rri_nmf_amd: MI355X-native rank-one residue iteration (RRI) NMF.

Mirrors the package surface of maksimt/rri_nmf (`nmf`, `matrixops`, `optimization`,
`initialization`, `sklearn_interface`); the sweep loop runs in librri_hip.so (include/rri_hip.h).
"""
from . import matrixops, optimization  # noqa: F401

__all__ = ['nmf', 'matrixops', 'optimization', 'initialization', 'sklearn_interface', 'engine']


def __getattr__(name):  # lazy: importing the package must not need sklearn / the GPU library
    if name in ('nmf', 'initialization', 'sklearn_interface', 'engine', 'distributed', 'synthetic'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
