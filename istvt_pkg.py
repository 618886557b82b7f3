"""Loader for the package directory ``2023-tifs-istvt_amd/`` (not an importable name).

    import istvt_pkg; istvt_amd = istvt_pkg.load()
    from istvt_amd.network.vivit.vivit import XceptionVidTr
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, '2023-tifs-istvt_amd')
ALIAS = 'istvt_amd'


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(PKG_DIR, '__init__.py'),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
