"""Mirror of the reference's ``network`` package for the ISTVT hot path only
(network/xception.py, network/vivit/module.py, network/vivit/vivit.py, network/models.py).

Importable either as ``istvt_amd.network`` (via ``istvt_pkg.load()``) or, for a drop-in
``from network.models import model_selection`` as in the reference's train_CNN.py:15, by putting
the package directory ``2023-tifs-istvt_amd/`` on ``sys.path``.
"""
import importlib.util
import os
import sys

if 'istvt_amd' not in sys.modules:       # imported as top-level `network`: load the parent package
    _pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _spec = importlib.util.spec_from_file_location('istvt_amd', os.path.join(_pkg, '__init__.py'),
                                                   submodule_search_locations=[_pkg])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules['istvt_amd'] = _mod
    _spec.loader.exec_module(_mod)
