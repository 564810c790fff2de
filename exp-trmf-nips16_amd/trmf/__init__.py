"""MI355X-native TRMF (temporal regularized matrix factorization) -- Python front end.

Importable exactly like the reference package (python/trmf/__init__.py:1-6):
``from trmf import Model, Metrics, train, rolling_validate, grid_search``.
"""
from .trmf import Model, Metrics
from .trmf import train, fit, rolling_validate, grid_search

__all__ = ['Model', 'Metrics', 'train', 'fit', 'rolling_validate', 'grid_search']
