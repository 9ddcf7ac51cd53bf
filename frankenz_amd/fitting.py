"""Import path used by the reference's demos (frankenz/fitting.py:25-30):
``from frankenz.fitting import BruteForce, NearestNeighbors``."""
from .bruteforce import BruteForce
from .knn import NearestNeighbors

__all__ = ["BruteForce", "NearestNeighbors"]
