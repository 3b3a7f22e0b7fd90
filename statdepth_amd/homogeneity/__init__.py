from .homogeneity import FunctionalHomogeneity, PointcloudHomogeneity   # noqa: F401  (reference: homogeneity/__init__.py:1-2)
from .homogeneity import P1_homogeneity, P2_homogeneity                 # noqa: F401
