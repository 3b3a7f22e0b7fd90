from .depth import FunctionalDepth, PointcloudDepth   # noqa: F401  (reference: statdepth/depth/__init__.py:1)
from .calculations._helper import DepthDegeneracy     # noqa: F401
