"""statdepth_amd -- MI355X-native band-depth engine behind statdepth's API.

Drop-in for the reference's `from statdepth import FunctionalDepth, PointcloudDepth`
(statdepth/__init__.py:1): same factories, same keyword arguments, same result
objects; the arithmetic runs in hand-written gfx950 HIP kernels
(statdepth_amd/csrc, C ABI in include/statdepth_hip.h).  No CPU fallback.
"""
from .depth import FunctionalDepth, PointcloudDepth, DepthDegeneracy   # noqa: F401

__all__ = ["FunctionalDepth", "PointcloudDepth", "DepthDegeneracy"]
__version__ = "0.1.0"
