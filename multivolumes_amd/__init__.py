"""multivolumes_amd -- MI355X-native cube-map-space volume ray-march path (StarsX/MultiVolumes hot path).

Only what the path needs: csrc/ (gfx950 HIP kernels + the C ABI of include/multivolumes.h + host helpers) and
raycaster.py, the host-side mirror of the reference's MultiRayCaster interface over that ABI.
"""
from . import _lib
from ._lib import MVError, build, load
from .raycaster import MultiRayCaster

__all__ = ["MultiRayCaster", "MVError", "build", "load", "_lib"]
