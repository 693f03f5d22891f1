"""pl-inertial-slam_amd — MI355X-native local bundle adjustment for point-line visual-inertial SLAM.

Python host-side access to the C ABI of include/plba.h implemented by hand-written HIP kernels
(csrc/, built into libplba_hip.so for gfx950).  There is NO CPU fallback: if the HIP library is
missing or cannot be loaded this package raises at first use.

The directory name contains a hyphen (it mirrors the reference repository's name); import it with
``__graft_entry__.load_package()`` which registers it as ``pl_inertial_slam_amd``.
"""
import os

from . import abi, window, protocol, distributed  # noqa: F401
from .abi import (EDGE_POINT, EDGE_LINE, EDGE_IMU_PVR, EDGE_IMU_BIAS, EDGE_PRIOR, PlbaError,  # noqa: F401
                  Problem)

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libplba_hip.so")
_lib = None


def hip_lib():
    """Load (once) the HIP implementation of plba.h.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise PlbaError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the product path)" % HIP_LIB_PATH)
        _lib = abi.Lib(HIP_LIB_PATH, "plba_")
    return _lib


def new_problem(**opts):
    """A fresh device-resident BA problem on the current HIP device."""
    return Problem(hip_lib(), **opts)
