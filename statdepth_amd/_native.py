"""ctypes binding of libstatdepth_hip.so (the C ABI in include/statdepth_hip.h).

The library is the product's only compute path: if it is not built, or no gfx950
device is visible, calls raise -- there is no CPU fallback.  torch is used purely
as plumbing here (device memory + streams); the signatures are plain C.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstatdepth_hip.so")
# The cross-check build of the same sources (-DSD_CROSSCHECK: retired kernel generations + the environment switches that
# select them).  TESTS ONLY: nothing in this package opens it; tests/conftest.py swaps it in for the comparisons.
XCHECK_LIB_PATH = os.path.join(_HERE, "lib", "libstatdepth_hip_xcheck.so")

SD_OK = 0
SD_ERR_INVALID, SD_ERR_HIP, SD_ERR_NO_DEVICE, SD_ERR_UNSUPPORTED, SD_ERR_OVERFLOW, SD_ERR_WORKSPACE = 1, 2, 3, 4, 5, 6
SD_MBD_AUTO, SD_MBD_PAIRWISE, SD_MBD_RANK = 0, 1, 2
ALGOS = {"auto": SD_MBD_AUTO, "pairwise": SD_MBD_PAIRWISE, "rank": SD_MBD_RANK}

# every symbol include/statdepth_hip.h declares, with its C signature
_c = ctypes
_vp, _i64, _int, _sz, _dbl, _u64 = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t, _c.c_double, _c.c_uint64
SIGNATURES = {
    "sd_abi_version": (_int, []),
    "sd_is_crosscheck_build": (_int, []),
    "sd_last_error": (_c.c_char_p, []),
    "sd_device_count": (_int, []),
    "sd_device_info": (_int, [_int, _c.c_char_p, _int, _c.POINTER(_int), _c.POINTER(_sz)]),
    "sd_set_device": (_int, [_int]),
    "sd_malloc": (_int, [_c.POINTER(_vp), _sz]),
    "sd_free": (_int, [_vp]),
    "sd_memcpy_h2d": (_int, [_vp, _vp, _sz, _vp]),
    "sd_memcpy_d2h": (_int, [_vp, _vp, _sz, _vp]),
    "sd_memset": (_int, [_vp, _int, _sz, _vp]),
    "sd_stream_synchronize": (_int, [_vp]),
    "sd_mbd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _int, _int]),
    "sd_mbd_counts": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "sd_mbd_counts_range": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "sd_mbd_wide_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _int, _int]),
    "sd_mbd_counts_wide": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "sd_mbd_external_counts": (_int, [_vp, _i64, _i64, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "sd_mbd_external_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "sd_mbd_subset_counts": (_int, [_vp, _i64, _i64, _vp, _i64, _int, _vp, _int, _vp, _vp]),
    "sd_bd_strict_subset_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "sd_bd_strict_subset_counts": (_int, [_vp, _i64, _i64, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "sd_bd_strict_subset_supported": (_int, [_i64, _int]),
    "sd_bd_strict_external_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "sd_bd_strict_external_counts": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sd_above_below": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sd_bd_strict_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "sd_bd_strict_nanfree_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "sd_bd_strict_min_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _int]),
    "sd_bd_strict_counts": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sd_bd_strict_j_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _int]),
    "sd_bd_strict_j_counts": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "sd_l1_depth": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _vp]),
    "sd_l1_external_depth": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _vp]),
    "sd_l1_subset_depth": (_int, [_vp, _i64, _int, _vp, _i64, _int, _vp, _vp]),
    "sd_pointcloud_simplex_external_counts": (_int, [_vp, _i64, _int, _vp, _i64, _dbl, _vp, _vp]),
    "sd_pointcloud_simplex_subset_counts": (_int, [_vp, _i64, _int, _vp, _i64, _int, _dbl, _vp, _vp]),
    "sd_pointcloud_simplex_counts": (_int, [_vp, _i64, _int, _vp, _i64, _dbl, _vp, _vp]),
    "sd_multi_simplex_counts": (_int, [_vp, _i64, _i64, _int, _vp, _i64, _int, _dbl, _vp, _vp]),
    "sd_simplex_sampled_workspace_bytes": (_sz, [_i64, _i64, _int, _i64]),
    "sd_pointcloud_simplex_sampled": (_int, [_vp, _i64, _int, _vp, _i64, _dbl, _i64, _u64, _vp, _vp, _sz, _vp]),
    "sd_multi_band_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "sd_multi_band_counts": (_int, [_vp, _i64, _i64, _int, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sd_multi_simplex_sampled": (_int, [_vp, _i64, _i64, _int, _vp, _i64, _int, _dbl, _i64, _u64, _vp, _vp, _sz, _vp]),
}


class StatdepthHipError(RuntimeError):
    """A call into libstatdepth_hip.so failed (code + the library's message)."""

    def __init__(self, code, msg):
        super().__init__(f"libstatdepth_hip error {code}: {msg}")
        self.code = code


_LIB = None


def open_library(path):
    """dlopen one build of the library and bind every symbol of include/statdepth_hip.h (no GPU needed)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C statdepth_amd/csrc`. statdepth_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.sd_abi_version() != 1:
        raise RuntimeError(f"{os.path.basename(path)} ABI version mismatch")
    return lib


def load():
    """Load the product library (no GPU needed for this step)."""
    global _LIB
    if _LIB is None:
        lib = open_library(LIB_PATH)
        if lib.sd_is_crosscheck_build() != 0:
            raise RuntimeError("libstatdepth_hip.so was built with -DSD_CROSSCHECK: not the product library")
        _LIB = lib
    return _LIB


def check(code):
    if code != SD_OK:
        msg = load().sd_last_error()
        raise StatdepthHipError(code, msg.decode() if msg else "")


def require_device():
    """Fail loudly when there is nothing to run on."""
    lib = load()
    if lib.sd_device_count() <= 0:
        raise RuntimeError("statdepth_amd: no HIP device visible (MI355X / gfx950 required; there is no CPU fallback)")
    return lib
