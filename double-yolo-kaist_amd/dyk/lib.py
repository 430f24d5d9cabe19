"""ctypes binding of libdyk_hip.so (C ABI declared in include/dyk_hip.h).

The product path has no CPU fallback: if the shared library cannot be loaded the import
of any operator that needs it raises ``DykLibraryError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libdyk_hip.so")

DYK_F32, DYK_BF16 = 0, 1
ACT_CODES = {"linear": 0, "leaky": 1, "mish": 2, "relu": 3, "relu6": 4, "hard-sigmoid": 5, "hard-swish": 6}
EPI_AFFINE, EPI_RESIDUAL, EPI_STATS, EPI_ACCUM, EPI_OUT_F32 = 1, 2, 4, 8, 16
MAX_TAPS = 25


class DykLibraryError(RuntimeError):
    pass


class DykError(RuntimeError):
    pass


class DykConvDesc(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("res", ctypes.c_void_p),
        ("stats", ctypes.c_void_p),
        ("dtype", ctypes.c_int32),
        ("ldx", ctypes.c_int32), ("ldy", ctypes.c_int32), ("ldr", ctypes.c_int32),
        ("B", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32),
        ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("Hg", ctypes.c_int32), ("Wg", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
        ("isy", ctypes.c_int32), ("isx", ctypes.c_int32), ("osy", ctypes.c_int32), ("osx", ctypes.c_int32),
        ("ooy", ctypes.c_int32), ("oox", ctypes.c_int32),
        ("ntaps", ctypes.c_int32),
        ("tdy", ctypes.c_int8 * MAX_TAPS), ("tdx", ctypes.c_int8 * MAX_TAPS), ("twt", ctypes.c_int8 * MAX_TAPS),
        ("_pad", ctypes.c_int8),
        ("act", ctypes.c_int32), ("flags", ctypes.c_int32),
    ]


class DykWgradDesc(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p),
        ("dtype", ctypes.c_int32), ("ldx", ctypes.c_int32), ("lddy", ctypes.c_int32),
        ("B", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("isy", ctypes.c_int32), ("isx", ctypes.c_int32),
        ("ntaps", ctypes.c_int32),
        ("tdy", ctypes.c_int8 * MAX_TAPS), ("tdx", ctypes.c_int8 * MAX_TAPS), ("twt", ctypes.c_int8 * MAX_TAPS),
        ("_pad", ctypes.c_int8),
        ("splits", ctypes.c_int32),
    ]


_lib = None

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against
# the declarations in include/dyk_hip.h.
_i32, _f32, _vp = ctypes.c_int32, ctypes.c_float, ctypes.c_void_p
SIGNATURES = {
    "dyk_abi_version": (_i32, []),
    "dyk_error_string": (ctypes.c_char_p, [_i32]),
    "dyk_conv_igemm": (_i32, [ctypes.POINTER(DykConvDesc), _vp]),
    "dyk_conv_wgrad": (_i32, [ctypes.POINTER(DykWgradDesc), _vp]),
    "dyk_pack_conv_weight": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dyk_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dyk_nhwc_to_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
}


def load(path=None):
    """Load (once) and return the ctypes handle of libdyk_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    p = path or os.environ.get("DYK_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise DykLibraryError(
            "libdyk_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C double-yolo-kaist_amd/csrc`). There is no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # pragma: no cover
        raise DykLibraryError("cannot load %s: %s" % (p, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().dyk_error_string(code).decode()
        raise DykError("%s failed: %s (%d)" % (what or "dyk call", msg, code))
