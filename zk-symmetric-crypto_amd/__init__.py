"""zk-symmetric-crypto_amd -- MI355X-native Groth16 prover for the gnark ChaCha20 / AES-CTR circuits.

Python host mirror of the reference's gnark operator boundary
(js/src/gnark/utils.ts:47-136 koffi bindings; gnark/libraries/prover/libprove.go:18-48):
the product is the C-ABI shared library `libzkprove.so` (HIP, gfx950); this module only loads it
with ctypes and exposes the same calls.  There is NO CPU fallback: if the HIP library is missing
the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkprove.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libzkprove.so (HIP extension) not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback in the product path)" % LIB_PATH)

lib = C.CDLL(LIB_PATH)


class GoSlice(C.Structure):  # js/src/gnark/utils.ts:47-51
    _fields_ = [("data", C.c_void_p), ("len", C.c_longlong), ("cap", C.c_longlong)]


class ProveReturn(C.Structure):  # js/src/gnark/utils.ts:53-56
    _fields_ = [("r0", C.c_void_p), ("r1", C.c_longlong)]


def _sig(name, restype, argtypes):
    fn = getattr(lib, name, None)
    if fn is not None:
        fn.restype = restype
        fn.argtypes = argtypes
    return fn


_vp, _i64, _i32 = C.c_void_p, C.c_long, C.c_int
_sig("zkp_last_error", C.c_char_p, [])
_sig("zkp_device_count", _i32, [])
_sig("zkp_ntt", _i32, [_i32, _i64, _vp, _i32])
_sig("zkp_compute_h", _i32, [_i32, _i64, _i64, _vp, _vp, _vp, _vp])
_sig("zkp_msm_g1", _i32, [_i64, _i64, _vp, _vp, _vp, _i32])
_sig("zkp_msm_g2", _i32, [_i64, _i64, _vp, _vp, _vp, _i32])
_sig("zkp_microbench", C.c_double, [_i32, _i64, _i32])


def last_error() -> str:
    return lib.zkp_last_error().decode()


def device_count() -> int:
    return lib.zkp_device_count()
