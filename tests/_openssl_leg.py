"""ctypes binding of the OpenSSL-backed CPU leg (oracle/libplume_openssl_leg.so) — measurement / test infrastructure only (bench.py cpu_baseline, tests/test_openssl_leg.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "oracle" / "libplume_openssl_leg.so"
_lib = None
u8p = C.POINTER(C.c_uint8)


def available() -> bool:
    try:
        lib()
        return True
    except Exception:
        return False


def lib():
    global _lib
    if _lib is None:
        srcs = [ROOT / "oracle" / "plume_openssl_leg.c", ROOT / "oracle" / "plume_oracle.c"]
        if not _SO.exists() or _SO.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "libplume_openssl_leg.so"])
        _lib = C.CDLL(str(_SO))
        _lib.ossl_verify_batch.restype = C.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


def verify_batch(version, msgs_buf, msg_off, pk, nul, c, s, r_point=None, hr=None, nthreads=1):
    n = len(msg_off) - 1
    ok = np.zeros(n, dtype=np.uint8)
    rc = lib().ossl_verify_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 _p(pk), _p(nul), _p(c), _p(s), _p(r_point), _p(hr), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return ok
