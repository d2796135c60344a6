"""ctypes binding of the optimised CPU leg (oracle/libplume_cpu_fast.so) — measurement / test infrastructure only (bench.py cpu_baseline, tests/test_cpu_fast.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "oracle" / "libplume_cpu_fast.so"
_lib = None
u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        srcs = [ROOT / "oracle" / "plume_cpu_fast.c", ROOT / "oracle" / "plume_oracle.c"]
        if not _SO.exists() or _SO.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "libplume_cpu_fast.so"])
        _lib = C.CDLL(str(_SO))
        _lib.fast_verify_batch.restype = C.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


def verify_batch(version, msgs_buf, msg_off, pk, nul, c, s, r_point=None, hr=None, nthreads=1):
    n = len(msg_off) - 1
    ok = np.zeros(n, dtype=np.uint8)
    rc = lib().fast_verify_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 _p(pk), _p(nul), _p(c), _p(s), _p(r_point), _p(hr), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return ok


def ff_op(op, a: int, b: int = 0) -> int:
    out = (C.c_uint8 * 32)()
    lib().fast_ff_op(C.c_int(op), (C.c_uint8 * 32).from_buffer_copy(a.to_bytes(32, "big")), (C.c_uint8 * 32).from_buffer_copy(b.to_bytes(32, "big")), out)
    return int.from_bytes(bytes(out), "big")


def glv(k: int):
    out = (C.c_uint8 * 50)()
    lib().fast_glv((C.c_uint8 * 32).from_buffer_copy(k.to_bytes(32, "big")), out)
    b = bytes(out)
    return (int.from_bytes(b[:24], "little"), b[24], int.from_bytes(b[25:49], "little"), b[49])
