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


def verify_non_zk_batch(version, msgs_buf, msg_off, pk, nul, s, r_point, hr, digest_private, nthreads=1):
    """rust-arkworks/src/tests.rs:28-78; 1 Ok(true), 0 Ok(false), 2 Err(HashToCurveError)"""
    n = len(msg_off) - 1
    ok = np.zeros(n, dtype=np.uint8)
    lib().fast_verify_non_zk_batch.restype = C.c_int
    rc = lib().fast_verify_non_zk_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        _p(pk), _p(nul), _p(s), _p(r_point), _p(hr), _p(digest_private), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return ok


def sign_batch(version, msgs_buf, msg_off, sk, r, pk_in=None, nthreads=1):
    """same outputs as tests/_oracle_c.sign_batch (without h)"""
    n = len(msg_off) - 1
    o = {k: np.zeros((n, w), dtype=np.uint8) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    status = np.zeros(n, dtype=np.uint8)
    lib().fast_sign_batch.restype = C.c_int
    rc = lib().fast_sign_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)), _p(sk), _p(r), _p(pk_in),
                               _p(o["pk"]), _p(o["nullifier"]), _p(o["c"]), _p(o["s"]), _p(o["r_point"]), _p(o["hashed_to_curve_r"]), _p(status), C.c_int(nthreads))
    assert rc == 0, rc
    o["status"] = status
    return o


def sec1_decompress_batch(rec33, nthreads=1):
    """33-byte SEC1 records -> (64-byte affine records, ok flags): the reference's deserialization (00 = identity; bad tag / x >= p / no curve point = ok 0)"""
    rec33 = np.ascontiguousarray(rec33, dtype=np.uint8).reshape(-1, 33)
    n = len(rec33)
    out, ok = np.zeros((n, 64), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    lib().fast_sec1_decompress_batch.restype = C.c_int
    rc = lib().fast_sec1_decompress_batch(C.c_size_t(n), _p(rec33), _p(out), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return out, ok


def ff_op(op, a: int, b: int = 0) -> int:
    out = (C.c_uint8 * 32)()
    lib().fast_ff_op(C.c_int(op), (C.c_uint8 * 32).from_buffer_copy(a.to_bytes(32, "big")), (C.c_uint8 * 32).from_buffer_copy(b.to_bytes(32, "big")), out)
    return int.from_bytes(bytes(out), "big")


def glv(k: int):
    out = (C.c_uint8 * 50)()
    lib().fast_glv((C.c_uint8 * 32).from_buffer_copy(k.to_bytes(32, "big")), out)
    b = bytes(out)
    return (int.from_bytes(b[:24], "little"), b[24], int.from_bytes(b[25:49], "little"), b[49])
