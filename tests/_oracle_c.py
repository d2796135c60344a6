"""ctypes binding of the plain-C oracle (oracle/libplume_oracle.so) — checker only (tests, smoke, bench cpu_baseline)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "oracle" / "libplume_oracle.so"
_lib = None
u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        src = ROOT / "oracle" / "plume_oracle.c"
        if not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle")])
        _lib = C.CDLL(str(_SO))
        _lib.oracle_verify_batch.restype = C.c_int
        _lib.oracle_sign_batch.restype = C.c_int
        _lib.oracle_hash_to_curve_batch.restype = C.c_int
        _lib.oracle_point_mul.restype = C.c_int
        _lib.oracle_sec1_compress.restype = C.c_size_t
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


def pack_msgs(msgs):
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(m) for m in msgs])
    buf = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
    return buf, off


def verify_batch(version, msgs_buf, msg_off, pk, nul, c, s, r_point=None, hr=None, nthreads=1):
    n = len(msg_off) - 1
    ok = np.zeros(n, dtype=np.uint8)
    rc = lib().oracle_verify_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                   _p(pk), _p(nul), _p(c), _p(s), _p(r_point), _p(hr), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return ok


def verify_non_zk_batch(version, msgs_buf, msg_off, pk, nul, s, r_point, hr, digest_private, nthreads=1):
    """rust-arkworks/src/tests.rs:28-78; 1 Ok(true), 0 Ok(false), 2 Err(HashToCurveError)"""
    n = len(msg_off) - 1
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_non_zk_batch.restype = C.c_int
    rc = lib().oracle_verify_non_zk_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          _p(pk), _p(nul), _p(s), _p(r_point), _p(hr), _p(digest_private), _p(ok), C.c_int(nthreads))
    assert rc == 0, rc
    return ok


def sign_batch(version, msgs_buf, msg_off, sk, r, pk_in=None, nthreads=1):
    n = len(msg_off) - 1
    o = {k: np.zeros((n, w), dtype=np.uint8) for k, w in
         [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64), ("h", 64)]}
    status = np.zeros(n, dtype=np.uint8)
    rc = lib().oracle_sign_batch(C.c_int(version), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 _p(sk), _p(r), _p(pk_in), _p(o["pk"]), _p(o["nullifier"]), _p(o["c"]), _p(o["s"]),
                                 _p(o["r_point"]), _p(o["hashed_to_curve_r"]), _p(o["h"]), _p(status), C.c_int(nthreads))
    assert rc == 0, rc
    o["status"] = status
    return o


def hash_to_curve_batch(msgs_buf, msg_off, pk, nthreads=1):
    n = len(msg_off) - 1
    h = np.zeros((n, 64), dtype=np.uint8)
    lib().oracle_hash_to_curve_batch(C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)), _p(pk), _p(h), C.c_int(nthreads))
    return h


def h2c_raw(data: bytes):
    out = (C.c_uint8 * 256)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    lib().oracle_h2c_raw(buf, C.c_size_t(len(data)), out)
    b = bytes(out)
    return dict(u0=b[:32], u1=b[32:64], q0=b[64:128], q1=b[128:192], p=b[192:256])


def point_mul(k: bytes, p: bytes):
    out = (C.c_uint8 * 64)()
    ok = lib().oracle_point_mul((C.c_uint8 * 32).from_buffer_copy(k), (C.c_uint8 * 64).from_buffer_copy(p), out)
    return bytes(out) if ok else None


def sec1_compress(p: bytes):
    out = (C.c_uint8 * 33)()
    n = lib().oracle_sec1_compress((C.c_uint8 * 64).from_buffer_copy(p), out)
    return bytes(out)[:n]


def sha256(data: bytes):
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    lib().oracle_sha256(buf, C.c_size_t(len(data)), out)
    return bytes(out)


def arr(items, key, width):
    return np.frombuffer(b"".join(bytes.fromhex(it[key]) for it in items), dtype=np.uint8).reshape(len(items), width).copy()


def aggregate_check(version, mode, msgs_buf, msg_off, pk, nul, c, s, r_point, hr, seed: bytes, index_base=0, nthreads=1):
    """the DEFINITION of plume_aggregate_check (include/plume_hip.h), item by item; returns (record 72 B, hash_ok)"""
    n = len(msg_off) - 1
    hash_ok = np.full(n, 0xEE, dtype=np.uint8)
    rec = np.zeros(72, dtype=np.uint8)
    sd = np.frombuffer(seed, dtype=np.uint8).copy()
    assert len(sd) == 32
    lib().oracle_aggregate_check.restype = C.c_int
    rc = lib().oracle_aggregate_check(C.c_int(version), C.c_int(mode), C.c_size_t(n), _p(msgs_buf), msg_off.ctypes.data_as(C.POINTER(C.c_uint64)), _p(pk), _p(nul), _p(c), _p(s),
                                      _p(r_point), _p(hr), _p(sd), C.c_uint64(index_base), _p(hash_ok), _p(rec), C.c_int(nthreads))
    assert rc == 0, rc
    return rec, hash_ok


def parse_aggregate_record(rec):
    rec = np.asarray(rec, dtype=np.uint8)
    return dict(all_ok=int(rec[0]), identity=int(rec[1]), n_bad=int.from_bytes(rec[4:8].tobytes(), "little"), point=rec[8:72].tobytes())
