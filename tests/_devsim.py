"""ctypes binding of tests/devsim/libplume_devsim.so: the product's device headers compiled for the host (test-only)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_DIR = ROOT / "tests" / "devsim"
_SO = _DIR / "libplume_devsim.so"
_lib = None
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        import os
        if os.environ.get("PLUME_DEVSIM_SO"):           # another build of the same harness (tests/test_devsim.py: the 5-bit-window build)
            _lib = C.CDLL(os.environ["PLUME_DEVSIM_SO"])
            return _lib
        srcs = [_DIR / "devsim.cpp"] + list((ROOT / "zk-nullifier-sig_amd" / "csrc").glob("*.h"))
        if not _SO.exists() or _SO.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["make", "-s", "-C", str(_DIR)])
        _lib = C.CDLL(str(_SO))
    return _lib


def _p(a, t=u8p):
    return None if a is None else a.ctypes.data_as(t)


def to_limbs(vals):
    """list of ints (< 2^256) -> uint32 array [n, 8] little-endian limbs"""
    return np.array([[(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)] for v in vals], dtype=np.uint32)


def from_limbs(a):
    return [sum(int(a[r, i]) << (32 * i) for i in range(a.shape[1])) for r in range(a.shape[0])]


def fe_op(op, a, b=None):
    A = to_limbs(a)
    Bm = to_limbs(b if b is not None else [0] * len(a))
    out = np.zeros_like(A)
    lib().ds_fe_op(C.c_int(op), C.c_size_t(len(a)), _p(A, u32p), _p(Bm, u32p), _p(out, u32p))
    return from_limbs(out)


def fe_raw(op, a, b=None, c=None, e=None):
    """products on raw 9 x 29-bit limb vectors (lists of 9 ints per element, any magnitude the contract allows) -> raw result limbs"""
    n = len(a)
    arrs = [np.ascontiguousarray(np.array(v if v is not None else [[0] * 9] * n, dtype=np.uint32).reshape(n, 9)) for v in (a, b, c, e)]
    out = np.zeros((n, 9), dtype=np.uint32)
    lib().ds_fe_raw(C.c_int(op), C.c_size_t(n), *[_p(x, u32p) for x in arrs], _p(out, u32p))
    return [[int(w) for w in row] for row in out]


def group_raw(op, px, py, pz, qx, qy):
    """jac_dbl / jac_dbl_neg / jac_madd / jac_add on raw limb vectors (lists of 9 ints): the host build's limb-bound assertions abort the process on a violation"""
    n = len(px)
    arrs = [np.ascontiguousarray(np.array(v, dtype=np.uint32).reshape(n, 9)) for v in (px, py, pz, qx, qy)]
    out = np.zeros((n, 27), dtype=np.uint32)
    lib().ds_group_raw(C.c_int(op), C.c_size_t(n), *[_p(x, u32p) for x in arrs], _p(out, u32p))
    return [[int(w) for w in row] for row in out]


def sc_op(op, a, b=None):
    A = to_limbs(a)
    Bm = to_limbs(b if b is not None else [0] * len(a))
    out = np.zeros_like(A)
    lib().ds_sc_op(C.c_int(op), C.c_size_t(len(a)), _p(A, u32p), _p(Bm, u32p), _p(out, u32p))
    return from_limbs(out)


NPOS = 65          # PLUME_NPOS: positions of the Eisenstein digits of a pair of 128-bit halves


def eis_digit(code):
    """digit code of csrc/plume_ec.h -> the Eisenstein integer (a, b) = a + b w it stands for: 0 -> 0; 1 + 6 row + 2 j + neg -> (-1)^neg w^j (1 | theta = 1 - w | 2)"""
    if code == 0:
        return (0, 0)
    c = code - 1
    row, j, neg = c // 6, (c % 6) >> 1, c & 1
    a, b = [(1, 0), (1, -1), (2, 0)][row]
    for _ in range(j):
        a, b = -b, a - b                                   # times w:  w (a + b w) = -b + (a - b) w
    return (-a, -b) if neg else (a, b)


def glv(ks):
    """k -> (|k1|, sign, |k2|, sign, the 65 digit codes of k1 + k2 w)"""
    K = to_limbs(ks)
    out = np.zeros((len(ks), 10), dtype=np.uint32)
    dig = np.zeros((len(ks), NPOS), dtype=np.int8)
    lib().ds_glv(C.c_size_t(len(ks)), _p(K, u32p), _p(out, u32p), dig.ctypes.data_as(C.POINTER(C.c_int8)))
    res = []
    for r in range(len(ks)):
        m1 = sum(int(out[r, i]) << (32 * i) for i in range(4))
        m2 = sum(int(out[r, 5 + i]) << (32 * i) for i in range(4))
        res.append((m1, int(out[r, 4]), m2, int(out[r, 9]), dig[r].tolist()))
    return res


def wbits():
    """window width of the build under test (PLUME_WBITS)"""
    lib().ds_wbits.restype = C.c_uint32
    return int(lib().ds_wbits())


def sha256(data: bytes):
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    lib().ds_sha256(buf, C.c_uint32(len(data)), out)
    return bytes(out)


def _aligned(a):
    """device code assumes 4-byte aligned records (HBM allocations are); numpy allocations are 16+ aligned"""
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    assert a.ctypes.data % 4 == 0
    return a


def set_tables_small(on):
    """the table stage of the harness then takes the small-batch path (tabj_pass_a / tabj_pass_b: Jacobian chain per job, one inversion)"""
    lib().ds_set_tables_small(C.c_int(1 if on else 0))


def set_ingest_two_roles(on):
    """the verify harness then runs the ingest stage in its two-role form (the small-batch kernel k_verify_ingest_split: verify_ingest_a1..a3 / b1..b2)"""
    lib().ds_set_ingest_two_roles(C.c_int(1 if on else 0))


def set_msm_pair(on):
    """the verify harness then runs every long-form chain as two halves on two "lanes" joined by a checked addition (the small-call kernel k_verify_msm_pair)"""
    lib().ds_set_msm_pair(C.c_int(1 if on else 0))


def verify_batch(version, msgs_buf, msg_off, pk, nul, c, s, r_point=None, hr=None, L=3):
    n = len(msg_off) - 1
    ok = np.full(n, 0xEE, dtype=np.uint8)
    pk, nul, c, s, r_point, hr = map(_aligned, (pk, nul, c, s, r_point, hr))
    rc = lib().ds_verify_batch(C.c_int(version), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(pk), _p(nul),
                               _p(c), _p(s), _p(r_point), _p(hr), _p(ok), C.c_int(L))
    assert rc == 0
    return ok


def verify_non_zk_batch(version, msgs_buf, msg_off, pk, nul, s, r_point, hr, digest_private, L=3):
    n = len(msg_off) - 1
    ok = np.full(n, 0xEE, dtype=np.uint8)
    pk, nul, s, r_point, hr, digest_private = map(_aligned, (pk, nul, s, r_point, hr, digest_private))
    rc = lib().ds_verify_non_zk_batch(C.c_int(version), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(pk), _p(nul), _p(s), _p(r_point), _p(hr),
                                      _p(digest_private), _p(ok), C.c_int(L))
    assert rc == 0
    return ok


def verify_batch_bounded(version, msgs_buf, msg_off, msgs_bytes, pk, nul, c, s, r_point=None, hr=None):
    """explicit msgs buffer size: items whose offsets are malformed must be rejected without touching msgs"""
    n = len(msg_off) - 1
    ok = np.full(n, 0xEE, dtype=np.uint8)
    pk, nul, c, s, r_point, hr = map(_aligned, (pk, nul, c, s, r_point, hr))
    rc = lib().ds_verify_batch_bounded(C.c_int(version), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), C.c_uint64(msgs_bytes), _p(pk), _p(nul),
                                       _p(c), _p(s), _p(r_point), _p(hr), _p(ok))
    assert rc == 0
    return ok


def verify_batch_sec1(version, msgs_buf, msg_off, pk33, nul33, c, s, r33=None, hr33=None):
    n = len(msg_off) - 1
    ok = np.full(n, 0xEE, dtype=np.uint8)
    c, s = map(_aligned, (c, s))
    f = lambda a: None if a is None else np.ascontiguousarray(a)  # noqa: E731  (33-byte records: no alignment requirement)
    rc = lib().ds_verify_batch_sec1(C.c_int(version), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(f(pk33)), _p(f(nul33)), _p(c), _p(s), _p(f(r33)), _p(f(hr33)), _p(ok))
    assert rc == 0
    return ok


def sign_batch(version, msgs_buf, msg_off, sk, r, pk_in=None, L=3, uniform=False):
    n = len(msg_off) - 1
    lib().ds_set_sign_uniform(C.c_int(int(uniform)))
    o = {k: np.zeros((n, w), dtype=np.uint8) for k, w in
         [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64), ("h", 64)]}
    status = np.zeros(n, dtype=np.uint8)
    sk, r, pk_in = map(_aligned, (sk, r, pk_in))
    rc = lib().ds_sign_batch(C.c_int(version), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(sk), _p(r), _p(pk_in),
                             _p(o["pk"]), _p(o["nullifier"]), _p(o["c"]), _p(o["s"]), _p(o["r_point"]), _p(o["hashed_to_curve_r"]), _p(o["h"]),
                             _p(status), C.c_int(L))
    lib().ds_set_sign_uniform(C.c_int(0))
    assert rc == 0
    o["status"] = status
    return o


def h2c_batch(msgs_buf, msg_off, pk):
    n = len(msg_off) - 1
    h = np.zeros((n, 64), dtype=np.uint8)
    lib().ds_h2c_batch(C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(_aligned(pk)), _p(h))
    return h


def h2c_intermediates(msgs_buf, msg_off, pk=None, registers=False):
    n = len(msg_off) - 1
    o = {k: np.zeros((n, w, 32), dtype=np.uint8) for k, w in [("u", 2), ("mapped", 4), ("q", 4), ("h", 2), ("hints", 6)]}
    rc = lib().ds_h2c_intermediates(C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(_aligned(pk)), C.c_int(1 if registers else 0), _p(o["u"]), _p(o["mapped"]), _p(o["q"]), _p(o["h"]),
                                    _p(o["hints"]))
    assert rc == 0
    return {k: v.view(np.uint64) for k, v in o.items()} if registers else o


def map2_to_curve(u0: int, u1: int) -> bytes:
    out = (C.c_uint8 * 64)()
    lib().ds_map2_to_curve((C.c_uint8 * 32).from_buffer_copy(u0.to_bytes(32, "big")), (C.c_uint8 * 32).from_buffer_copy(u1.to_bytes(32, "big")), out)
    return bytes(out)


def scalars_to_der(scalars):
    sc = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    der, st = np.zeros((len(sc), 109), dtype=np.uint8), np.zeros(len(sc), dtype=np.uint8)
    lib().ds_scalars_to_der(C.c_uint32(len(sc)), _p(_aligned(sc)), _p(der), _p(st))
    return der, st


def registers_from_be(values):
    v = np.ascontiguousarray(values, dtype=np.uint8)
    out = np.zeros(v.shape, dtype=np.uint8)
    lib().ds_registers_from_be(C.c_size_t(v.size // 32), _p(v), _p(out))
    return out.view(np.uint64)


def tables_raw(points):
    """window tables (rows P, theta P = P - lambda P, 2P as 64-byte records) of raw, UNVALIDATED affine bases built by one lane"""
    pts = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 64)
    lib().ds_tab_entries.restype = C.c_uint32
    out = np.zeros((len(pts), int(lib().ds_tab_entries()), 64), dtype=np.uint8)
    lib().ds_tables_raw(C.c_uint32(len(pts)), _p(pts), _p(out))
    return out


def point_mul(k: bytes, p: bytes):
    out = (C.c_uint8 * 64)()
    ok = lib().ds_point_mul((C.c_uint8 * 32).from_buffer_copy(k), (C.c_uint8 * 64).from_buffer_copy(p), out)
    return bytes(out) if ok else None


def eq1(s: bytes, c: bytes, pk: bytes):
    out = (C.c_uint8 * 64)()
    ok = lib().ds_eq1((C.c_uint8 * 32).from_buffer_copy(s), (C.c_uint8 * 32).from_buffer_copy(c), (C.c_uint8 * 64).from_buffer_copy(pk), out)
    return bytes(out) if ok else None


def set_eq1_short(v):
    """the verifier's first equation where R is given: 1 = short form (plume_eis.h, the default), 0 = long form always, 2 = every item takes the scalar stage's fallback
    (filed as long form, run by the checked chain of the redo launch)"""
    lib().ds_set_eq1_short(C.c_int(int(v)))


def eis_half_gcd(cs):
    """the half-GCD in Z[w] for a list of challenges (ints mod n): [(t0 - 1, t1, u0, u1, tau mod n, ok)]"""
    m = len(cs)
    cb = np.frombuffer(b"".join(int(c).to_bytes(32, "big") for c in cs), dtype=np.uint8).copy()
    out, tau, ok = np.zeros(64 * m, np.uint8), np.zeros(32 * m, np.uint8), np.zeros(m, np.uint8)
    lib().ds_eis_half_gcd(C.c_uint32(m), _p(cb), _p(out), _p(tau), _p(ok))
    res = []
    for i in range(m):
        v = [int.from_bytes(out[64 * i + 16 * k:64 * i + 16 * k + 16].tobytes(), "little", signed=True) for k in range(4)]
        res.append((v[0], v[1], v[2], v[3], int.from_bytes(tau[32 * i:32 * i + 32].tobytes(), "big"), bool(ok[i])))
    return res


def eis_consistent(cs, which=0):
    """eis_consistent on the half-GCD's pair of each challenge, after tamper `which` (0 = none; tests/devsim/devsim.cpp ds_eis_consistent)"""
    m = len(cs)
    cb = np.frombuffer(b"".join(int(c).to_bytes(32, "big") for c in cs), dtype=np.uint8).copy()
    out = np.zeros(m, np.uint8)
    lib().ds_eis_consistent(C.c_uint32(m), _p(cb), C.c_int(which), _p(out))
    return out.astype(bool)


def eq1_short(s: bytes, c: bytes, pk: bytes, r: bytes):
    """k G - upsilon pk - (tau - 1) R through the scalar stage, the table stage and the multi-scalar body; (64 bytes | None, the scalar stage fell back to the long form)"""
    out = (C.c_uint8 * 64)()
    lng = C.c_int(0)
    ok = lib().ds_eq1_short((C.c_uint8 * 32).from_buffer_copy(s), (C.c_uint8 * 32).from_buffer_copy(c), (C.c_uint8 * 64).from_buffer_copy(pk), (C.c_uint8 * 64).from_buffer_copy(r), out,
                            C.byref(lng))
    return (bytes(out) if ok else None), bool(lng.value)


def fallback_count():
    """multi-scalar chains redone with checked additions so far (p == +-q met inside an unchecked chain)"""
    lib().ds_fallback_count.restype = C.c_ulong
    return int(lib().ds_fallback_count())


def nullifier_first_occurrence(nul, live=None, ids=None, order=None):
    nul = np.ascontiguousarray(nul, dtype=np.uint8).reshape(-1, 64)
    n = len(nul)
    buf = _aligned(nul)
    first = np.zeros(n, dtype=np.uint8)
    cnt = C.c_uint64(0)
    live = None if live is None else np.ascontiguousarray(live, dtype=np.uint8)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
    order = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
    lib().ds_nullifier_first_occurrence(C.c_uint32(n), _p(buf), _p(live), _p(ids, u64p), _p(order, u32p), _p(first), C.byref(cnt))
    return first, int(cnt.value)


def aggregate_check(version, mode, msgs_buf, msg_off, pk, nul, c, s, r_point, hr, seed: bytes, index_base=0, W=0, carry=None):
    """plume_aggregate.h's per-lane bodies in the library's launch order; returns (record 72 B, hash_ok)"""
    n = len(msg_off) - 1
    hash_ok = np.full(n, 0xEE, dtype=np.uint8)
    rec = np.zeros(72 + 8, dtype=np.uint8)[:72]
    pk, nul, c, s, r_point, hr = map(_aligned, (pk, nul, c, s, r_point, hr))
    sd = np.frombuffer(seed, dtype=np.uint8).copy()
    cr = None if carry is None else _aligned(np.asarray(carry, dtype=np.uint8).copy())
    rc = lib().ds_aggregate_check(C.c_int(version), C.c_int(mode), C.c_uint32(n), _p(msgs_buf), _p(msg_off, u64p), _p(pk), _p(nul), _p(c), _p(s), _p(r_point), _p(hr),
                                  _p(sd), C.c_uint64(index_base), C.c_int(W), _p(cr), _p(hash_ok), _p(rec))
    assert rc == 0
    return rec.copy(), hash_ok


def aggregate_combine(records):
    recs = _aligned(np.concatenate([np.asarray(r, dtype=np.uint8) for r in records]))
    out = np.zeros(72, dtype=np.uint8)
    lib().ds_aggregate_combine(_p(recs), C.c_uint32(len(records)), _p(out))
    return out
