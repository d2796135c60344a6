"""Independent third-party cross-check of the oracles' group arithmetic: OpenSSL libcrypto (NID_secp256k1, EC_POINT_mul)
through ctypes.  OpenSSL has no hash_to_curve, so this covers k*G, k*P and a*G + b*P — the operations the PLUME path is
built from — for the Python oracle, the C oracle and (through them) everything pinned to them."""
import ctypes as C
import ctypes.util
import random

import pytest

from oracle import plume_oracle as O
from tests import _oracle_c as OC

_path = ctypes.util.find_library("crypto")
pytestmark = pytest.mark.skipif(_path is None, reason="libcrypto not available")
NID_secp256k1 = 714


class Ossl:
    def __init__(self):
        L = C.CDLL(_path)
        for f, res, args in [("EC_GROUP_new_by_curve_name", C.c_void_p, [C.c_int]), ("EC_POINT_new", C.c_void_p, [C.c_void_p]),
                             ("BN_new", C.c_void_p, []), ("BN_CTX_new", C.c_void_p, []), ("BN_bin2bn", C.c_void_p, [C.c_char_p, C.c_int, C.c_void_p]),
                             ("EC_POINT_mul", C.c_int, [C.c_void_p] * 6), ("EC_POINT_set_affine_coordinates", C.c_int, [C.c_void_p] * 5),
                             ("EC_POINT_get_affine_coordinates", C.c_int, [C.c_void_p] * 5), ("BN_bn2binpad", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
                             ("EC_POINT_is_at_infinity", C.c_int, [C.c_void_p, C.c_void_p]), ("EC_POINT_free", None, [C.c_void_p]), ("BN_free", None, [C.c_void_p])]:
            fn = getattr(L, f)
            fn.restype, fn.argtypes = res, args
        self.L = L
        self.g = L.EC_GROUP_new_by_curve_name(NID_secp256k1)
        assert self.g
        self.ctx = L.BN_CTX_new()

    def bn(self, v):
        b = v.to_bytes(32, "big")
        return self.L.BN_bin2bn(b, 32, None)

    def lincomb(self, a, p, b):
        """a*G + b*P (P affine tuple or None -> only a*G)"""
        L = self.L
        r = L.EC_POINT_new(self.g)
        q = None
        bn_a, bn_b = self.bn(a), (self.bn(b) if p is not None else None)
        if p is not None:
            q = L.EC_POINT_new(self.g)
            x, y = self.bn(p[0]), self.bn(p[1])
            assert L.EC_POINT_set_affine_coordinates(self.g, q, x, y, self.ctx) == 1
            L.BN_free(x); L.BN_free(y)
        assert L.EC_POINT_mul(self.g, r, bn_a, q, bn_b, self.ctx) == 1
        out = None
        if not L.EC_POINT_is_at_infinity(self.g, r):
            x, y = L.BN_new(), L.BN_new()
            assert L.EC_POINT_get_affine_coordinates(self.g, r, x, y, self.ctx) == 1
            bx, by = C.create_string_buffer(32), C.create_string_buffer(32)
            L.BN_bn2binpad(x, bx, 32); L.BN_bn2binpad(y, by, 32)
            out = (int.from_bytes(bx.raw, "big"), int.from_bytes(by.raw, "big"))
            L.BN_free(x); L.BN_free(y)
        for z in (bn_a, bn_b):
            if z:
                L.BN_free(z)
        L.EC_POINT_free(r)
        if q:
            L.EC_POINT_free(q)
        return out


def test_scalar_multiplications_agree_with_openssl():
    ossl = Ossl()
    rng = random.Random(2024)
    ks = [1, 2, 3, O.N - 1, O.N - 2, 2**128, 2**255 % O.N] + [rng.randrange(1, O.N) for _ in range(12)]
    g64 = O.pt_bytes(O.G)
    for k in ks:
        want = ossl.lincomb(k, None, 0)
        assert O.pt_mul(k, O.G) == want
        assert OC.point_mul(k.to_bytes(32, "big"), g64) == O.pt_bytes(want)
    p = O.pt_mul(rng.randrange(1, O.N), O.G)
    for _ in range(6):
        a, b = rng.randrange(1, O.N), rng.randrange(1, O.N)
        want = ossl.lincomb(a, p, b)                                   # a*G + b*P: the shape of  s*G - c*pk  (b = n - c)
        assert O.pt_add(O.pt_mul(a, O.G), O.pt_mul(b, p)) == want
        got_c = OC.point_mul(b.to_bytes(32, "big"), O.pt_bytes(p))
        assert O.pt_add(O.pt_mul(a, O.G), O.pt_from_bytes(got_c)) == want
    assert ossl.lincomb(O.N - 5, O.pt_mul(5, O.G), 1) is None          # (n-5)*G + 5*G = identity


def test_reference_vector_points_agree_with_openssl(kats):
    """pk = sk*G and g^r of the reference's fixed vector, recomputed by OpenSSL"""
    ossl = Ossl()
    v = kats["plume_vector"]
    assert ossl.lincomb(int(v["sk"], 16), None, 0) == (int(v["pk_x"], 16), int(v["pk_y"], 16))
    assert ossl.lincomb(int(v["r"], 16), None, 0) == (int(v["g_r_x"], 16), int(v["g_r_y"], 16))
    h = (int(v["h_x"], 16), int(v["h_y"], 16))
    # nullifier = sk*H and H^r: EC_POINT_mul always includes a generator term, so check (1*G + k*H) - G
    t = ossl.lincomb(1, h, int(v["sk"], 16))
    assert O.pt_add(t, O.pt_neg(O.G)) == (int(v["nullifier_x"], 16), int(v["nullifier_y"], 16))
    t = ossl.lincomb(1, h, int(v["r"], 16))
    assert O.pt_add(t, O.pt_neg(O.G)) == (int(v["h_r_x"], 16), int(v["h_r_y"], 16))
