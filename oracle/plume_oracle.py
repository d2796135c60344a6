"""PLUME (ERC-7524) on secp256k1 — pure-Python ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (and only as
the checker).  The product path (zk-nullifier-sig_amd/) never imports it and fails loudly without the HIP
library.

This is a CPU restatement of the reference's sign/verify path with Python big integers + hashlib:

  * control flow, ordering, encodings, edge semantics: rust-k256/src/lib.rs:93-168 (verify, c-hash),
    rust-k256/src/utils.rs:11-25 (hash_to_curve, encode_pt), rust-k256/src/randomizedsigner.rs:43-112 (sign),
    rust-arkworks/src/lib.rs:76-163,229-278 (sec1_affine, compute_c_v1/v2, sign_with_r),
    rust-arkworks/src/tests.rs:28-78 (verify_non_zk);
  * expand_message_xmd / hash_to_field: rust-arkworks/src/fixed_hasher/expander.rs:89-134, mod.rs:32-62;
  * every constant: rust-arkworks/src/secp256k1/{fields/fq.rs:12, fields/fr.rs:19, curves/mod.rs:36-112},
    DST rust-k256/src/lib.rs:61;
  * the arithmetic itself lives in the un-vendored crate k256 ~0.13.3 (rust-k256/Cargo.toml:18; with
    elliptic-curve 0.13 hash2curve and sha2 0.10), which is NOT under /root/reference and cannot be built here
    (no rustc/cargo).  Its published algorithm is restated from the specs it implements: RFC 9380 (suite
    secp256k1_XMD:SHA-256_SSWU_RO_: §5.3.1 expand_message_xmd, §5.2 hash_to_field, §6.6.2/F.2 simplified SWU
    with F.2.1.2 sqrt_ratio_3mod4, App. E.1 3-isogeny), SEC1 §2.3.3 point compression, FIPS 180-4 SHA-256.

Parity is PINNED: tests/test_oracle_kats.py checks this module against every known-answer vector the
reference's own tests hold for this path (tests/golden/reference_kats.json, extracted by
tests/golden/make_reference_kats.py): the full V1+V2 signature of the fixed (sk, r, msg) triple and all its
intermediates, the literal 62-byte h2c preimage, h2c("abc"), the RFC 9380 J.8.1 vector (u0,u1,Q0,Q1,P),
enc(G) and the 100 k*G SEC1 vectors.
"""
import hashlib
from typing import List, Optional, Tuple

# ------------------------------------------------------------------------------------------------ constants
P = 2**256 - 2**32 - 977                       # rust-arkworks/src/secp256k1/fields/fq.rs:12
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141  # fields/fr.rs:19
B = 7                                          # curves/mod.rs:39
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798  # curves/mod.rs:52-53
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8  # curves/mod.rs:57-58
ISO_A = 0x3F8731ABDD661ADCA08A5558F0F5D272E953D363CB6F0E5D405447C01A444533  # curves/mod.rs:71-72
ISO_B = 1771                                   # curves/mod.rs:73
Z = P - 11                                     # curves/mod.rs:80
# 3-isogeny E' -> E, ascending degree (curves/mod.rs:88-111)
ISO_XNUM = [0x8E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38DAAAAA8C7,
            0x07D3D4C80BC321D5B9F315CEA7FD44C5D595D2FC0BF63B92DFFF1044F17C6581,
            0x534C328D23F234E6E2A413DECA25CAECE4506144037C40314ECBD0B53D9DD262,
            0x8E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38E38DAAAAA88C]
ISO_XDEN = [0xD35771193D94918A9CA34CCBB7B640DD86CD409542F8487D9FE6B745781EB49B,
            0xEDADC6F64383DC1DF7C4B2D51B54225406D36B641F5E41BBC52A56612A8C6D14,
            1, 0]
ISO_YNUM = [0x4BDA12F684BDA12F684BDA12F684BDA12F684BDA12F684BDA12F684B8E38E23C,
            0xC75E0C32D5CB7C0FA9D0A54B12A0A6D5647AB046D686DA6FDFFC90FC201D71A3,
            0x29A6194691F91A73715209EF6512E576722830A201BE2018A765E85A9ECEE931,
            0x2F684BDA12F684BDA12F684BDA12F684BDA12F684BDA12F684BDA12F38E38D84]
ISO_YDEN = [0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFF93B,
            0x7A06534BB8BDB49FD5E9E6632722C2989467C1BFC8E8D978DFB425D2685C2573,
            0x6484AA716545CA2CF3A70C3FA8FE337E0A3D21162F0D6299A7BF8192BFD2A76F,
            1]
DST = b"QUUX-V01-CS02-with-secp256k1_XMD:SHA-256_SSWU_RO_"   # rust-k256/src/lib.rs:61
L_H2F = 48                                     # fixed_hasher/mod.rs:53-62: ceil((256+128)/8)

Point = Optional[Tuple[int, int]]              # None = identity
G: Point = (GX, GY)


# ---------------------------------------------------------------------------------------------- curve group
def is_on_curve(pt: Point) -> bool:
    if pt is None:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


def pt_neg(a: Point) -> Point:
    return None if a is None else (a[0], (-a[1]) % P)


def pt_add(a: Point, b: Point) -> Point:
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def pt_mul(k: int, a: Point) -> Point:
    """k*a by plain double-and-add (the reference's `ProjectivePoint * Scalar`, rust-k256/src/lib.rs:101,109)."""
    k %= N
    acc = None
    while k:
        if k & 1:
            acc = pt_add(acc, a)
        a = pt_add(a, a)
        k >>= 1
    return acc


def sec1_compress(pt: Point) -> bytes:
    """encode_pt (rust-k256/src/utils.rs:23-25) == helper(sec1_affine()) (rust-arkworks/src/lib.rs:76-88,112-118):
    02|03 by parity of y then x big-endian; the identity is the single byte 00."""
    if pt is None:
        return b"\x00"
    return bytes([2 + (pt[1] & 1)]) + pt[0].to_bytes(32, "big")


def sec1_decompress(b: bytes) -> Point:
    """inverse of sec1_compress (used only for the wasm README vector and the 'next' 33-byte ingest row)."""
    if b == b"\x00":
        return None
    assert len(b) == 33 and b[0] in (2, 3)
    x = int.from_bytes(b[1:], "big")
    assert x < P
    y = pow((x * x * x + B) % P, (P + 1) // 4, P)
    assert (y * y - x * x * x - B) % P == 0, "x not on curve"
    if (y & 1) != (b[0] & 1):
        y = P - y
    return (x, y)


# ------------------------------------------------------------------------------------------ hash_to_curve
def expand_message_xmd(msg: bytes, dst: bytes, n: int) -> bytes:
    """rust-arkworks/src/fixed_hasher/expander.rs:89-134 (RFC 9380 §5.3.1) with H = SHA-256."""
    b_len, block = 32, 64
    ell = (n + b_len - 1) // b_len
    assert ell <= 255 and n < 65536 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])                      # expander.rs:53-57
    b0 = hashlib.sha256(bytes(block) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()   # :107-113
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()   # :115-118
    out = bi
    for i in range(2, ell + 1):                              # :122-131
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]


def hash_to_field2(msg: bytes, dst: bytes = DST) -> Tuple[int, int]:
    """rust-arkworks/src/fixed_hasher/mod.rs:32-50 with N=2, m=1, L=48."""
    u = expand_message_xmd(msg, dst, 2 * L_H2F)
    return (int.from_bytes(u[:L_H2F], "big") % P, int.from_bytes(u[L_H2F:], "big") % P)


def _sgn0(x: int) -> int:
    return x & 1


_C1 = (P - 3) // 4
_C2 = pow((-Z) % P, (P + 1) // 4, P)       # sqrt(-Z); -Z = 11 is a square mod p
assert _C2 * _C2 % P == (-Z) % P


def sqrt_ratio_3mod4(u: int, v: int) -> Tuple[bool, int]:
    """RFC 9380 F.2.1.2."""
    tv1 = v * v % P
    tv2 = u * v % P
    tv1 = tv1 * tv2 % P
    y1 = pow(tv1, _C1, P)
    y1 = y1 * tv2 % P
    y2 = y1 * _C2 % P
    tv3 = y1 * y1 % P
    tv3 = tv3 * v % P
    is_qr = tv3 == u % P
    return is_qr, (y1 if is_qr else y2)


def map_to_curve_sswu(u: int) -> Tuple[int, int]:
    """RFC 9380 F.2 straight-line simplified SWU on E': y^2 = x^3 + A'x + B' (curves/mod.rs:70-81)."""
    A, Bc = ISO_A, ISO_B
    tv1 = u * u % P
    tv1 = Z * tv1 % P
    tv2 = tv1 * tv1 % P
    tv2 = (tv2 + tv1) % P
    tv3 = (tv2 + 1) % P
    tv3 = Bc * tv3 % P
    tv4 = Z if tv2 == 0 else (-tv2) % P
    tv4 = A * tv4 % P
    tv2 = tv3 * tv3 % P
    tv6 = tv4 * tv4 % P
    tv5 = A * tv6 % P
    tv2 = (tv2 + tv5) % P
    tv2 = tv2 * tv3 % P
    tv6 = tv6 * tv4 % P
    tv5 = Bc * tv6 % P
    tv2 = (tv2 + tv5) % P
    x = tv1 * tv3 % P
    is_gx1_square, y1 = sqrt_ratio_3mod4(tv2, tv6)
    y = tv1 * u % P
    y = y * y1 % P
    if is_gx1_square:
        x, y = tv3, y1
    if _sgn0(u) != _sgn0(y):
        y = (-y) % P
    x = x * pow(tv4, -1, P) % P
    return x, y


def sswu_hints(u: int) -> Tuple[int, int, int]:
    """(gx1_sqrt, gx2_sqrt, y_pos) of one simplified-SWU map as include/plume_hip.h DEFINES them for plume_h2c_hints_batch (UNPINNED: the reference's generator of these
    circuit inputs, secp256k1_hash_to_curve_circom/ts/generate_inputs, is not in the reference tree).  Written from RFC 9380 F.2's (non-straight-line) description,
    independently of map_to_curve_sswu above: x1, gx1, x2, gx2 by their formulas, roots by exponentiation, every choice explicit."""
    A, Bc = ISO_A, ISO_B
    d = (Z * Z * pow(u, 4, P) + Z * u * u) % P
    x1 = Bc * pow(Z * A % P, -1, P) % P if d == 0 else (-Bc) * pow(A, -1, P) % P * (1 + pow(d, -1, P)) % P
    gx = lambda x: (pow(x, 3, P) + A * x + Bc) % P  # noqa: E731
    x2 = Z * u * u % P * x1 % P
    g1, g2 = gx(x1), gx(x2)
    is_sq = lambda v: v == 0 or pow(v, (P - 1) // 2, P) == 1  # noqa: E731

    def even_root(v):
        r = pow(v, (P + 1) // 4, P)
        assert r * r % P == v % P
        return r if r % 2 == 0 else P - r
    assert is_sq(g1) != is_sq(g2) or g1 == 0
    r1 = even_root(g1) if is_sq(g1) else even_root(Z * g1 % P)
    r2 = even_root(g2) if is_sq(g2) else even_root(Z * g2 % P)
    y = even_root(g1 if is_sq(g1) else g2)
    if y % 2 != u % 2:
        y = P - y
    return r1, r2, y


def _poly(coeffs: List[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def iso_map(pt: Tuple[int, int]) -> Point:
    """RFC 9380 App. E.1; coefficient tables curves/mod.rs:87-112. A zero denominator maps to the identity."""
    x, y = pt
    xn, xd, yn, yd = _poly(ISO_XNUM, x), _poly(ISO_XDEN, x), _poly(ISO_YNUM, x), _poly(ISO_YDEN, x)
    if xd == 0 or yd == 0:
        return None
    return (xn * pow(xd, -1, P) % P, y * yn % P * pow(yd, -1, P) % P)


def hash_to_curve_bytes(data: bytes, dst: bytes = DST, want_intermediates: bool = False):
    """Secp256k1::hash_from_bytes::<ExpandMsgXmd<Sha256>>(&[data], &[DST]) (rust-k256/src/utils.rs:15-19)."""
    u0, u1 = hash_to_field2(data, dst)
    q0 = iso_map(map_to_curve_sswu(u0))
    q1 = iso_map(map_to_curve_sswu(u1))
    r = pt_add(q0, q1)                       # cofactor 1 (curves/mod.rs:27-32): no clearing
    if want_intermediates:
        return r, (u0, u1, q0, q1)
    return r


def hash_to_curve(msg: bytes, pk: Point) -> Point:
    """rust-k256/src/utils.rs:11-20: h2c over  m || SEC1c(pk)."""
    return hash_to_curve_bytes(msg + sec1_compress(pk))


# ------------------------------------------------------------------------------------------------- PLUME
def c_hash(version: int, pk: Point, h: Point, nul: Point, r_pt: Point, hr: Point) -> bytes:
    """c_sha256_vec_signal (rust-k256/src/lib.rs:159-168) over G,pk,H,nul,R,Hr (V1, :128-135) or nul,R,Hr (V2, :139-143)."""
    pts = [G, pk, h, nul, r_pt, hr] if version == 1 else [nul, r_pt, hr]
    return hashlib.sha256(b"".join(sec1_compress(p) for p in pts)).digest()


STATUS_C_NOT_CANONICAL = 1   # digest == 0 or >= n  (k256 sign panics, randomizedsigner.rs:90-91; arkworks reduces, lib.rs:257)
STATUS_BAD_SCALAR = 2        # sk or r not in [1, n-1] (NonZeroScalar / SecretKey invariants)
STATUS_IDENTITY = 4          # H == identity (randomizedsigner.rs:61) or s == 0 (:95)


def sign(version: int, sk: int, r: int, msg: bytes, pk: Point = "derive"):
    """PlumeSigner::try_sign_with_rng with the nonce given (rust-k256/src/randomizedsigner.rs:43-112), which is
    also plume_arkworks::sign_with_r (rust-arkworks/src/lib.rs:229-278) when `pk` is supplied.
    Returns dict(pk, h, nullifier, c (int, digest mod n), s, r_point, hashed_to_curve_r, status, digest)."""
    status = 0
    if not (1 <= sk < N and 1 <= r < N):
        status |= STATUS_BAD_SCALAR
    r_point = pt_mul(r, G)                                        # :51
    if pk == "derive":
        pk = pt_mul(sk, G)                                        # :53
    h = hash_to_curve(msg, pk)                                    # :57-61
    if h is None:
        status |= STATUS_IDENTITY
    hr = pt_mul(r, h)                                             # :67
    nul = pt_mul(sk, h)                                           # :70
    digest = c_hash(version, pk, h, nul, r_point, hr)             # :73-89
    d = int.from_bytes(digest, "big")
    if d == 0 or d >= N:
        status |= STATUS_C_NOT_CANONICAL                          # :90-91
    c = d % N
    s = (r + sk * c) % N                                          # :94
    if s == 0:
        status |= STATUS_IDENTITY                                 # :95
    return dict(pk=pk, h=h, nullifier=nul, c=c, s=s, r_point=r_point, hashed_to_curve_r=hr, status=status, digest=digest)


def verify(version: int, msg: bytes, pk: Point, nul: Point, c: int, s: int,
           r_point: Point = None, hashed_to_curve_r: Point = None) -> bool:
    """PlumeSignature::verify (rust-k256/src/lib.rs:93-145).  Input invariants that the Rust types enforce
    (c, s NonZeroScalar; points on curve) are explicit here: a violation returns False."""
    if not (1 <= c < N and 1 <= s < N):
        return False
    pts = [pk, nul] + ([r_point, hashed_to_curve_r] if version == 1 else [])
    if not all(is_on_curve(p) for p in pts):
        return False
    r_calc = pt_add(pt_mul(s, G), pt_neg(pt_mul(c, pk)))          # :101
    h = hash_to_curve(msg, pk)                                    # :103
    hr_calc = pt_add(pt_mul(s, h), pt_neg(pt_mul(c, nul)))        # :109
    if version == 1:
        if r_calc != r_point:                                     # :117
            return False
        if hr_calc != hashed_to_curve_r:                          # :122
            return False
    digest = c_hash(version, pk, h, nul, r_calc, hr_calc)         # :127-143
    return c == int.from_bytes(digest, "big") % N                 # Scalar::reduce, :128,:139


class HashToCurveError(Exception):
    """rust-arkworks/src/lib.rs:99-101: `pk` shouldn't be the identity element"""


def verify_non_zk(version: int, msg: bytes, pk: Point, nul: Point, s: int, r_point: Point, hr: Point, digest_private: int) -> bool:
    """rust-arkworks/src/tests.rs:28-78: c' hashed from the GIVEN R, Hr; both EC equations checked for V1 and V2.
    Raises HashToCurveError where the reference returns Err (pk = identity, tests.rs:36 -> lib.rs:99-101)."""
    if pk is None:
        raise HashToCurveError("`pk` shouldn't be the identity element")
    h = hash_to_curve(msg, pk)
    c2 = int.from_bytes(c_hash(version, pk, h, nul, r_point, hr), "big") % N
    if r_point != pt_add(pt_mul(s, G), pt_neg(pt_mul(digest_private, pk))):
        return False
    if hr != pt_add(pt_mul(s, h), pt_neg(pt_mul(digest_private, nul))):
        return False
    return c2 == digest_private % N


# ---------------------------------------------------------------------------- synthetic batches
SEED = 0x504C554D45


def blk(tag: str, i: int, seed: int = SEED) -> bytes:
    return hashlib.sha256(tag.encode() + seed.to_bytes(8, "little") + i.to_bytes(8, "little")).digest()


def synth_sk(i: int, seed: int = SEED) -> int:
    return int.from_bytes(blk("sk", i, seed), "big") % (N - 1) + 1


def synth_r(i: int, seed: int = SEED) -> int:
    return int.from_bytes(blk("r", i, seed), "big") % (N - 1) + 1


def synth_msg(i: int, seed: int = SEED) -> bytes:
    return blk("msg", i, seed)


def pt_bytes(pt: Point) -> bytes:
    """64-byte affine x||y big-endian; all-zero = identity (the C-ABI point format, include/plume_hip.h)."""
    return bytes(64) if pt is None else pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def pt_from_bytes(b: bytes) -> Point:
    if b == bytes(64):
        return None
    return (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))


# ------------------------------------------------------------------------------------------------------------------
# Nullifier-set post-processing.  The reference has no such function: PLUME's stated purpose is one
# nullifier per (pk, message) (README.md:5), and this is the check its consumers run on a verified batch.  Plain definition:
def nullifier_first_occurrence(nullifiers, live=None, ids=None):
    """nullifiers: sequence of 64-byte records; live: optional sequence of truthy/falsy; ids: optional distinct integers (default
    position).  Returns (first flags list, count): first[i] = live[i] and no live j with the same record and a smaller id."""
    n = len(nullifiers)
    best = {}
    for i in range(n):
        if live is not None and not live[i]:
            continue
        key = bytes(nullifiers[i])
        ident = i if ids is None else int(ids[i])
        if key not in best or ident < best[key][0]:
            best[key] = (ident, i)
    first = [0] * n
    for ident, i in best.values():
        first[i] = 1
    return first, len(best)
