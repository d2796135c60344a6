"""Host façade with the reference's API shape, computing through the MI355X engine (capi.Engine) only.

plume_rustcrypto shape (rust-k256/src/lib.rs:43-156, rust-k256/src/randomizedsigner.rs:25-112):
    DST, AffinePoint, NonZeroScalar, SecretKey, PlumeSignature{message, pk, nullifier, c, s, v1specific},
    PlumeSignatureV1Fields{r_point, hashed_to_curve_r}, PlumeSignature.verify(), PlumeSignature.sign_v1 / sign_v2,
    PlumeSigner(secret_key, v1).try_sign_with_rng(rng, msg) / sign_with_rng(rng, msg)
plume_arkworks shape (rust-arkworks/src/lib.rs:66-69,185-201,229-291):
    PlumeVersion, PlumeSignaturePublic, PlumeSignaturePrivate, sign_with_r(keypair, message, r, version), sign(rng, ...)

Everything cryptographic (hash_to_curve, the scalar multiplications, the c-hash, s = r + sk*c) runs in the HIP
kernels; this module only marshals bytes and reproduces the reference's error behaviour:
  * `expect(..)` panics of the signer (randomizedsigner.rs:61,91,95) -> PlumePanic
  * `signature::Error` (randomizedsigner.rs:59) -> SignatureError (unreachable for this DST, as in the reference)
  * NonZeroScalar / on-curve invariants of the Rust types -> ValueError at construction
A single sign/verify is a batch of one; callers with many signatures should use Engine.verify_batch / sign_batch.
"""
from dataclasses import dataclass
from enum import Enum
from typing import Optional, Tuple

import numpy as np

from .capi import Engine, default_engine, pack_messages

DST = b"QUUX-V01-CS02-with-secp256k1_XMD:SHA-256_SSWU_RO_"  # rust-k256/src/lib.rs:61
_P = 2**256 - 2**32 - 977
_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


class PlumePanic(RuntimeError):
    """the reference signer would `panic!` here (randomizedsigner.rs:61,91,95)"""


class SignatureError(Exception):
    """signature::Error (randomizedsigner.rs:59)"""


@dataclass(frozen=True)
class AffinePoint:
    """k256::AffinePoint: on-curve point or the identity (x = y = None)."""
    x: Optional[int] = None
    y: Optional[int] = None

    def __post_init__(self):
        if (self.x is None) != (self.y is None):
            raise ValueError("both coordinates or none")
        if self.x is not None:
            if not (0 <= self.x < _P and 0 <= self.y < _P) or (self.y * self.y - self.x**3 - 7) % _P:
                raise ValueError("point is not on secp256k1")

    @property
    def is_identity(self):
        return self.x is None

    def to_bytes64(self) -> bytes:
        return bytes(64) if self.x is None else self.x.to_bytes(32, "big") + self.y.to_bytes(32, "big")

    @staticmethod
    def from_bytes64(b: bytes) -> "AffinePoint":
        b = bytes(b)
        if b == bytes(64):
            return AffinePoint()
        return AffinePoint(int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))

    def to_encoded_point(self, compress: bool = True) -> bytes:
        """SEC1 (encode_pt, rust-k256/src/utils.rs:23-25): identity = single 00"""
        if self.x is None:
            return b"\x00"
        if compress:
            return bytes([2 + (self.y & 1)]) + self.x.to_bytes(32, "big")
        return b"\x04" + self.to_bytes64()

    @staticmethod
    def generator() -> "AffinePoint":
        return AffinePoint(0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
                           0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


@dataclass(frozen=True)
class NonZeroScalar:
    """k256::NonZeroScalar: integer in [1, n-1]"""
    value: int

    def __post_init__(self):
        if not (1 <= self.value < _N):
            raise ValueError("scalar must be in [1, n-1]")

    def to_bytes(self) -> bytes:
        return self.value.to_bytes(32, "big")

    @staticmethod
    def from_repr(b: bytes) -> "NonZeroScalar":
        return NonZeroScalar(int.from_bytes(bytes(b), "big"))


class SecretKey(NonZeroScalar):
    """k256::SecretKey"""

    @staticmethod
    def from_bytes(b: bytes) -> "SecretKey":
        return SecretKey(int.from_bytes(bytes(b), "big"))

    @staticmethod
    def random(rng) -> "SecretKey":
        """SecretKey::random: 32 bytes from rng.fill_bytes, big-endian, rejection-sampled (pinned by the mock RNG of
        rust-k256/tests/signing.rs:23-44)."""
        while True:
            b = rng.fill_bytes(32)
            v = int.from_bytes(b, "big")
            if 1 <= v < _N:
                return SecretKey(v)


@dataclass
class PlumeSignatureV1Fields:  # rust-k256/src/lib.rs:84-89
    r_point: AffinePoint
    hashed_to_curve_r: AffinePoint


@dataclass
class PlumeSignature:  # rust-k256/src/lib.rs:67-80
    message: bytes
    pk: AffinePoint
    nullifier: AffinePoint
    c: NonZeroScalar
    s: NonZeroScalar
    v1specific: Optional[PlumeSignatureV1Fields] = None

    def verify(self, engine: Optional[Engine] = None) -> bool:
        """PlumeSignature::verify (rust-k256/src/lib.rs:93-145) on the GPU."""
        eng = engine or default_engine()
        msgs, off = pack_messages([bytes(self.message)])
        a = lambda b: np.frombuffer(b, dtype=np.uint8)  # noqa: E731
        v1 = self.v1specific
        ok = eng.verify_batch(1 if v1 else 2, msgs, off, a(self.pk.to_bytes64()), a(self.nullifier.to_bytes64()), a(self.c.to_bytes()), a(self.s.to_bytes()),
                              a(v1.r_point.to_bytes64()) if v1 else None, a(v1.hashed_to_curve_r.to_bytes64()) if v1 else None)
        return bool(ok[0])

    @staticmethod
    def sign_v1(secret_key: SecretKey, msg: bytes, rng, engine: Optional[Engine] = None) -> "PlumeSignature":  # lib.rs:149-151
        return PlumeSigner(secret_key, True, engine).sign_with_rng(rng, msg)

    @staticmethod
    def sign_v2(secret_key: SecretKey, msg: bytes, rng, engine: Optional[Engine] = None) -> "PlumeSignature":  # lib.rs:154-156
        return PlumeSigner(secret_key, False, engine).sign_with_rng(rng, msg)


class PlumeSigner:  # rust-k256/src/randomizedsigner.rs:25-41
    def __init__(self, secret_key: SecretKey, v1: bool, engine: Optional[Engine] = None):
        self.secret_key = secret_key
        self.v1 = bool(v1)
        self._engine = engine

    def try_sign_with_rng(self, rng, msg: bytes) -> PlumeSignature:  # randomizedsigner.rs:43-112
        eng = self._engine or default_engine()
        r = SecretKey.random(rng)                                    # :49
        msgs, off = pack_messages([bytes(msg)])
        a = lambda b: np.frombuffer(b, dtype=np.uint8)  # noqa: E731
        o = eng.sign_batch(1 if self.v1 else 2, msgs, off, a(self.secret_key.to_bytes()), a(r.to_bytes()))
        st = int(o["status"][0])
        if st & 4 and AffinePoint.from_bytes64(o["nullifier"][0]).is_identity:
            raise PlumePanic("something is drammatically wrong if the input hashed to the identity")               # :61
        if st & 1:
            raise PlumePanic("it should be impossible to get the hash equal to zero")                             # :91
        if st & 4:
            raise PlumePanic("something is terribly wrong if the nonce is equal to negated product of the secret and the hash")  # :95
        pt = lambda k: AffinePoint.from_bytes64(o[k][0].tobytes())  # noqa: E731
        return PlumeSignature(
            message=bytes(msg), pk=pt("pk"), nullifier=pt("nullifier"),
            c=NonZeroScalar.from_repr(o["c"][0].tobytes()), s=NonZeroScalar.from_repr(o["s"][0].tobytes()),
            v1specific=PlumeSignatureV1Fields(pt("r_point"), pt("hashed_to_curve_r")) if self.v1 else None)

    def sign_with_rng(self, rng, msg: bytes) -> PlumeSignature:
        return self.try_sign_with_rng(rng, msg)


# ------------------------------------------------------------------------------------------- plume_arkworks shape
class PlumeVersion(Enum):  # rust-arkworks/src/lib.rs:66-69
    V1 = 1
    V2 = 2


@dataclass
class PlumeSignaturePublic:  # rust-arkworks/src/lib.rs:185-191
    message: bytes
    s: int
    nullifier: AffinePoint
    variant: Optional[PlumeVersion]


@dataclass
class PlumeSignaturePrivate:  # rust-arkworks/src/lib.rs:194-201
    hashed_to_curve_r: AffinePoint
    r_point: AffinePoint
    digest_private: int
    variant: PlumeVersion

    def zeroize(self):  # lib.rs:202-208
        self.digest_private = 0
        self.hashed_to_curve_r = AffinePoint()
        self.r_point = AffinePoint()


def sign_with_r(keypair: Tuple[AffinePoint, int], message: bytes, r_scalar: int, version: PlumeVersion,
                engine: Optional[Engine] = None) -> Tuple[PlumeSignaturePublic, PlumeSignaturePrivate]:
    """plume_arkworks::sign_with_r (rust-arkworks/src/lib.rs:229-278): pk supplied (not recomputed), explicit r,
    c reduced mod n (never panics), result split public / private."""
    eng = engine or default_engine()
    pk, sk = keypair
    if pk.is_identity:
        raise SignatureError("`pk` shouldn't be the identity element")      # lib.rs:99-101
    msgs, off = pack_messages([bytes(message)])
    a = lambda b: np.frombuffer(b, dtype=np.uint8)  # noqa: E731
    o = eng.sign_batch(version.value, msgs, off, a((sk % _N).to_bytes(32, "big")), a((r_scalar % _N).to_bytes(32, "big")), pk_in=a(pk.to_bytes64()))
    pt = lambda k: AffinePoint.from_bytes64(o[k][0].tobytes())  # noqa: E731
    return (PlumeSignaturePublic(bytes(message), int.from_bytes(o["s"][0].tobytes(), "big"), pt("nullifier"), version),
            PlumeSignaturePrivate(pt("hashed_to_curve_r"), pt("r_point"), int.from_bytes(o["c"][0].tobytes(), "big"), version))


def verify_non_zk(sig: Tuple[PlumeSignaturePublic, PlumeSignaturePrivate], pk: AffinePoint, message: bytes, version: PlumeVersion,
                  engine: Optional[Engine] = None) -> bool:
    """plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78) on the GPU: c' from the GIVEN r_point / hashed_to_curve_r, both
    equations g^s pk^-c == g^r and h^s nul^-c == z for V1 and V2, then c' == digest_private.  Raises SignatureError where the reference
    returns Err(HashToCurveError) (pk = identity, rust-arkworks/src/lib.rs:99-101)."""
    eng = engine or default_engine()
    pub, prv = sig
    msgs, off = pack_messages([bytes(message)])
    a = lambda b: np.frombuffer(b, dtype=np.uint8)  # noqa: E731
    ok = eng.verify_non_zk_batch(version.value, msgs, off, a(pk.to_bytes64()), a(pub.nullifier.to_bytes64()), a((pub.s % _N).to_bytes(32, "big")),
                                 a(prv.r_point.to_bytes64()), a(prv.hashed_to_curve_r.to_bytes64()), a((prv.digest_private % _N).to_bytes(32, "big")))
    if int(ok[0]) == 2:
        raise SignatureError("`pk` shouldn't be the identity element")
    return bool(ok[0])


def circuit_inputs(sig: "PlumeSignature", engine: Optional[Engine] = None) -> dict:
    """All inputs of the circom verifier (circuits/circom/verify_nullifier.circom:14-31; test/v1.test.ts:68-78) for one signature, as lists of four 64-bit
    little-endian registers (circuits/circom/utils.ts:11-17): c, s, pk, nullifier from the signature, q{0,1}_x_mapped / q{0,1}_y_mapped from the GPU hash_to_curve
    (pinned), and q{0,1}_gx1_sqrt / gx2_sqrt / y_pos as include/plume_hip.h DEFINES them (UNPINNED: their generator is not vendored in the reference tree)."""
    from .capi import registers_from_be
    eng = engine or default_engine()
    msgs, off = pack_messages([bytes(sig.message)])
    a = lambda b: np.frombuffer(b, dtype=np.uint8)  # noqa: E731
    o = eng.h2c_intermediates_batch(msgs, off, a(sig.pk.to_bytes64()), registers=True)
    reg = lambda b: [int(x) for x in registers_from_be(a(b).reshape(1, 32))[0]]  # noqa: E731
    pt = lambda p: [reg(p.to_bytes64()[:32]), reg(p.to_bytes64()[32:])]  # noqa: E731
    m = o["mapped"][0]
    hints = {k: [int(x) for x in v[0]] for k, v in eng.h2c_hints_batch(msgs, off, a(sig.pk.to_bytes64()), registers=True).items()}
    return {**hints, "c": reg(sig.c.to_bytes()), "s": reg(sig.s.to_bytes()), "plume_message": list(bytes(sig.message)), "pk": pt(sig.pk), "nullifier": pt(sig.nullifier),
            "q0_x_mapped": [int(x) for x in m[0]], "q0_y_mapped": [int(x) for x in m[1]], "q1_x_mapped": [int(x) for x in m[2]], "q1_y_mapped": [int(x) for x in m[3]]}


def sign(rng, keypair: Tuple[AffinePoint, int], message: bytes, version: PlumeVersion, engine: Optional[Engine] = None):
    """plume_arkworks::sign (rust-arkworks/src/lib.rs:281-291): r = Fr::rand(rng)"""
    r = int.from_bytes(rng.fill_bytes(48), "big") % _N
    return sign_with_r(keypair, message, r, version, engine)
