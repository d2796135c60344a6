"""The product's DEVICE code (zk-nullifier-sig_amd/csrc/plume_*.h), compiled for the host, against the oracles.
No GPU needed: these pin the exact limb arithmetic, recoding, tables and pipeline logic that the HIP kernels run."""
import hashlib
import json
import random
from pathlib import Path

import numpy as np
import pytest

from oracle import plume_oracle as O
from tests import _devsim as D
from tests import _oracle_c as OC
from tests import synth

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).parent / "golden" / "golden_batches.json").read_text())
P, N = O.P, O.N
PC = 2**32 + 977
EDGE_FE = [0, 1, 2, PC - 1, PC, PC + 1, P - 2, P - 1, P, P + 1, P + PC - 1, 2**256 - PC, 2**256 - PC + 1, 2**256 - 2, 2**256 - 1,
           2**255, 2**128, 2**128 - 1, (1 << 224) - 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, int("f" * 56 + "0" * 8, 16)]


def fe_cases(rng, nrand=300):
    vals = list(EDGE_FE) + [rng.randrange(2**256) for _ in range(nrand)] + [rng.randrange(P) for _ in range(nrand)]
    a, b = [], []
    for x in EDGE_FE:
        for y in EDGE_FE:
            a.append(x); b.append(y)
    for _ in range(nrand * 2):
        a.append(rng.choice(vals)); b.append(rng.choice(vals))
    return a, b


def test_fe_arith():
    rng = random.Random(1)
    a, b = fe_cases(rng)
    for op, fn in [(0, lambda x, y: x * y), (2, lambda x, y: x + y), (3, lambda x, y: x - y)]:
        got = D.fe_op(op, a, b)
        for x, y, g in zip(a, b, got):
            assert g < 2**256 and g % P == fn(x, y) % P, (op, hex(x), hex(y), hex(g))
    got = D.fe_op(1, a)
    assert all(g % P == x * x % P for x, g in zip(a, got))
    got = D.fe_op(4, a)
    assert all(g % P == (-x) % P for x, g in zip(a, got))
    got = D.fe_op(7, a)
    assert all(g == x % P for x, g in zip(a, got))
    ks = [1, 2, 3, 4, 8, 11, 1771, 65535, 2**20 - 1]
    for k in ks:
        got = D.fe_op(8, a, [k] * len(a))
        assert all(g % P == x * k % P for x, g in zip(a, got)), k
    assert [g for g in D.fe_op(9, a)] == [int(x % P == 0) for x in a]
    assert [g for g in D.fe_op(10, a, b)] == [int((x - y) % P == 0) for x, y in zip(a, b)]
    assert [g for g in D.fe_op(11, a)] == [(x % P) & 1 for x in a]
    # round 3: products with a subtraction riding in their fold (plume_fe_mul.inc fe_mul_sub / fe_sqr_sub / fe_sqr_sub2) and the doubling's two-product Y'
    for op, fn in [(16, lambda x, y: x * y - x), (17, lambda x, y: x * x - 2 * y - x), (18, lambda x, y: x * x - 2 * y), (19, lambda x, y: x * (y - x) - 2 * y * y),
                   (20, lambda x, y: (x - y) * (x + y) - y)]:
        got = D.fe_op(op, a, b)
        for x, y, g in zip(a, b, got):
            assert g == fn(x, y) % P, (op, hex(x), hex(y), hex(g))


def test_products_at_their_operand_bounds_come_back_tight():
    """Round 4: the product's top columns run first (gen_fe_mul.py), so the excess over 2^256 is folded into limbs 0..2 before they are formed and limb 8 takes a
    remainder at the end.  Drive the column sums to the contract's bounds (plume_field.h: tight = limbs <= 2^29 + 2^19, limb 8 <= 2^24 + 2^10; multiplication inputs up
    to 9 * max(a) * max(b) < 2^64 - 2^50 with limb 8 <= 2^26) and check value and tightness of every form; the host build asserts the intermediate bounds."""
    val = lambda L: sum(x << (29 * i) for i, x in enumerate(L))
    tight = lambda L: all(x <= 2**29 + 2**19 for x in L[:8]) and L[8] <= 2**24 + 2**10
    T = 2**29 + 2**19
    rng = random.Random(44)
    big = int(((2**64 - 2**50) // 9) ** 0.5) - 1                       # both operands at the square root of the bound (~1.33 * 2^30)
    shapes = [([T] * 8 + [2**24 + 2**10], [T] * 8 + [2**24 + 2**10]), ([big] * 8 + [2**26], [big] * 8 + [2**26]), ([2**31 - 1] * 8 + [2**26], [T] * 8 + [2**26]),
              ([0] * 9, [big] * 8 + [2**26]), ([big] * 8 + [0], [0] * 8 + [2**26])]
    for _ in range(200):
        m = rng.choice([T, big, 2**29 - 1])
        shapes.append(([rng.randrange(m + 1) for _ in range(8)] + [rng.randrange(2**26 + 1)], [rng.randrange(m + 1) for _ in range(8)] + [rng.randrange(2**26 + 1)]))
    a, b = [s[0] for s in shapes], [s[1] for s in shapes]
    for got, x, y in zip(D.fe_raw(0, a, b), a, b):
        assert tight(got) and val(got) % P == val(x) * val(y) % P
    sq = [x for x in a if max(x[:8]) <= big]
    for got, x in zip(D.fe_raw(1, sq), sq):
        assert tight(got) and val(got) % P == val(x) ** 2 % P
    # tight operands for the scaled squarings; subtrahends up to 2p / 4p limbwise for the fused forms
    ta = [x for x in a if tight(x)] + [[rng.randrange(T + 1) for _ in range(8)] + [rng.randrange(2**24 + 2**10 + 1)] for _ in range(100)]
    for op, f in [(6, 3), (7, 2)]:
        for got, x in zip(D.fe_raw(op, ta), ta):
            assert tight(got) and val(got) % P == f * val(x) ** 2 % P
    p_l = [0x1FFFFC2F, 0x1FFFFFF7] + [0x1FFFFFFF] * 6 + [0x00FFFFFF]
    sub2 = [[rng.choice([0, 2 * q, rng.randrange(2 * q + 1)]) for q in p_l] for _ in a]
    sub4 = [[rng.choice([0, 4 * q, rng.randrange(4 * q + 1)]) for q in p_l] for _ in a]
    for got, x, y, s in zip(D.fe_raw(3, a, b, sub2), a, b, sub2):
        assert tight(got) and val(got) % P == (val(x) * val(y) - val(s)) % P
    for got, x, s in zip(D.fe_raw(4, sq, None, sub4[:len(sq)]), sq, sub4):
        assert tight(got) and val(got) % P == (val(x) ** 2 - val(s)) % P
    for got, x, s in zip(D.fe_raw(5, sq, None, sub2[:len(sq)]), sq, sub2):
        assert tight(got) and val(got) % P == (val(x) ** 2 - 2 * val(s)) % P
    # two products, one fold: the doubling's Y' at its bound (E * (D - X') + (-B) * 2B: 2^29 * 3 * 2^29 + 2^30 * 2^30 per column term)
    e1, e2, e3, e4 = [T] * 8 + [2**24], [3 * 2**29 + 2**19] * 8 + [2**26], [2**30 - 2000] * 8 + [2**25], [2**30 + 2**20] * 8 + [2**25]
    qa = [e1] + [[rng.randrange(v + 1) for v in e1] for _ in range(100)]
    qb = [e2] + [[rng.randrange(v + 1) for v in e2] for _ in range(100)]
    qc = [e3] + [[rng.randrange(v + 1) for v in e3] for _ in range(100)]
    qe = [e4] + [[rng.randrange(v + 1) for v in e4] for _ in range(100)]
    for got, x, y, z, w in zip(D.fe_raw(2, qa, qb, qc, qe), qa, qb, qc, qe):
        assert tight(got) and val(got) % P == (val(x) * val(y) + val(z) * val(w)) % P


def test_group_law_compositions_keep_every_product_operand_in_bounds():
    """the product's limb-8 precondition (<= 2^26, since round 4's top-columns-first fold) is asserted in host builds only -- on whatever values a test happens to
    carry.  Here the group law's lazy sums and differences meet operands at the ENDS of the tight range (every limb 0 or its maximum, 2^29 + 2^19 and 2^24 + 2^10 on top, in
    every combination that matters plus random ones): each fe_mul / fe_sqr / fe_muladd inside jac_dbl, jac_dbl_neg, jac_madd (row y negated or not) and jac_add is checked by
    fe_mul_inputs_ok (the process aborts on a violation), and what comes back must be tight again, so that the operations compose."""
    T, T8 = 2**29 + 2**19, 2**24 + 2**10
    rng = random.Random(99)
    tight = lambda L: all(x <= T for x in L[:8]) and L[8] <= T8
    hi, lo = [T] * 8 + [T8], [0] * 9

    def pick():
        r = rng.random()
        if r < 0.3:
            return list(hi)
        if r < 0.45:
            return list(lo)
        if r < 0.8:
            return [rng.choice([0, T]) for _ in range(8)] + [rng.choice([0, T8])]
        return [rng.randrange(T + 1) for _ in range(8)] + [rng.randrange(T8 + 1)]
    n = 600
    ops = [[pick() for _ in range(n)] for _ in range(5)]
    for k in range(5):                                   # the all-maximum and the all-but-one-maximum combinations first
        for j in range(5):
            ops[j][k] = list(hi if j != k else lo)
    ops[0][5], ops[1][5], ops[2][5], ops[3][5], ops[4][5] = list(hi), list(hi), list(hi), list(hi), list(hi)
    for op in range(6):
        for row in D.group_raw(op, *ops):
            assert tight(row[0:9]) and tight(row[9:18]) and tight(row[18:27]), op
    # ... and composed: the outputs of one operation are the inputs of the next, twelve deep, every order
    cur = [ops[0], ops[1], ops[2]]
    for step in range(12):
        op = rng.choice([0, 1, 2, 4, 5])
        out = D.group_raw(op, cur[0], cur[1], cur[2], ops[3], ops[4])
        cur = [[r[0:9] for r in out], [r[9:18] for r in out], [r[18:27] for r in out]]
        assert all(tight(a) and tight(b) and tight(c) for a, b, c in zip(*cur)), (step, op)


def test_fe_inv_pow():
    rng = random.Random(2)
    a = [1, 2, P - 1, P + 1, 2**256 - 1, PC] + [rng.randrange(1, 2**256) for _ in range(40)]
    a = [x for x in a if x % P]
    assert [g % P for g in D.fe_op(5, a)] == [pow(x, P - 2, P) for x in a]
    assert [g % P for g in D.fe_op(6, a)] == [pow(x, (P - 3) // 4, P) for x in a]
    assert D.fe_op(5, [0, P])[0] % P == 0


def test_sc_arith():
    rng = random.Random(3)
    edge = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, 2**128, 2**255 % N, 2**256 - N]
    a = edge * len(edge) + [rng.randrange(N) for _ in range(300)]
    b = [y for y in edge for _ in edge] + [rng.randrange(N) for _ in range(300)]
    assert D.sc_op(0, a, b) == [x * y % N for x, y in zip(a, b)]
    assert D.sc_op(1, a, b) == [(x + y) % N for x, y in zip(a, b)]
    assert D.sc_op(2, a) == [(-x) % N for x in a]
    wide_lo = [rng.randrange(2**256) for _ in range(200)] + [2**256 - 1, 0, 2**256 - 1]
    wide_hi = [rng.randrange(2**256) for _ in range(200)] + [2**256 - 1, 2**256 - 1, 0]
    assert D.sc_op(3, wide_lo, wide_hi) == [(lo + (hi << 256)) % N for lo, hi in zip(wide_lo, wide_hi)]


def test_glv_and_booth():
    rng = random.Random(4)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    ks = [0, 1, 2, N - 1, N - 2, lam, lam + 1, N - lam, (N - 1) // 2, (N + 1) // 2, 2**128, 2**128 - 1, 2**255 % N] + [rng.randrange(N) for _ in range(2000)]
    # round 4: the split is exact integer arithmetic on 29-bit limbs (plume_ec.h glv_split): k1 = k - c1 a1 - c2 a2, k2 = c1 (-b1) - c2 b2 with c_i = round(k g_i / 2^384)
    g1 = 0x3086D221A7D46BCDE86C90E49284EB153DAA8A1471E8CA7FE893209A45DBB031
    g2 = 0xE4437ED6010E88286F547FA90ABFE4C4221208AC9DF506C61571B4AE8AC47F71
    a1, mb1, a2 = 0x3086D221A7D46BCDE86C90E49284EB15, 0xE4437ED6010E88286F547FA90ABFE4C3, 0x114CA50F7A8E2F3F657C1108D9D44CFD8
    ks += [N - lam, 2**255, 2**256 % N, (1 << 174) - 1, 1 << 174, (1 << 174) + 1, (1 << 145) - 1, g1 % N, g2 % N] + [rng.randrange(N) for _ in range(6000)]
    nonzero = total = 0
    for k, (m1, n1, m2, n2, codes) in zip(ks, D.glv(ks)):
        assert m1 < 2**128 and m2 < 2**128
        k1 = -m1 if n1 else m1
        k2 = -m2 if n2 else m2
        assert (k1 + k2 * lam) % N == k
        c1, c2 = (k * g1 + (1 << 383)) >> 384, (k * g2 + (1 << 383)) >> 384
        assert (k1, k2) == (k - c1 * a1 - c2 * a2, c1 * mb1 - c2 * a1), hex(k)
        # round 5: ONE Eisenstein digit per position (plume_ec.h eisd_store): k1 + k2 w = sum d_i 4^i, d_i in {0, units, associates of 1 - w, associates of 2}
        assert len(codes) == D.NPOS and all(0 <= c <= 18 for c in codes)
        va = vb = 0
        for c in reversed(codes):
            da, db = D.eis_digit(c)
            va, vb = 4 * va + da, 4 * vb + db
        assert (va, vb) == (k1, k2), hex(k)
        nonzero += sum(1 for c in codes[:64] if c)
        total += 64
    assert 0.92 < nonzero / total < 0.95                       # fifteen of the sixteen residues mod 4 are non-zero digits


def test_eisenstein_digit_table():
    """the 64-entry table behind eisd_entry (plume_ec.h), regenerated from its definition: for t = (ta, tb), |t| <= 4, the digit is the representative of t mod 4 among
    0, the units, the associates of theta = 1 - w and of 2 that leaves a carry (t - d) / 4 in {-1, 0, 1}^2; and the header carries exactly this table"""
    import re
    digits = {c: D.eis_digit(c) for c in range(1, 19)}
    assert len(set(digits.values())) == 18 and all(max(abs(a), abs(b)) <= 2 for a, b in digits.values())
    assert {(a % 4, b % 4) for a, b in digits.values()} | {(0, 0)} == {(a, b) for a in range(4) for b in range(4)}      # a complete residue system of Z[w] / 4
    tab = []
    for idx in range(64):
        ca, cb, na, nb = (idx >> 2) & 3, idx & 3, (idx >> 4) & 1, (idx >> 5) & 1
        ta_s = [t for t in range(-4, 5) if t % 4 == ca and (t < 0) == bool(na)]
        tb_s = [t for t in range(-4, 5) if t % 4 == cb and (t < 0) == bool(nb)]
        if not ta_s or not tb_s:
            tab.append(0)
            continue
        if (ca, cb) == (0, 0):
            tab.append(0 | (2 << 5) | (2 << 8))
            continue
        best = None
        for code, d in digits.items():
            if (d[0] % 4, d[1] % 4) != (ca, cb):
                continue
            worst = max([abs((t - d[0]) // 4) for t in ta_s] + [abs((t - d[1]) // 4) for t in tb_s])
            if best is None or worst < best[0]:
                best = (worst, code, d)
        assert best and best[0] <= 1, idx
        tab.append(best[1] | ((best[2][0] + 2) << 5) | ((best[2][1] + 2) << 8))
    src = (Path(__file__).parent.parent / "zk-nullifier-sig_amd" / "csrc" / "plume_ec.h").read_text()
    body = src[src.index("static const uint16_t T[64] = {"):]
    body = body[:body.index("};")]
    assert [int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]{3}", body)] == tab
    # ... and the register-resident packing the kernels run (eisd_entry) is this table for every t the recoding can meet
    import ctypes as C
    out = np.zeros(162, dtype=np.uint32)
    D.lib().ds_eisd_entries(out.ctypes.data_as(C.POINTER(C.c_uint32)))
    k = 0
    for ta in range(-4, 5):
        for tb in range(-4, 5):
            idx = ((ta & 3) << 2) | (tb & 3) | (16 if ta < 0 else 0) | (32 if tb < 0 else 0)
            assert int(out[k]) == tab[idx] and int(out[k + 1]) == tab[idx], (ta, tb)
            k += 2


def test_sha256_generic():
    for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 121, 198, 99, 500):
        d = bytes((i * 13 + n) & 0xFF for i in range(n))
        assert D.sha256(d) == hashlib.sha256(d).digest(), n


def test_point_mul_matches_oracle():
    rng = random.Random(5)
    g = O.pt_bytes(O.G)
    pts = [g, O.pt_bytes(O.pt_mul(rng.randrange(N), O.G)), O.pt_bytes(O.pt_mul(7, O.G))]
    ks = [0, 1, 2, 3, 7, 8, 9, 15, 16, 17, N - 1, N - 2, N, 2**128, 2**128 - 1, 2**255, 2**256 - 1] + [rng.randrange(N) for _ in range(12)]
    for p in pts:
        for k in ks:
            kb = k.to_bytes(32, "big")
            assert D.point_mul(kb, p) == OC.point_mul(kb, p), (k, p.hex())
    assert D.point_mul((5).to_bytes(32, "big"), bytes(64)) == bytes(64)
    bad = bytearray(g); bad[63] ^= 1
    assert D.point_mul((5).to_bytes(32, "big"), bytes(bad)) is None


def test_h2c_kats(kats):
    v = kats["plume_vector"]
    msgs = [v["msg_utf8"].encode(), b"", b"x" * 300]
    mb, off = OC.pack_msgs(msgs)
    pk = np.frombuffer(bytes.fromhex(v["pk_x"] + v["pk_y"]) * 3, dtype=np.uint8).reshape(3, 64).copy()
    h = D.h2c_batch(mb, off, pk)
    assert h[0].tobytes().hex() == v["h_x"] + v["h_y"]
    assert np.array_equal(h, OC.hash_to_curve_batch(mb, off, pk))
    # raw-bytes mode: RFC 9380 J.8.1 "", "abc", and the literal 62-byte preimage of the TS test
    raw = [b"", b"abc", bytes.fromhex(kats["h2c_preimage"]["preimage_hex"])]
    mb, off = OC.pack_msgs(raw)
    h = D.h2c_batch(mb, off, None)
    assert h[0].tobytes().hex() == kats["rfc9380_empty"]["p_x"] + kats["rfc9380_empty"]["p_y"]
    assert h[1].tobytes().hex() == kats["h2c_abc"]["x"] + kats["h2c_abc"]["y"]
    assert h[2].tobytes().hex() == kats["h2c_preimage"]["x"] + kats["h2c_preimage"]["y"]
    # identity pk: encoding is the single byte 00
    mb, off = OC.pack_msgs([b"edge"])
    z = np.zeros((1, 64), dtype=np.uint8)
    assert D.h2c_batch(mb, off, z)[0].tobytes() == O.pt_bytes(O.hash_to_curve(b"edge", None))


@pytest.mark.parametrize("ver", [1, 2])
@pytest.mark.parametrize("L", [1, 3, 7])
def test_golden_verify(ver, L):
    items = GOLD[f"verify_v{ver}"][: (256 if L == 3 else 48)]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    ok = D.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                        OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None, L=L)
    assert list(ok) == [it["ok"] for it in items]


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_edge(ver):
    items = [e for e in GOLD["edge"] if e["version"] == ver]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    ok = D.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                        OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None)
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_sign(ver, kats):
    items = GOLD[f"sign_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    o = D.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32))
    for key, w in [("pk", 64), ("h", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]:
        assert np.array_equal(o[key], OC.arr(items, key, w)), key
    assert list(o["status"]) == [it["status"] for it in items]
    # arkworks-shaped (pk supplied) gives identical outputs (BASELINE config 5)
    o2 = D.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32), pk_in=OC.arr(items, "pk", 64))
    for k in o:
        assert np.array_equal(o[k], o2[k]), k
    # the reference's fixed vector (BASELINE config 1)
    v = kats["plume_vector"]
    mb, off = OC.pack_msgs([v["msg_utf8"].encode()])
    o = D.sign_batch(ver, mb, off, np.frombuffer(bytes.fromhex(v["sk"]), dtype=np.uint8).reshape(1, 32).copy(),
                     np.frombuffer(bytes.fromhex(v["r"]), dtype=np.uint8).reshape(1, 32).copy())
    assert o["c"][0].tobytes().hex() == v[f"c_v{ver}"] and o["s"][0].tobytes().hex() == v[f"s_v{ver}"]
    assert o["nullifier"][0].tobytes().hex() == v["nullifier_x"] + v["nullifier_y"]


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("ver", [1, 2])
def test_uniform_schedule_signer_gives_the_same_bytes(ver, level):
    """plume_set_sign_uniform (round 4): the signer's chains with no branch on a digit -- every slot adds, zero digits are dropped by a masked select, the accumulator starts
    at an offset point -- and, at level 2, with no table address derived from a digit (all eight rows of a window read, one kept by masked selects; the multiplications by G
    on the small scanned table instead of the comb) produce exactly the default signer's outputs: the golden batch, scalars full of zero digits (0x...0001, 2^k, n - 1: long runs of dummy additions),
    tiny keys (the crafted collisions of the default path), out-of-range scalars (status bits) and a supplied pk"""
    items = GOLD[f"sign_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    sk, r = OC.arr(items, "sk", 32), OC.arr(items, "r", 32)
    specials = [1, 2, 3, 7, 8, 9, 15, 16, 17, 2**16, 2**32 - 1, 2**64, 2**127, 2**128 - 1, 2**128, 2**128 + 1, 2**192, 2**255, N - 1, N - 2, N // 2, N // 3, (N + 1) // 2,
                0x1111111111111111111111111111111111111111111111111111111111111111 % N, 0x8888888888888888888888888888888888888888888888888888888888888888 % N, 0, N, N + 1, 2**256 - 1]
    sk, r = sk.copy(), r.copy()
    for k in range(len(items)):
        if k % 3 == 0:
            sk[k] = np.frombuffer(specials[::-1][(k // 3) % len(specials)].to_bytes(32, "big"), dtype=np.uint8)
        if k % 5 == 0:
            r[k] = np.frombuffer(specials[(k // 5 + 11) % len(specials)].to_bytes(32, "big"), dtype=np.uint8)
        if k % 7 == 0:
            r[k] = sk[k]
    want = D.sign_batch(ver, mb, off, sk, r)
    got = D.sign_batch(ver, mb, off, sk, r, uniform=level)
    for key in want:
        assert np.array_equal(got[key], want[key]), key
    assert int((want["status"] != 0).sum()) > 0
    got2 = D.sign_batch(ver, mb, off, sk, r, pk_in=want["pk"], uniform=level)
    for key in want:
        assert np.array_equal(got2[key], want[key]), key



_LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def _table_rows(pt):
    """the three rows of a window table (csrc/plume_ec.h, round 5): P, theta P = P - lambda P, 2P"""
    lp = O.pt_mul(_LAMBDA, pt)
    return [O.pt_bytes(pt), O.pt_bytes(O.pt_add(pt, O.pt_neg(lp))), O.pt_bytes(O.pt_mul(2, pt))]


def test_small_batch_table_path_gives_the_same_tables_and_results():
    """(Round 4 had two table builders, an affine chain and a Jacobian one for small batches, and this test held them to each other.  Round 5 has ONE: rows P, theta P,
    2P by a single round of inversions, plume_ec.h.)  The rows are the right POINTS -- tables of honest bases next to a record that is no curve point --, and the pipeline on
    top gives the oracle's bytes: the golden signs (affine pk bases and the Jacobian H), goldens + edge cases + a fuzzed batch through verify."""
    from tests import _fuzz
    rng = random.Random(5)
    pts = [O.pt_mul(rng.randrange(1, N), O.G) for _ in range(11)]
    bases = np.zeros((len(pts) + 1, 64), dtype=np.uint8)
    for j, pt in enumerate(pts):
        bases[j] = np.frombuffer(pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big"), dtype=np.uint8)
    bases[len(pts), 31] = 5                                               # (5, 0): no curve point, and y = 0 makes a denominator vanish: the guard must keep the neighbours' rows intact
    D.set_tables_small(True)                                              # (a no-op since round 5)
    try:
        got = D.tables_raw(bases)
        assert got.shape[1] == 3
        for j, pt in enumerate(pts):
            assert [got[j, k].tobytes() for k in range(3)] == _table_rows(pt), j
        for ver in (1, 2):
            items = GOLD[f"sign_v{ver}"]
            mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
            o = D.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32))
            for key, w in [("pk", 64), ("h", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]:
                assert np.array_equal(o[key], OC.arr(items, key, w)), key
            for items in (GOLD[f"verify_v{ver}"], [e for e in GOLD["edge"] if e["version"] == ver]):
                mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
                rp = OC.arr(items, "r_point", 64) if ver == 1 else None
                hr = OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None
                got = D.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32), rp, hr)
                assert [int(x) for x in got] == [it["ok"] for it in items]
            n = 256
            b = synth.sign_inputs(n, start=777 + ver)
            signed = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
            v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=17 + ver)
            args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
            assert np.array_equal(D.verify_batch(*args), OC.verify_batch(*args, nthreads=8))
    finally:
        D.set_tables_small(False)


def test_two_role_ingest_gives_the_same_verdicts():
    """round 4: below 2^17 items the ingest stage runs with two lanes per item (role A: pk, message, XMD, map(u0), the sum and the isogeny; role B: nullifier, scalars, window
    digits, map(u1)).  The host harness runs that form of the stage over the goldens, the edge cases (identity / off-curve / out-of-range inputs, ragged messages), the
    verify_non_zk goldens and fuzzed batches: every verdict as the oracle's."""
    from tests import _fuzz
    from tests.test_oracle_c import non_zk_args
    D.set_ingest_two_roles(True)
    try:
        for ver in (1, 2):
            for items in (GOLD[f"verify_v{ver}"], [e for e in GOLD["edge"] if e["version"] == ver]):
                mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
                rp = OC.arr(items, "r_point", 64) if ver == 1 else None
                hr = OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None
                got = D.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32), rp, hr)
                assert [int(o) for o in got] == [it["ok"] for it in items]
            n = 384
            b = synth.sign_inputs(n, start=555 + ver)
            signed = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
            v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=91 + ver)
            args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
            assert np.array_equal(D.verify_batch(*args), OC.verify_batch(*args, nthreads=8))
            nz = [it for it in NONZK if it["version"] == ver]
            got = D.verify_non_zk_batch(ver, *non_zk_args(nz))
            assert [int(o) for o in got] == [it["ok"] for it in nz]
    finally:
        D.set_ingest_two_roles(False)


def test_paired_half_chains_give_the_same_verdicts():
    """round 5: calls of at most 2^14 items run every long-form chain of the verifier as two halves -- the first joint slot on one lane, the second on another -- joined by one
    checked addition (k_verify_msm_pair).  The host harness runs that form over the goldens, the edge cases, the verify_non_zk goldens, fuzzed batches and the crafted items
    whose unchecked chains meet p == +-q (they are filed by the join and redone whole): every verdict as the oracle's; the same with the two-role ingest beside it and with
    equation 1's short form asked for (the pair form then stays out of the way)."""
    from tests import _fuzz
    from tests.test_oracle_c import non_zk_args
    D.set_msm_pair(True)
    try:
        for two_roles, eq1 in ((False, 0), (True, 0), (False, 1)):
            D.set_ingest_two_roles(two_roles)
            D.set_eq1_short(eq1)
            for ver in (1, 2):
                for items in (GOLD[f"verify_v{ver}"], [e for e in GOLD["edge"] if e["version"] == ver]):
                    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
                    rp = OC.arr(items, "r_point", 64) if ver == 1 else None
                    hr = OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None
                    got = D.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32), rp, hr)
                    assert [int(o) for o in got] == [it["ok"] for it in items]
                n = 256
                b = synth.sign_inputs(n, start=7777 + ver)
                signed = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
                v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=191 + ver)
                args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
                assert np.array_equal(D.verify_batch(*args), OC.verify_batch(*args, nthreads=8))
                nz = [it for it in NONZK if it["version"] == ver]
                got = D.verify_non_zk_batch(ver, *non_zk_args(nz))
                assert [int(o) for o in got] == [it["ok"] for it in nz]
    finally:
        D.set_msm_pair(False)
        D.set_ingest_two_roles(False)
        D.set_eq1_short(1)


def test_sign_edge_status():
    """out-of-range scalars and ragged messages: device sign path == C oracle, including status bits"""
    rng = random.Random(9)
    sks = [0, N, N + 5, 1, N - 1, 2**256 - 1, rng.randrange(1, N), rng.randrange(1, N)]
    rs = [rng.randrange(1, N), rng.randrange(1, N), 0, N - 1, 1, 7, N, rng.randrange(1, N)]
    msgs = [bytes(rng.randrange(256) for _ in range(l)) for l in (0, 1, 32, 55, 56, 64, 100, 257)]
    mb, off = OC.pack_msgs(msgs)
    sk = np.frombuffer(b"".join(x.to_bytes(32, "big") for x in sks), dtype=np.uint8).reshape(-1, 32).copy()
    r = np.frombuffer(b"".join(x.to_bytes(32, "big") for x in rs), dtype=np.uint8).reshape(-1, 32).copy()
    for ver in (1, 2):
        a = D.sign_batch(ver, mb, off, sk, r)
        b = OC.sign_batch(ver, mb, off, sk, r)
        for k in a:
            assert np.array_equal(a[k], b[k]), (ver, k)


def test_wide_generator_digits_special_scalars():
    """fixed-base comb (11-bit Booth digits over the whole 256-bit scalar, magnitudes up to 1024; the 8-bit patterns of the previous
    window size are kept): pk = sk*G and R = r*G for scalars that hit the digit extremes, against the C oracle"""
    rng = random.Random(21)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    ks = [1, 2, 127, 128, 129, 255, 256, 257, 0x8080, 0x7F7F, 0x80808080, 2**64 - 1, 2**127, 2**127 - 1, 2**127 + 1, 2**128 - 1, 2**128,
          lam, lam - 1, (128 * lam) % N, (N - 128) % N, N - 1, N - 2, (0x80 << 120), int("80" * 16, 16), int("7f" * 16, 16), int("ff" * 16, 16)]
    w11 = [int("10000000000" * 23, 2), int("01111111111" * 23, 2), int("11111111111" * 23, 2), int("10000000001" * 23, 2), (1 << 253) + 1024, (1 << 255) + (1 << 252)]
    ks += [1023, 1024, 1025, 2047, 2048, 2049, N - 1024, N - 1025] + [k % N for k in w11]
    ks += [rng.randrange(1, N) for _ in range(37)]
    n = len(ks)
    mb, off = OC.pack_msgs([b"w"] * n)
    sk = np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8).reshape(n, 32).copy()
    r = np.frombuffer(b"".join(k.to_bytes(32, "big") for k in reversed(ks)), dtype=np.uint8).reshape(n, 32).copy()
    a = D.sign_batch(2, mb, off, sk, r)
    b = OC.sign_batch(2, mb, off, sk, r, nthreads=8)
    for key in a:
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("ver", [1, 2])
def test_sec1_compressed_ingest(ver):
    """'next' row f-1: 33-byte SEC1 records, decompressed + validated by the device code"""
    from tests import _sec1
    items = GOLD[f"verify_v{ver}"][:64] + [e for e in GOLD["edge"] if e["version"] == ver and "off curve" not in e["note"] and "non-canonical" not in e["note"] and "= p" not in e["note"]]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    c33 = lambda k: _sec1.compress(OC.arr(items, k, 64))  # noqa: E731
    ok = D.verify_batch_sec1(ver, mb, off, c33("pk"), c33("nullifier"), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                             c33("r_point") if ver == 1 else None, c33("hashed_to_curve_r") if ver == 1 else None)
    assert list(ok) == [it["ok"] for it in items]
    cases = _sec1.malformed_cases(ver, GOLD[f"verify_v{ver}"])
    mb, off = OC.pack_msgs([c[0] for c in cases])
    col = lambda j, w: np.frombuffer(b"".join(c[j] for c in cases), dtype=np.uint8).reshape(-1, w).copy()  # noqa: E731
    ok = D.verify_batch_sec1(ver, mb, off, col(1, 33), col(2, 33), col(3, 32), col(4, 32), col(5, 33) if ver == 1 else None, col(6, 33) if ver == 1 else None)
    want = [int(_sec1.oracle_verify_sec1(ver, *c[:7])) for c in cases]
    bad = [(c[7], int(o), w) for c, o, w in zip(cases, ok, want) if int(o) != w]
    assert not bad, bad
    assert sum(want) >= 2   # the honest mutations still verify


def test_verify_equation1_wide_digits_special_scalars():
    """s*G - c*pk through the verify multi-scalar body for scalars that hit the wide-digit extremes of the generator's slots
    (12-bit Booth digits: +-2048, the carry into the top digit, the sign / high-nibble byte; and the 8-bit patterns of the previous
    window size), and for pk = +-G where the two bases coincide"""
    rng = random.Random(33)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    w12 = [int("800" * 10 + "80", 16), int("7ff" * 10 + "7f", 16), int("fff" * 10 + "ff", 16), int("801" * 10 + "80", 16), int("001" * 10 + "00", 16)]
    specials = [1, 2, 127, 128, 129, 255, 256, 0x8080, 0x80808080, 2**127, 2**127 - 1, 2**128 - 1, 2**128, lam, (128 * lam) % N, N - 128, N - 1,
                int("80" * 16, 16), int("7f" * 16, 16), int("ff" * 16, 16), (int("80" * 16, 16) * lam + int("80" * 16, 16)) % N,
                2047, 2048, 2049, 4095, 4096, 4097, 0x800800, 0x7FF800, N - 2048, (2048 * lam) % N, (2048 * lam + 2048) % N] + w12 + [(w12[0] * lam + w12[1]) % N]
    pks = [O.G, O.pt_neg(O.G), O.pt_mul(rng.randrange(1, N), O.G)]
    for pk in pks:
        for s_ in specials + [rng.randrange(1, N) for _ in range(4)]:
            for c_ in (1, s_, rng.randrange(1, N)):
                want = O.pt_add(O.pt_mul(s_, O.G), O.pt_neg(O.pt_mul(c_, pk)))
                got = D.eq1(s_.to_bytes(32, "big"), c_.to_bytes(32, "big"), O.pt_bytes(pk))
                assert got == O.pt_bytes(want), (hex(s_), hex(c_))


@pytest.mark.parametrize("ver", [1, 2])
def test_fuzzed_verify_batch_vs_oracle(ver):
    """device code on the host: 384 honest-then-mutated items, ok[] == C oracle"""
    from tests import _fuzz, synth
    n = 384
    b = synth.sign_inputs(n, start=700000)
    rng = random.Random(77 + ver)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 33, 55, 56, 64, 100]))) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    signed = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    v = _fuzz.fuzz_verify_batch(ver, signed, dict(msgs=mb, off=off), seed=5 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    got = D.verify_batch(*args)
    want = OC.verify_batch(*args, nthreads=8)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert 0.2 * n < int(got.sum()) < 0.8 * n


NONZK = json.loads((Path(__file__).parent / "golden" / "golden_non_zk.json").read_text())["items"]


@pytest.mark.parametrize("ver", [1, 2])
def test_verify_non_zk_golden_and_fuzz(ver):
    """device code on the host, PLUME_MODE_NON_ZK (rust-arkworks/src/tests.rs:28-78): the Python oracle's vectors (incl. Err = 2, zero scalars),
    then 768 fuzzed items against the C oracle"""
    from tests import _fuzz
    from tests.test_oracle_c import non_zk_args
    items = [it for it in NONZK if it["version"] == ver]
    ok = D.verify_non_zk_batch(ver, *non_zk_args(items))
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad
    n = 768
    b = synth.sign_inputs(n, start=830000)
    signed = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
    v = _fuzz.fuzz_non_zk_batch(ver, signed, b, seed=21 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    got = D.verify_non_zk_batch(*args)
    want = OC.verify_non_zk_batch(*args, nthreads=8)
    assert np.array_equal(got, want), [(int(i), int(got[i]), int(want[i])) for i in np.nonzero(got != want)[0][:10]]
    assert {int(x) for x in got} == {0, 1, 2} and 0.15 * n < int((got == 1).sum()) < 0.8 * n


def test_malformed_message_offsets_reject_without_reading():
    """device-resident calls hand the kernels device-side offsets: decreasing offsets, offsets past msgs_bytes and spans over 4 GiB must reject
    the ITEM (ok = 0) and never dereference msgs (here: msgs is a 64-byte buffer, any such read would be far out of bounds)"""
    n = 8
    b = synth.sign_inputs(n, start=820000)
    signed = OC.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"], nthreads=4)
    args = lambda off, nbytes: D.verify_batch_bounded(1, b["msgs"], off, nbytes, signed["pk"], signed["nullifier"], signed["c"], signed["s"],  # noqa: E731
                                                      signed["r_point"], signed["hashed_to_curve_r"])
    assert list(args(b["off"], 32 * n)) == [1] * n
    off = b["off"].copy(); off[3] = 2**40                  # item 2 reaches far past the buffer, item 3 starts there and "ends" before it starts
    assert list(args(off, 32 * n)) == [1, 1, 0, 0, 1, 1, 1, 1]
    assert list(args(b["off"], 32 * 5 + 1)) == [1] * 5 + [0] * 3      # msgs_bytes cuts items 5.. off
    off = b["off"].copy(); off[8] = off[7] + 2**32 + 5     # a span the 32-bit SHA front end cannot take
    assert list(args(off, 2**33)) == [1] * 7 + [0]


@pytest.mark.parametrize("ver", [1, 2])
def test_small_secret_keys_hit_the_exceptional_additions(ver):
    """pk = +-G, +-2G, ... makes s*G - c*pk (and s*H - c*nullifier) run into p == +-q inside the multi-scalar chain.  The hot loop
    uses unchecked additions and redoes such a lane with the checked form (plume_ec.h jac_madd): results must still equal the
    oracle's, and the fallback must actually have been taken."""
    n = 96
    b = synth.sign_inputs(n, start=31000)
    small = [1, 2, 3, 4, 7, 8, 9, 16, 17, 128, 129, 255, 256, 257, O.N - 1, O.N - 2, O.N - 8, O.N - 16, O.N - 128, O.N - 256]
    for i in range(n):
        b["sk"][i] = np.frombuffer(small[i % len(small)].to_bytes(32, "big"), dtype=np.uint8)
    want = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
    before = D.fallback_count()
    got = D.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r", "status"):
        assert np.array_equal(np.asarray(got[k]).reshape(n, -1), np.asarray(want[k]).reshape(n, -1)), k
    v = synth.corrupt_for_verify(ver, b, want, start=31000)
    ok = D.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    want_ok = OC.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"), nthreads=8)
    assert np.array_equal(ok, want_ok)


def test_crafted_collisions_take_the_checked_fallback():
    """R' = s*G - c*pk with pk = G and s = +-c = d <= 8: the first window loads d*G and the pk slot then adds -+d*G, i.e. p == -q
    (result: identity) or p == q (result: 2d*G) in the very first addition.  Both must come out right, through the fallback."""
    N = O.N
    g = O.pt_bytes(O.G)
    before = D.fallback_count()
    for d in range(1, 9):
        assert D.eq1(d.to_bytes(32, "big"), d.to_bytes(32, "big"), g) == bytes(64), d                         # d*G - d*G
        assert D.eq1(d.to_bytes(32, "big"), (N - d).to_bytes(32, "big"), g) == O.pt_bytes(O.pt_mul(2 * d, O.G)), d   # d*G + d*G
    assert D.fallback_count() - before >= 12, "the crafted collisions did not go through the checked fallback"


def test_inversion_by_divsteps_matches_fermat_and_python():
    """fe_inv (Bernstein-Yang divsteps, 20 x 30 steps) against pow(x, -1, p), and against the Fermat chain it replaced; 0 and p give 0"""
    rng = random.Random(2024)
    vals = [1, 2, 3, P - 1, P - 2, P + 1, 2**255, 2**256 - 1, 977, 2**32 + 977, 2**128, 2**30, 2**30 - 1, 2**60, (P + 1) // 2]
    vals += [rng.getrandbits(256) for _ in range(3000)] + [rng.getrandbits(k) | 1 for k in (1, 8, 29, 30, 31, 58, 60, 61, 200, 255) for _ in range(20)]
    vals = [v for v in vals if v % P]
    got = D.fe_op(15, vals)
    assert got == [pow(v % P, -1, P) for v in vals]
    assert [g % P for g in D.fe_op(5, vals)] == got
    assert D.fe_op(15, [0, P]) == [0, 0]


def _check_h2c_intermediates(get, regs_from_be, kats):
    """shared by the host simulation and the GPU test: u0, u1, Q0, Q1, H against the RFC 9380 J.8.1 vector the reference holds
    (rust-arkworks/src/secp256k1/tests.rs:89-108) and the goldens' u0, u1, q0, q1; the E' points against the Python oracle's simplified SWU;
    the register form against circuits/circom/utils.ts:11-17 (value = sum r_i 2^(64 i))"""
    k = kats["rfc9380_empty"]
    mb, off = OC.pack_msgs([b"", b"abc"])
    o = get(mb, off, None, False)
    hx = lambda a: a.tobytes().hex()  # noqa: E731
    assert (hx(o["u"][0, 0]), hx(o["u"][0, 1])) == (k["u0"], k["u1"])
    assert (hx(o["q"][0, 0]), hx(o["q"][0, 1]), hx(o["q"][0, 2]), hx(o["q"][0, 3])) == (k["q0_x"], k["q0_y"], k["q1_x"], k["q1_y"])
    assert (hx(o["h"][0, 0]), hx(o["h"][0, 1])) == (k["p_x"], k["p_y"])
    assert hx(o["h"][1, 0]) + hx(o["h"][1, 1]) == kats["h2c_abc"]["x"] + kats["h2c_abc"]["y"]
    items = GOLD["sign_v1"][:32]
    msgs = [bytes.fromhex(it["msg"]) for it in items]
    mb, off = OC.pack_msgs(msgs)
    pk = OC.arr(items, "pk", 64)
    o = get(mb, off, pk, False)
    r = get(mb, off, pk, True)
    for i, it in enumerate(items):
        assert (hx(o["u"][i, 0]), hx(o["u"][i, 1])) == (it["u0"], it["u1"])
        assert hx(o["q"][i, :2]) == it["q0"] and hx(o["q"][i, 2:]) == it["q1"] and hx(o["h"][i]) == it["h"]
        for j, uu in enumerate((int(it["u0"], 16), int(it["u1"], 16))):
            x, y = O.map_to_curve_sswu(uu)                                # on E'
            assert (int(hx(o["mapped"][i, 2 * j]), 16), int(hx(o["mapped"][i, 2 * j + 1]), 16)) == (x, y)
            assert (y * y - x**3 - O.ISO_A * x - O.ISO_B) % P == 0
            assert O.pt_bytes(O.iso_map((x, y))).hex() == it["q0" if j == 0 else "q1"]
        for key in ("u", "mapped", "q", "h"):                             # registers: little-endian 64-bit limbs of the same values
            for j in range(o[key].shape[1]):
                val = int(hx(o[key][i, j]), 16)
                assert [int(x) for x in r[key][i, j]] == [(val >> (64 * t)) & (2**64 - 1) for t in range(4)]
    # c, s, pk, nullifier -> registers (scalarToCircuitValue / pointToCircuitValue)
    vals = np.concatenate([OC.arr(items, "c", 32), OC.arr(items, "s", 32), OC.arr(items, "nullifier", 64).reshape(-1, 32)])
    regs = regs_from_be(vals)
    for v, rg in zip(vals, regs.reshape(-1, 4)):
        assert sum(int(x) << (64 * t) for t, x in enumerate(rg)) == int(v.tobytes().hex(), 16)
    # invalid pk -> zeros
    bad = pk.copy(); bad[0, 63] ^= 1
    z = get(mb, off, bad, False)
    assert not z["u"][0].any() and not z["h"][0].any() and z["h"][1].any()


def test_h2c_intermediates_and_registers(kats):
    _check_h2c_intermediates(lambda mb, off, pk, regs: D.h2c_intermediates(mb, off, pk, regs), D.registers_from_be, kats)


def test_one_isogeny_after_adding_on_the_isogenous_curve():
    """hash_to_curve evaluates the 3-isogeny ONCE, on Q0' + Q1' added on E' (RFC 9380 section 6.6.3; plume_h2c.h map2_to_curve_jac): equal to iso(Q0') + iso(Q1') from the
    Python oracle for random pairs and for the pairs the chord formula cannot take -- u1 = u0 (a doubling) and u1 = -u0 (opposite points: the identity) -- which go through
    the two-isogeny fallback"""
    rng = random.Random(99)
    us = [rng.randrange(P) for _ in range(24)] + [0, 1, 2, P - 1]
    pairs = [(rng.choice(us), rng.choice(us)) for _ in range(40)] + [(u, u) for u in us[:6]] + [(u, (P - u) % P) for u in us[:6]]
    for u0, u1 in pairs:
        want = O.pt_add(O.iso_map(O.map_to_curve_sswu(u0)), O.iso_map(O.map_to_curve_sswu(u1)))
        assert D.map2_to_curve(u0, u1) == O.pt_bytes(want), (u0, u1)
    assert D.map2_to_curve(5, P - 5) == bytes(64)


def _check_h2c_hints(get_hints, get_inter):
    """the square-root hints (UNPINNED: no reference vector exists; include/plume_hip.h defines them).  Checked (a) against the definitions themselves -- algebra on the
    values the pinned outputs give: gx1, gx2 from u (RFC 9380 F.2), root^2 = gx or Z gx, evenness, y_pos = y_mapped with sgn0 = sgn0(u) -- and (b) against the Python
    oracle's independent big-integer restatement (O.sswu_hints), on RFC 9380 J.8.1's messages and ragged PLUME inputs"""
    rng = random.Random(77)
    items = GOLD["sign_v1"][:24]
    cases = [(OC.pack_msgs([b"", b"abc", b"abcdef0123456789", b"q128_" + b"q" * 128, b"a512_" + b"a" * 512]), None),
             (OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items]), OC.arr(items, "pk", 64)),
             (OC.pack_msgs([rng.randbytes(rng.randrange(0, 200)) for _ in range(40)]), None)]
    val = lambda a: int(a.tobytes().hex(), 16)  # noqa: E731
    branches = set()
    for (mb, off), pk in cases:
        hints, inter = get_hints(mb, off, pk, False), get_inter(mb, off, pk, False)
        regs = get_hints(mb, off, pk, True)
        for i in range(len(off) - 1):
            for k in range(2):
                u = val(inter["u"][i, k])
                g1s, g2s, ypos = (val(hints[f"q{k}_{nm}"][i]) for nm in ("gx1_sqrt", "gx2_sqrt", "y_pos"))
                assert (g1s, g2s, ypos) == O.sswu_hints(u)
                # the definitions, from u alone
                A, B, Zc = O.ISO_A, O.ISO_B, O.Z % P
                d = (Zc * Zc * pow(u, 4, P) + Zc * u * u) % P
                x1 = (-B) * pow(A, -1, P) % P * (1 + pow(d, -1, P)) % P
                x2 = Zc * u * u % P * x1 % P
                gx1, gx2 = (pow(x1, 3, P) + A * x1 + B) % P, (pow(x2, 3, P) + A * x2 + B) % P
                sq1 = pow(gx1, (P - 1) // 2, P) == 1
                branches.add(sq1)
                assert g1s % 2 == 0 and g2s % 2 == 0 and g1s < P and g2s < P
                assert g1s * g1s % P == (gx1 if sq1 else Zc * gx1 % P) and g2s * g2s % P == (Zc * gx2 % P if sq1 else gx2)
                assert ypos * ypos % P == (gx1 if sq1 else gx2) and ypos % 2 == u % 2
                assert ypos == val(inter["mapped"][i, 2 * k + 1]) and val(inter["mapped"][i, 2 * k]) == (x1 if sq1 else x2)
                for nm in ("gx1_sqrt", "gx2_sqrt", "y_pos"):                          # register form
                    assert sum(int(x) << (64 * t) for t, x in enumerate(regs[f"q{k}_{nm}"][i])) == val(hints[f"q{k}_{nm}"][i])
    assert branches == {True, False}


def test_h2c_hints_unpinned_definitions():
    names = ["q0_gx1_sqrt", "q0_gx2_sqrt", "q0_y_pos", "q1_gx1_sqrt", "q1_gx2_sqrt", "q1_y_pos"]

    def hints(mb, off, pk, regs):
        h = D.h2c_intermediates(mb, off, pk, regs)["hints"]
        return {nm: h[:, k] for k, nm in enumerate(names)}
    _check_h2c_hints(hints, lambda mb, off, pk, regs: D.h2c_intermediates(mb, off, pk, regs))


def _check_sec1_der(to_der, from_der, kats):
    """SecretKey::to_sec1_der / from_sec1_der as the wasm layer uses them (javascript/src/lib.rs:98-110,124-140), against the README's records: the whole
    109-byte secret-key record (javascript/README.md:26-32) and the first 100 bytes of the sample output's `s` (README :58-72, cut there by the listing)"""
    w = kats["wasm_readme"]
    sk_der, s_pre = bytes.fromhex(w["sk_sec1_der"]), bytes.fromhex(w["s_sec1_der_first_100_bytes"])
    rng = random.Random(4)
    ks = [int.from_bytes(sk_der[7:39], "big"), int.from_bytes(s_pre[7:39], "big"), 1, 2, N - 1] + [rng.randrange(1, N) for _ in range(27)] + [0, N, 2**256 - 1]
    arr = np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8).reshape(-1, 32)
    der, st = to_der(arr)
    assert der[0].tobytes() == sk_der and der[1].tobytes()[:100] == s_pre
    assert list(st) == [0] * 32 + [2, 2, 2] and not der[32:].any()
    for k, d in zip(ks[:32], der):
        x, y = O.pt_mul(k, O.G)
        assert d.tobytes() == bytes.fromhex("306b0201010420") + k.to_bytes(32, "big") + bytes.fromhex("a14403420004") + x.to_bytes(32, "big") + y.to_bytes(32, "big")
    if from_der is not None:
        bad = der.copy()
        bad[3, 0] ^= 1; bad[4, 40] ^= 1; bad[5, 7:39] = 0; bad[6, 7:39] = np.frombuffer(N.to_bytes(32, "big"), dtype=np.uint8)
        sc, ok = from_der(bad)
        assert list(ok) == [1, 1, 1, 0, 0, 0, 0] + [1] * 25 + [0, 0, 0]
        assert np.array_equal(sc[:3], arr[:3]) and not sc[3:7].any() and np.array_equal(sc[7:32], arr[7:32])


def test_sec1_der_scalar_marshalling(kats):
    _check_sec1_der(D.scalars_to_der, None, kats)
    for level in (1, 2):                           # ... and with the comb's uniform schedules (plume_set_sign_uniform covers the export: its scalars are secret keys)
        D.lib().ds_set_sign_uniform(level)
        try:
            _check_sec1_der(D.scalars_to_der, None, kats)
        finally:
            D.lib().ds_set_sign_uniform(0)


def test_affine_table_chain_and_its_zero_denominator_guard():
    """the table passes: (a) the tables of honest bases are P, theta P = P - lambda P, 2P, whatever else shares the lane; (b) a record that is no curve point and makes a
    denominator vanish (y = 0: the doubling divides by 2y; x = 0: theta P divides by (beta - 1) x) cannot poison the OTHER jobs of its lane -- the pass is redone with that
    denominator replaced.  (Such a record never reaches the builder through the API: bases are validated and replaced by G first.)"""
    rng = random.Random(12)
    pts = [O.G, O.pt_mul(rng.randrange(1, N), O.G), O.pt_mul(N - 1, O.G), O.pt_mul(rng.randrange(1, N), O.G)]
    rec = lambda p: p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")  # noqa: E731
    good = np.frombuffer(b"".join(rec(p) for p in pts), dtype=np.uint8).reshape(-1, 64)
    got = D.tables_raw(good)
    E = got.shape[1]
    assert E == 3
    want = [_table_rows(p) for p in pts]
    assert [[got[j, k].tobytes() for k in range(E)] for j in range(len(pts))] == want
    for bogus in ((5).to_bytes(32, "big") + bytes(32), bytes(32) + (3).to_bytes(32, "big")):      # y = 0; x = 0
        mixed = np.concatenate([good[:2], np.frombuffer(bogus, dtype=np.uint8).reshape(1, 64), good[2:]])
        got = D.tables_raw(mixed)
        for j, src in enumerate([0, 1, None, 2, 3]):
            if src is not None:
                assert [got[j, k].tobytes() for k in range(E)] == want[src], j


def test_twenty_bit_generator_window_build_of_the_device_headers(tmp_path):
    """The GPU build ships a 24-bit generator window (2^23 rows, 1 GiB: no CPU builds that per test run) whose digits travel in the THREE-byte form of the digit rows; the
    harness built with -DPLUME_GW=20 (2^19 rows, the same three-byte form, the same two-byte top digit) passes the recoding and whole-pipeline tests of this file
    (child process: the harness is loaded once per process).  The 24-bit table itself is exercised by the GPU parity tests."""
    import os
    import subprocess
    import sys
    if os.environ.get("PLUME_DEVSIM_SO"):
        pytest.skip("already inside the child run")
    from tests import _prebuild
    so = _prebuild.get("devsim_gw20")                  # compiled in the background since collection when the whole suite runs (tests/_prebuild.py); else here, the same command
    if so is None:
        so = tmp_path / "libplume_devsim_gw20.so"
        subprocess.run(_prebuild.devsim_gw20_cmd(so), check=True, capture_output=True, text=True)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", str(Path(__file__)), "-k", "wide or golden_verify or equation1 or crafted"],
                       env=dict(os.environ, PLUME_DEVSIM_SO=str(so)), capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])


# ------------------------------------------------------------------------------------------------ round 5: the short form of equation 1 (csrc/plume_eis.h)
_LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def _eis_cases(rng, nrand):
    n = O.N
    a1, b1 = 0x3086D221A7D46BCDE86C90E49284EB15, 0xE4437ED6010E88286F547FA90ABFE4C3
    cs = [0, 1, 2, 3, n - 1, n - 2, _LAM, _LAM + 1, _LAM * _LAM % n, 2**128, 2**64, 2**64 + 1, 2**65, 2**96, 2**127, 2**192, (n - 1) // 2, (n + 1) // 2, a1, b1, n - a1, n - b1]
    cs += [(x + y * _LAM) % n for x in (1, 2**63, 2**64, 2**65, 2**66, -2**64, 3 * 2**62) for y in (0, 1, -1, 2**63, 2**64, -2**65)]      # short on the gamma side already
    cs += [pow(rng.randrange(2, 2**70), -1, n) for _ in range(100)]                                                                       # tau short, upsilon = 1
    cs += [rng.randrange(1, 2**66) * pow(rng.randrange(1, 2**66), -1, n) % n for _ in range(100)]                                      # both short: the answer is (nearly) unique
    cs += [rng.getrandbits(k) % n for k in (8, 16, 32, 64, 96, 127, 129, 160, 200) for _ in range(10)]
    cs += [rng.randrange(1, n) for _ in range(nrand)]
    return [c % n for c in cs]


def test_eis_half_gcd_relation_and_bounds():
    """tau gamma = upsilon (mod pi) for every c -- i.e. (t0 + t1 lambda) c = u0 + u1 lambda (mod n) --, tau != 0, every coefficient below 2^67 (in fact <= 65 bits: the chain
    has seventeen 4-bit windows), no fallback; random and crafted challenges, c = 0 included (verify_non_zk admits it).  The quotients are double-precision estimates: the
    RELATION holds whatever they are, the BOUND is what this test measures."""
    rng = random.Random(20261002)
    cs = _eis_cases(rng, 20000)
    worst = 0
    for c, (tm1, t1, u0, u1, tau, ok) in zip(cs, D.eis_half_gcd(cs)):
        t0 = tm1 + 1
        assert tau == (t0 + t1 * _LAM) % O.N and tau != 0
        assert (tau * c - (u0 + u1 * _LAM)) % O.N == 0, hex(c)
        assert ok, hex(c)
        worst = max(worst, abs(tm1), abs(t1), abs(u0), abs(u1))
    assert worst.bit_length() <= 66, worst.bit_length()


def test_eis_pair_is_checked_before_it_is_used():
    """the scalar stage re-derives tau c == upsilon (mod n) from the pair it is about to use (eis_consistent) and falls back to the long form when it does not hold: true for
    every honest pair, crafted challenges included; false after any single tamper -- a coefficient's bit, a sign, tau itself, the pair of a neighbouring challenge"""
    rng = random.Random(99)
    cs = _eis_cases(rng, 4000)
    assert D.eis_consistent(cs).all()
    pairs = D.eis_half_gcd(cs)
    for which in (1, 2, 3, 4):
        got = D.eis_consistent(cs, which)
        for c, g, (_, _, _, u1, _, _) in zip(cs, got, pairs):
            if g:       # a tamper that changes nothing: the sign of a ZERO coefficient (upsilon = u0 for small c and for c near n); tau + 1 for c = 0 (both sides stay 0)
                assert (which == 2 and u1 == 0) or (which == 3 and c == 0), (which, hex(c))
        assert got.sum() < len(cs) // 20, (which, int(got.sum()))


def test_equation1_short_form_value_and_fallback():
    """the chain of the short form -- scalar stage, four tables, 64 doublings, the comb -- returns k G - upsilon pk - (tau - 1) R: checked through its meaning (for ANY s, c,
    pk, R it equals R + tau (s G - c pk - R)), which a valid signature turns into R itself.  Crafted scalars; the forced fallback (long form in the checked chain) gives the
    long form's value s G - c pk."""
    rng = random.Random(7)
    n = O.N
    G = O.G
    be = lambda p: p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")  # noqa: E731
    cases = []
    for c in _eis_cases(rng, 24)[1:]:
        if c == 0:
            continue
        sk, r = rng.randrange(1, n), rng.randrange(1, n)
        cases.append((sk, r, c, True))
        cases.append((sk, r, c, False))
    try:
        for sk, r, c, valid in cases[:160]:
            pk = O.pt_mul(sk, G)
            s = (r + sk * c) % n if valid else rng.randrange(1, n)
            R = O.pt_mul(r, G)
            if s == 0:
                continue
            got, lng = D.eq1_short(s.to_bytes(32, "big"), c.to_bytes(32, "big"), be(pk), be(R))
            assert not lng
            tau = D.eis_half_gcd([c])[0][4]
            # R + tau (s G - c pk - R)
            e = O.pt_add(O.pt_add(O.pt_mul(s, G), O.pt_neg(O.pt_mul(c, pk))), O.pt_neg(R))
            want = O.pt_add(R, O.pt_mul(tau, e)) if e is not None else R
            assert got == (be(want) if want is not None else bytes(64)), (hex(c), valid)
            if valid:
                assert got == be(R)
        D.set_eq1_short(2)
        for sk, r, c, valid in cases[:24]:
            pk = O.pt_mul(sk, G)
            s = (r + sk * c) % n
            R = O.pt_mul(r, G)
            got, lng = D.eq1_short(s.to_bytes(32, "big"), c.to_bytes(32, "big"), be(pk), be(R))
            assert lng and got == be(R)                       # the long form's value: s G - c pk
    finally:
        D.set_eq1_short(1)


@pytest.mark.parametrize("mode", [0, 2])
def test_verdicts_do_not_depend_on_the_form_of_equation1(mode):
    """the golden batches, the edge cases and a fuzzed batch (V1 verify, verify_non_zk V1 and V2) with the short form off (0) and with every item forced through the fallback
    (2) -- the default (1) is what every other test of this file runs"""
    from tests import _fuzz
    try:
        D.set_eq1_short(mode)
        for items in (GOLD["verify_v1"][:48], [e for e in GOLD["edge"] if e["version"] == 1]):
            mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
            got = D.verify_batch(1, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32), OC.arr(items, "r_point", 64),
                                 OC.arr(items, "hashed_to_curve_r", 64))
            assert [int(x) for x in got] == [it["ok"] for it in items]
        n = 96
        b = synth.sign_inputs(n, start=4242)
        for ver in (1, 2):
            signed = OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
            v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=31 + ver)
            if ver == 1:
                args = (1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
                assert np.array_equal(D.verify_batch(*args), OC.verify_batch(*args, nthreads=8))
            nz = (ver, b["msgs"], b["off"], signed["pk"], signed["nullifier"], signed["s"], signed["r_point"], signed["hashed_to_curve_r"], signed["c"])
            assert np.array_equal(D.verify_non_zk_batch(*nz), OC.verify_non_zk_batch(*nz, nthreads=8))
    finally:
        D.set_eq1_short(1)
