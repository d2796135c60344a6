#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors and constants into reference_kats.json.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_reference_kats.py

Only DATA is extracted (hex strings / integers that the reference's tests assert, and the curve
constants of its in-repo curve config); no reference source text is kept.  Every entry records
the file:line it was read from so the parity chain can be audited.
"""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).with_name("reference_kats.json")


def lines(rel):
    return (REF / rel).read_text().splitlines()


def find(rel, pattern, start=0, group=1):
    """first regex match at/after line `start` (0-based) -> (value, 'rel:lineno')"""
    ls = lines(rel)
    rx = re.compile(pattern)
    for i in range(start, len(ls)):
        m = rx.search(ls[i])
        if m:
            return m.group(group), f"{rel}:{i + 1}", i
    raise SystemExit(f"pattern {pattern!r} not found in {rel} after line {start}")


def hex_after(rel, anchor, n=1, hexlen=64):
    """the n-th 64-hex-digit literal at/after the first line matching `anchor`"""
    _, _, i = find(rel, anchor, group=0)
    ls = lines(rel)
    rx = re.compile(r'"([0-9a-fA-F]{%d})"' % hexlen)
    got = 0
    for j in range(i, len(ls)):
        m = rx.search(ls[j])
        if m:
            got += 1
            if got == n:
                return m.group(1).lower(), f"{rel}:{j + 1}"
    raise SystemExit(f"hex literal #{n} after {anchor!r} not found in {rel}")


def main():
    k = {"_generated_by": "tests/golden/make_reference_kats.py", "_source": "plume-sig/zk-nullifier-sig @ 2025-06-20"}

    # ---- 1. the pinned (sk, r, msg) triple and its c/s (rust-k256/tests/signing.rs) -------------------
    f = "rust-k256/tests/signing.rs"
    msg, src, _ = find(f, r'const message: &\[u8; 29\] = b"([^"]+)"')
    vec = {"msg_utf8": msg, "msg_src": src}
    for name, anchor in [("r", r"^const R:"), ("sk", r"^const SK:"), ("c_v1", r"^const V1_C:"), ("s_v1", r"^const V1_S:"),
                         ("c_v2", r"^const V2_C:"), ("s_v2", r"^const V2_S:")]:
        vec[name], vec[name + "_src"] = hex_after(f, anchor)
    # ---- 2. intermediates (rust-arkworks/src/tests.rs) -------------------------------------------------
    f = "rust-arkworks/src/tests.rs"
    for name, anchor in [("pk", r"fn test_against_zk_nullifier_sig_pk"), ("g_r", r"fn test_against_zk_nullifier_sig_g_r"),
                         ("h", r"fn test_against_zk_nullifier_sig_h\b"), ("h_r", r"fn test_against_zk_nullifier_sig_h_r"),
                         ("nullifier", r"fn test_against_zk_nullifier_sig_h_sk")]:
        vec[name + "_x"], vec[name + "_x_src"] = hex_after(f, anchor, 1)
        vec[name + "_y"], vec[name + "_y_src"] = hex_after(f, anchor, 2)
    # arkworks asserts of c/s (BigInt!("0x..")) — must equal the k256 ones
    ls = lines(f)
    ark = [(m.group(1).lower(), f"{f}:{i + 1}") for i, l in enumerate(ls) for m in [re.search(r'BigInt!\("0x([0-9a-f]{64})"\)', l)] if m]
    assert len(ark) == 4, ark
    vec["arkworks_c_v1"], vec["arkworks_s_v1"], vec["arkworks_c_v2"], vec["arkworks_s_v2"] = [a[0] for a in ark]
    vec["arkworks_cs_src"] = [a[1] for a in ark]
    assert vec["arkworks_c_v1"] == vec["c_v1"] and vec["arkworks_s_v2"] == vec["s_v2"]
    # C_V1 in verification.rs
    vec["verification_c_v1"], vec["verification_c_v1_src"] = hex_after("rust-k256/tests/verification.rs", r"^const C_V1")
    k["plume_vector"] = vec

    # ---- 3. literal h2c preimage (TS test) --------------------------------------------------------------
    f = "circuits/circom/test/javascript/test/hashToCurve.test.ts"
    txt = (REF / f).read_text()
    m = re.search(r"const testPreimage = \[([^\]]+)\]", txt)
    pre = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    hx = re.findall(r'"([0-9a-f]{64})"', txt)
    k["h2c_preimage"] = {"preimage_hex": bytes(pre).hex(), "x": hx[0], "y": hx[1], "src": f + ":5-19"}

    # ---- 4. h2c("abc") -----------------------------------------------------------------------------------
    f = "rust-k256/tests/verification.rs"
    x, sx = hex_after(f, r"fn test_hash_to_curve", 1)
    y, sy = hex_after(f, r"fn test_hash_to_curve", 2)
    k["h2c_abc"] = {"msg_utf8": "abc", "x": x, "y": y, "src": [sx, sy]}

    # ---- 5. RFC 9380 J.8.1 vector quoted in the arkworks tests ------------------------------------------
    f = "rust-arkworks/src/secp256k1/tests.rs"
    ls = lines(f)
    _, _, i0 = find(f, r"fn test_h2c", group=0)
    comment = []
    for j in range(i0, len(ls)):
        comment.append(ls[j])
        if "*/" in ls[j]:
            break
    blob = "\n".join(comment)
    rfc = {"msg_utf8": "", "src": f"{f}:{i0 + 2}-{i0 + len(comment)}"}
    for key, name in [("u[0]", "u0"), ("u[1]", "u1"), ("Q0.x", "q0_x"), ("Q0.y", "q0_y"), ("Q1.x", "q1_x"), ("Q1.y", "q1_y"),
                      ("P.x", "p_x"), ("P.y", "p_y")]:
        m = re.search(re.escape(key) + r"\s*=\s*([0-9a-f]+)\s*\n\s*([0-9a-fA-F]+)", blob)
        a, b = m.group(1), m.group(2)
        val = a if len(a) == 64 else a + b
        assert len(val) == 64, (key, val)
        rfc[name] = val.lower()
    decs = [(m.group(1), f"{f}:{i + 1}") for i, l in enumerate(ls) for m in [re.search(r'^\s*"(\d{60,})"', l)] if m]
    rfc["u0_dec"], rfc["p_x_dec"], rfc["p_y_dec"] = [d[0] for d in decs]
    rfc["dec_src"] = [d[1] for d in decs]
    assert int(rfc["u0_dec"]) == int(rfc["u0"], 16) and int(rfc["p_x_dec"]) == int(rfc["p_x"], 16)
    k["rfc9380_empty"] = rfc

    # ---- 6. SEC1 k*G vectors ------------------------------------------------------------------------------
    f = "rust-arkworks/src/tests/test_vectors.rs"
    txt = (REF / f).read_text()
    trip = re.findall(r'\(\s*(\d+)u64,\s*String::from\("([0-9A-F]+)"\),\s*String::from\("([0-9A-F]+)"\),\s*\)', txt)
    assert len(trip) == 100, len(trip)
    k["sec1_kG"] = {"src": f + ":1-504", "vectors": [[int(a), b.lower(), c.lower()] for a, b, c in trip]}
    k["enc_G"] = dict(zip(("hex", "src"), hex_after("rust-k256/src/lib.rs", r"fn test_encode_pt", 1, 66)))

    # ---- 7. wasm README sample (sk as SEC1-DER, pk and nullifier as SEC1 compressed) -----------------------
    f = "javascript/README.md"
    txt = (REF / f).read_text()
    arrs = re.findall(r"\[([\d,\s]+)\]", txt)
    byte_arrs = [bytes(int(x) for x in a.replace("\n", " ").split(",") if x.strip()) for a in arrs]
    der = next(b for b in byte_arrs if len(b) > 100 and b[0] == 48)
    nul = next(b for b in byte_arrs if len(b) == 33 and b[0] == 3 and b[1] == 87)
    pk = next(b for b in byte_arrs if len(b) == 33 and b[0] == 3 and b[1] == 12)
    # `s` of the README's sample output, as SEC1-DER (SecretKey::from(value.s).to_sec1_der(), javascript/src/lib.rs:101-104): the listing is cut after
    # 100 of its 109 bytes ("... 9 more items") -- scalar, the whole x and 23 bytes of y of s*G are there
    m = re.search(r"\bs: \[([\d,\s]+)\.\.\. 9 more items", txt)
    s_pre = bytes(int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip())
    assert len(s_pre) == 100 and s_pre[:7] == der[:7]
    k["wasm_readme"] = {"sk_sec1_der": der.hex(), "nullifier_sec1": nul.hex(), "pk_sec1": pk.hex(), "s_sec1_der_first_100_bytes": s_pre.hex(),
                        "src": f + ":26-32,48-54,58-72,73-79"}

    # ---- 8. constants ---------------------------------------------------------------------------------------
    c = {}
    c["p_dec"], c["p_src"], _ = find("rust-arkworks/src/secp256k1/fields/fq.rs", r'#\[modulus = "(\d+)"\]')
    c["n_dec"], c["n_src"], _ = find("rust-arkworks/src/secp256k1/fields/fr.rs", r'#\[modulus = "(\d+)"\]')
    f = "rust-arkworks/src/secp256k1/curves/mod.rs"
    ls = lines(f)
    mont = [(m.group(1), i + 1) for i, l in enumerate(ls) for m in [re.search(r'MontFp!\("(-?(?:0x)?[0-9a-fA-F]+)"\)', l)] if m]

    def val(s):
        return int(s, 16) if s.startswith("0x") else int(s)

    vals = [(val(s), ln) for s, ln in mont]
    # order of appearance: b, Gx, Gy, A', B', gen'x, gen'y, Z, then 4+4+4+4 isogeny coefficients
    names = ["b", "gx", "gy", "iso_a", "iso_b", "iso_gx", "iso_gy", "z"]
    assert len(vals) == len(names) + 16, len(vals)
    for nme, (v, ln) in zip(names, vals):
        c[nme] = str(v)
        c[nme + "_src"] = f"{f}:{ln}"
    iso = vals[len(names):]
    for t, nme in enumerate(["x_num", "x_den", "y_num", "y_den"]):
        c["iso_" + nme] = [hex(v) for v, _ in iso[4 * t:4 * t + 4]]  # ascending degree, as in the reference
        c["iso_" + nme + "_src"] = f"{f}:{iso[4 * t][1]}-{iso[4 * t + 3][1]}"
    c["dst"], c["dst_src"], _ = find("rust-k256/src/lib.rs", r'pub const DST: &\[u8\] = b"([^"]+)"')
    dst2, src2, _ = find("rust-arkworks/src/lib.rs", r'new\(b"([^"]+)"\)')
    assert dst2 == c["dst"]
    c["dst_arkworks_src"] = src2
    k["constants"] = c

    OUT.write_text(json.dumps(k, indent=1) + "\n")
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    sys.exit(main())
