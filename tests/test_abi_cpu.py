"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/plume_hip.h declares,
fails loudly without a GPU, and the host façade's marshalling / error mapping follows the reference.
(No compute calls: there is no GPU here.  The façade test drives a FAKE engine built on the oracle — a test double
living in tests/, never part of the product.)"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _built():
    import zk_nullifier_sig_amd as plume
    return plume.library_path().exists()


needs_lib = pytest.mark.skipif(not _built(), reason="libplume_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")


@needs_lib
def test_library_exports_every_declared_symbol():
    import zk_nullifier_sig_amd as plume
    from zk_nullifier_sig_amd import capi
    header = (ROOT / "include" / "plume_hip.h").read_text()
    declared = sorted(set(re.findall(r"\b(plume_[a-z0-9_]+)\s*\(", header)))
    assert sorted(capi.exported_symbols()) == declared
    lib = C.CDLL(str(plume.library_path()))
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.plume_version.restype = C.c_char_p
    assert b"gfx950" in lib.plume_version()


@needs_lib
def test_fails_loudly_without_a_gpu():
    import torch
    import zk_nullifier_sig_amd as plume
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(plume.PlumeHipError, match="no CPU fallback"):
        plume.Engine(0)


def test_product_package_never_imports_the_oracle():
    for p in list((ROOT / "zk-nullifier-sig_amd").rglob("*.py")) + list((ROOT / "zk-nullifier-sig_amd").rglob("*.h*")) + list((ROOT / "zk_nullifier_sig_amd").rglob("*.py")):
        txt = p.read_text()
        assert "plume_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, p


class FakeEngine:
    """test double with Engine's batch signature, computing with the C oracle"""

    def sign_batch(self, version, msgs, off, sk, r, pk_in=None):
        from tests import _oracle_c as OC
        o = OC.sign_batch(version, np.ascontiguousarray(msgs), np.ascontiguousarray(off), np.ascontiguousarray(sk).reshape(-1, 32),
                          np.ascontiguousarray(r).reshape(-1, 32), None if pk_in is None else np.ascontiguousarray(pk_in).reshape(-1, 64))
        o.pop("h")
        return o

    def verify_batch(self, version, msgs, off, pk, nul, c, s, r_point=None, hr=None):
        from tests import _oracle_c as OC
        f = lambda a, w: None if a is None else np.ascontiguousarray(a).reshape(-1, w)  # noqa: E731
        return OC.verify_batch(version, np.ascontiguousarray(msgs), np.ascontiguousarray(off), f(pk, 64), f(nul, 64), f(c, 32), f(s, 32), f(r_point, 64), f(hr, 64))


    def verify_non_zk_batch(self, version, msgs, off, pk, nul, s, r_point, hr, digest_private):
        from tests import _oracle_c as OC
        f = lambda a, w: np.ascontiguousarray(a).reshape(-1, w)  # noqa: E731
        return OC.verify_non_zk_batch(version, np.ascontiguousarray(msgs), np.ascontiguousarray(off), f(pk, 64), f(nul, 64), f(s, 32), f(r_point, 64), f(hr, 64), f(digest_private, 32))


def test_facade_marshalling_and_reference_shapes(kats):
    import zk_nullifier_sig_amd as plume
    v = kats["plume_vector"]
    eng = FakeEngine()
    calls = []

    class Mock:  # rust-k256/tests/signing.rs:23-44, plus a rejected first draw to exercise SecretKey::random's rejection sampling
        def fill_bytes(self, n):
            calls.append(n)
            return bytes(32) if len(calls) == 1 else bytes.fromhex(v["r"])

    sk = plume.SecretKey.from_bytes(bytes.fromhex(v["sk"]))
    msg = v["msg_utf8"].encode()
    s1 = plume.PlumeSigner(sk, True, eng).try_sign_with_rng(Mock(), msg)
    assert calls == [32, 32]
    assert s1.c.to_bytes().hex() == v["c_v1"] and s1.s.to_bytes().hex() == v["s_v1"] and s1.v1specific is not None
    assert s1.message == msg and s1.pk.to_encoded_point(True).hex() == kats["wasm_readme"]["pk_sec1"]
    assert s1.nullifier.to_encoded_point(True).hex() == kats["wasm_readme"]["nullifier_sec1"]
    assert s1.verify(eng)
    s2 = plume.PlumeSignature.sign_v2(sk, msg, Mock(), eng)
    assert s2.v1specific is None and s2.c.to_bytes().hex() == v["c_v2"] and s2.verify(eng)
    pub, prv = plume.sign_with_r((s1.pk, sk.value), msg, int(v["r"], 16), plume.PlumeVersion.V2, eng)
    assert prv.digest_private == int(v["c_v2"], 16) and pub.s == int(v["s_v2"], 16) and pub.variant is plume.PlumeVersion.V2
    # verify_non_zk (rust-arkworks/src/tests.rs:28-78): Ok(true) / Ok(false) / Err
    assert plume.verify_non_zk((pub, prv), s1.pk, msg, plume.PlumeVersion.V2, eng) is True
    assert plume.verify_non_zk((pub, prv), s1.pk, msg, plume.PlumeVersion.V1, eng) is False
    with pytest.raises(plume.SignatureError):
        plume.verify_non_zk((pub, prv), plume.AffinePoint(), msg, plume.PlumeVersion.V2, eng)
    prv.zeroize()
    assert prv.digest_private == 0 and prv.r_point.is_identity
    # type invariants of the Rust types
    with pytest.raises(ValueError):
        plume.NonZeroScalar(0)
    with pytest.raises(ValueError):
        plume.AffinePoint(1, 1)
    assert plume.AffinePoint().to_encoded_point() == b"\x00"
    assert plume.AffinePoint.generator().to_encoded_point().hex() == kats["enc_G"]["hex"]
    with pytest.raises(plume.SignatureError):
        plume.sign_with_r((plume.AffinePoint(), 5), msg, 7, plume.PlumeVersion.V1, eng)


def test_generated_field_multiplication_is_current():
    """zk-nullifier-sig_amd/csrc/plume_fe_mul.inc is generated by gen_fe_mul.py and committed: the two must not drift apart"""
    import subprocess
    import sys
    csrc = ROOT / "zk-nullifier-sig_amd" / "csrc"
    out = subprocess.run([sys.executable, str(csrc / "gen_fe_mul.py")], capture_output=True, text=True, check=True).stdout
    assert out == (csrc / "plume_fe_mul.inc").read_text(), "run `python gen_fe_mul.py > plume_fe_mul.inc` in zk-nullifier-sig_amd/csrc"


def build_abi_smoke(tmp_path):
    """compile tests/abi_c/abi_smoke.c with plain gcc against include/plume_hip.h and the in-tree library"""
    import subprocess
    import zk_nullifier_sig_amd as plume
    exe = tmp_path / "abi_smoke"
    libdir = plume.library_path().parent
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "tests" / "abi_c" / "abi_smoke.c"), "-L", str(libdir), "-lplume_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, text=True)
    return exe


def build_reference_tests(tmp_path):
    """compile tests/abi_cpp/reference_tests.cpp (the reference's own tests restated against include/plume.hpp, the C++ host side) with plain g++"""
    import subprocess
    import zk_nullifier_sig_amd as plume
    exe = tmp_path / "reference_tests"
    libdir = plume.library_path().parent
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "tests" / "abi_cpp" / "reference_tests.cpp"), "-L", str(libdir),
                    "-lplume_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, text=True)
    return exe


def write_kg_vectors(tmp_path):
    """the reference's 100 k*G SEC1 vectors (rust-arkworks/src/tests/test_vectors.rs, extracted into tests/golden/reference_kats.json) as `k compressed-hex` lines"""
    import json
    kats = json.loads((ROOT / "tests" / "golden" / "reference_kats.json").read_text())
    path = tmp_path / "kg_vectors.txt"
    path.write_text("".join(f"{k} {comp}\n" for k, comp, _ in kats["sec1_kG"]["vectors"]))
    return path


@needs_lib
def test_cpp_host_side_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """include/plume.hpp (plume_rustcrypto / plume_arkworks shapes in C++ over the C ABI) compiles warning-free with g++ -std=c++17 and, without a GPU, its
    engine reports PLUME_ERR_NODEV instead of computing anywhere else"""
    import subprocess
    import torch
    exe = build_reference_tests(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (the GPU suite runs the program)")
    r = subprocess.run([str(exe), str(write_kg_vectors(tmp_path))], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


@needs_lib
def test_c_caller_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """a non-Python FFI caller: the header is valid C, every symbol it uses links, and without a GPU plume_init reports PLUME_ERR_NODEV"""
    import subprocess
    import torch
    exe = build_abi_smoke(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (the GPU suite runs the program)")
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr
