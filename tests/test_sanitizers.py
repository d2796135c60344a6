"""Sanitizers in the suite.  CPU only -- the GPU pool offers no sanitizer.

* the host logic of libplume_hip.so that touches no GPU (csrc/plume_host_logic.h: shard / sub-batch / piece bounds, offset rebasing, the DER and register parsers --
  the same bodies plume_capi.hip calls) fuzzed by tests/hostsim/hostsim_fuzz.cpp under AddressSanitizer + UBSan with exactly-sized heap buffers;
* the device headers compiled for the host (tests/devsim) built with -fsanitize=address,undefined into a temporary directory and tests/test_devsim.py run against that
  build (PLUME_DEVSIM_SO) under LD_PRELOAD=libasan.
Both skip cleanly where g++ or libasan is missing."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "zk-nullifier-sig_amd" / "csrc"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def _libasan():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    p = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not p or not Path(p).is_absolute() or not Path(p).exists():
        pytest.skip("libasan is not installed")
    return p


def test_host_logic_fuzzed_under_asan_ubsan(tmp_path):
    _libasan()
    exe = tmp_path / "hostsim_fuzz"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", *SAN, "-Wall", "-Wextra", "-Werror", f"-I{CSRC}", str(ROOT / "tests" / "hostsim" / "hostsim_fuzz.cpp"), "-o", str(exe)])
    for seed in (1, 20261002):
        r = subprocess.run([str(exe), str(seed), "30000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-4000:])
        assert "ok" in r.stdout


def test_the_capi_uses_the_fuzzed_bodies():
    """the harness would prove nothing if plume_capi.hip kept copies of its own: the ABI implementation must call plume_host:: for each of them and define none itself"""
    src = (CSRC / "plume_capi.hip").read_text()
    assert '#include "plume_host_logic.h"' in src
    for fn in ("shard_bounds", "sub_batch_bounds", "piece_schedule", "rebase_offsets", "sec1_der_to_scalars", "registers_from_be"):
        assert f"plume_host::{fn}(" in src, fn
    assert "strtoull" not in src and "0x30, 0x6b" not in src            # the schedule parser and the DER template live in the header only


def test_device_headers_host_build_under_asan_ubsan(tmp_path):
    """tests/devsim built with ASan + UBSan (every limb-bound assertion on), tests/test_devsim.py run against it."""
    asan = _libasan()
    from tests import _prebuild
    so = _prebuild.get("devsim_asan")                  # compiled in the background since collection when the whole suite runs (tests/_prebuild.py); else here, the same command
    if so is None:
        so = tmp_path / "libplume_devsim_asan.so"
        subprocess.check_call(_prebuild.devsim_asan_cmd(so))
    env = dict(os.environ, PLUME_DEVSIM_SO=str(so), LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", PYTHONPATH=str(ROOT))
    try:                                            # four workers where pytest-xdist exists (the run is 80 s of single-threaded table arithmetic under ASan otherwise)
        import xdist  # noqa: F401
        par = ["-n", "4"]
    except ImportError:
        par = []
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_devsim.py"), "-x", "-q", "-p", "no:cacheprovider", *par], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
