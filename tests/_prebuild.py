"""Native test artefacts that take tens of seconds to compile -- the sanitizer builds of the host-side harness (tests/hostsim), the ASan build of the device headers' host
harness, its 20-bit-window build -- are started in the background as soon as collection shows that a selected test needs them (tests/conftest.py), so that they compile
while the quick tests run instead of in front of their own test.  A test asks `get(name)`: the finished artefact's path, or None (not started, failed, no compiler) -- in which
case the test builds it itself, exactly as before.  Nothing here changes WHAT is built: the commands are the tests' own."""
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "zk-nullifier-sig_amd" / "csrc"
HOSTSIM = ROOT / "tests" / "hostsim"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
HOSTSIM_BUILDS = {"asan": "address,undefined", "tsan": "thread", "plain": ""}

_pool = None
_jobs = {}
_dir = None


def devsim_asan_cmd(so):
    return ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", *SAN, "-DPLUME_FE_CHECK", "-DPLUME_GW=16", "-DPLUME_COMB_W=14", f"-I{CSRC}", "-o", str(so), str(ROOT / "tests" / "devsim" / "devsim.cpp")]


def devsim_gw20_cmd(so):
    return ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DPLUME_FE_CHECK", "-DPLUME_GW=20", "-DPLUME_COMB_W=14", f"-I{CSRC}", "-o", str(so), str(ROOT / "tests" / "devsim" / "devsim.cpp")]


def hostsim_cmd(name, out):
    return ["make", "-C", str(HOSTSIM), f"SAN={HOSTSIM_BUILDS[name]}", f"OUT={out}", "-j2"]


def _run(cmd, result):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    return Path(result) if r.returncode == 0 else None


def start(names):
    """called once, from conftest, with the artefacts the selected tests need"""
    global _pool, _dir
    if _pool is not None or not names or not shutil.which("g++") or os.environ.get("PLUME_NO_PREBUILD") or os.environ.get("PLUME_DEVSIM_SO") or os.environ.get("PYTEST_XDIST_WORKER"):
        return                                           # (child runs of the harness tests and xdist workers build nothing in the background: their tests build what they need)
    _dir = Path(tempfile.mkdtemp(prefix="plume_prebuild_"))
    _pool = ThreadPoolExecutor(4)
    for n in names:
        if n == "devsim_asan":
            _jobs[n] = _pool.submit(_run, devsim_asan_cmd(_dir / "libplume_devsim_asan.so"), _dir / "libplume_devsim_asan.so")
        elif n == "devsim_gw20":
            _jobs[n] = _pool.submit(_run, devsim_gw20_cmd(_dir / "libplume_devsim_gw20.so"), _dir / "libplume_devsim_gw20.so")
        elif n.startswith("hostsim_") and shutil.which("make"):
            b = n[len("hostsim_"):]
            _jobs[n] = _pool.submit(_run, hostsim_cmd(b, _dir / n), _dir / n)


def get(name):
    f = _jobs.get(name)
    if f is None:
        return None
    try:
        return f.result(timeout=1500)
    except Exception:
        return None


def cleanup():
    global _pool
    if _pool is not None:
        _pool.shutdown(wait=True)
        _pool = None
    if _dir is not None:
        shutil.rmtree(_dir, ignore_errors=True)
