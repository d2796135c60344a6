"""The HOST side of libplume_hip.so under the sanitizers, on eight (mock) devices -- CPU only ("host C++ with worker threads ... never run under any
sanitizer and never on two devices").

tests/hostsim builds csrc/plume_capi.hip -- the file the product ships, unchanged -- as plain C++ against a mock of the HIP runtime calls it makes
(tests/hostsim/mockhip/hip/hip_runtime.h: streams are queues, run lazily or in random order, so everything the library did not order happens in the worst order), with the
kernels as host loops over the same per-lane bodies and the same buffers (tests/hostsim/host_launch.cpp), and drives it through the C ABI against the C oracle
(tests/hostsim/pipeline_driver.cpp): host-pointer verify / sign / aggregate / hash_to_curve / DER / first-occurrence calls with small pieces and chunks on pageable,
page-locked and registered arrays, device-resident calls on caller streams with one and two batches in flight, plume_init_multi over eight devices and over a repeated device,
teardown with work queued, leak accounting of every runtime object.

* AddressSanitizer + UBSan (leak detection on) over every group, lazy and random schedules -- including every device-resident entry point with the overlapped sub-batch
  order (`devapi`) and allocation-failure injection (`faults`: the k-th hipMalloc / hipHostMalloc of a call fails, for every k: an error code, nothing leaked, the context
  usable afterwards);
* ThreadSanitizer over the multi-device group (eight shard worker threads + the caller's) and the device group;
* mutants: the harness has to FAIL when an event wait, the slot hand-back, the workspace guard or the final quiesce is taken out of the library -- otherwise it proves nothing.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HOSTSIM = ROOT / "tests" / "hostsim"
CSRC = ROOT / "zk-nullifier-sig_amd" / "csrc"
BUILDS = {"asan": "address,undefined", "tsan": "thread", "plain": ""}


def _have(lib):
    p = subprocess.run(["g++", f"-print-file-name={lib}"], capture_output=True, text=True).stdout.strip()
    return bool(p) and Path(p).is_absolute() and Path(p).exists()


@pytest.fixture(scope="module")
def builds(tmp_path_factory):
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    if not _have("libasan.so") or not _have("libtsan.so"):
        pytest.skip("libasan / libtsan is not installed")
    out = tmp_path_factory.mktemp("hostsim")

    from tests import _prebuild

    def build(name):
        pre = _prebuild.get(f"hostsim_{name}")           # compiled in the background since collection when the whole suite runs (tests/_prebuild.py); else here, the same command
        if pre is not None and (pre / "pipeline_driver").exists():
            return pre
        r = subprocess.run(["make", "-C", str(HOSTSIM), f"SAN={BUILDS[name]}", f"OUT={out / name}", "-j3"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-4000:])
        return out / name
    with ThreadPoolExecutor(3) as ex:
        return dict(zip(BUILDS, ex.map(build, BUILDS)))


def _run(exe, group, sched=None, seed=1, extra_env=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("PLUME_MOCK_SCHED", None)
    if sched:
        env["PLUME_MOCK_SCHED"] = sched
    env.update(extra_env or {})
    r = subprocess.run([str(exe), group, str(seed)], capture_output=True, text=True, timeout=1500, env=env)
    return r


def _all_ok(results):
    for what, r in results:
        assert r.returncode == 0, (what, r.stdout[-1000:], r.stderr[-5000:])
        assert ": ok (" in r.stdout, (what, r.stdout[-1000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (what, r.stderr[-5000:])


def test_host_side_under_asan_ubsan_on_eight_devices(builds):
    exe = builds["asan"] / "pipeline_driver"
    jobs = [("faults", None, 1), ("multi", None, 1), ("benchseq", None, 1), ("benchseq", "random:6", 2), ("verify", None, 1), ("sign", None, 1), ("misc", None, 2), ("device", None, 1), ("device", "random:1", 1), ("device", "random:2", 2),
            ("verify", "random:3", 3), ("sign", "eager", 1), ("sign", "random:7", 2), ("sign", "random:8", 3), ("devapi", None, 1), ("devapi", "random:4", 2)]      # (sign under the random scheduler: the two-lane uniform pieces of round 6)
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda j: (j, _run(exe, j[0], j[1], j[2])), jobs))
    _all_ok(res)
    # the pipeline's own timeline (PLUME_HOST_TRACE=1: six timing events per piece, read after the call) through the same runs
    r = _run(exe, "verify", None, 4, {"PLUME_HOST_TRACE": "1"})
    _all_ok([("host trace", r)])
    assert "plume_host_trace:" in r.stderr and "piece  1 lane" in r.stderr
    # a machine with another GPU, a machine with none: refused by name, no context
    _all_ok([("gfx942", _run(exe, "noarch", None, 1, {"PLUME_MOCK_ARCH": "gfx942"})), ("no device", _run(exe, "noarch", None, 1, {"PLUME_MOCK_DEVICES": "0"}))])


def test_shard_threads_under_tsan(builds):
    exe = builds["tsan"] / "pipeline_driver"
    jobs = [("multi", None, 1), ("benchseq", None, 3), ("device", "random:5", 1)]
    with ThreadPoolExecutor(2) as ex:
        res = list(ex.map(lambda j: (j, _run(exe, j[0], j[1], j[2])), jobs))
    _all_ok(res)


# (text taken out of plume_capi.hip, replacement, group, scheduler): each must make the harness fail
MUTANTS = {
    "download does not wait for the kernels": ("HIPCHK(hipStreamWaitEvent(ctx->down, prev.sl->computed, 0));", "", "verify", None),
    "kernels do not wait for the upload": ("HIPCHK(hipStreamWaitEvent(on->stream, sl.ready, 0));", "", "sign", None),
    "a staging slot is reused before its piece has left it": ("if (sl.in_flight) { HIPCHK(hipEventSynchronize(sl.drained)); sl.in_flight = false; }", "", "verify", None),
    "calls on different streams do not queue for the workspace": ("if (ctx->ws_used && ctx->ws_stream != st) HIPCHK(hipStreamWaitEvent(st, ctx->ws_free, 0));", "", "device", "random:1"),
    "a host-pointer call returns with copies still queued": ("    const int rc = body();\n    quiesce(ctx);", "    const int rc = body();", "verify", None),
}


def test_the_harness_fails_when_a_dependency_is_taken_out(builds, tmp_path):
    plain = builds["plain"]
    flags = ["-x", "c++", "-O1", "-std=c++17", "-ffp-contract=off", "-DPLUME_GW=16", "-DPLUME_COMB_W=10", f"-I{HOSTSIM / 'mockhip'}", '-DPLUME_BUILD_ID="mutant"']

    def mutant(item):
        k, (name, (old, new, group, sched)) = item
        d = tmp_path / f"m{k}"
        (d / "pkg" / "csrc").mkdir(parents=True)
        (d / "include").mkdir()
        shutil.copy(ROOT / "include" / "plume_hip.h", d / "include")
        for f in CSRC.iterdir():
            if f.suffix in (".h", ".inc", ".hip"):
                shutil.copy(f, d / "pkg" / "csrc")
        src = d / "pkg" / "csrc" / "plume_capi.hip"
        text = src.read_text()
        assert text.count(old) >= 1, f"mutant '{name}': its text is no longer in plume_capi.hip -- update the mutant"
        src.write_text(text.replace(old, new, 1))
        subprocess.check_call(["g++", *flags, f"-I{d / 'pkg' / 'csrc'}", "-c", str(src), "-o", str(d / "capi.o")])
        subprocess.check_call(["g++", "-o", str(d / "driver"), str(d / "capi.o"), *(str(plain / o) for o in ("launch.o", "oracle.o", "driver.o")), "-lpthread"])
        return name, _run(d / "driver", group, sched)
    with ThreadPoolExecutor(5) as ex:
        res = list(ex.map(mutant, enumerate(MUTANTS.items())))
    survivors = [name for name, r in res if r.returncode == 0]
    assert not survivors, f"the harness did not notice: {survivors}"
    # ... and the unmutated library passes the same runs in the same build
    _all_ok([((g, s), _run(plain / "pipeline_driver", g, s)) for g, s in {(m[2], m[3]) for m in MUTANTS.values()}])
