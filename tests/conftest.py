import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the process's first HIP call: two caller streams on one hardware queue run one after the other (include/plume_hip.h, plume_set_in_flight)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not silently pass: only skip when the user did not ask for gpu tests
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# which slow-to-compile native artefacts the selected tests need (tests/_prebuild.py compiles them in the background while the quick tests run)
_PREBUILD_NEEDS = {
    "test_device_headers_host_build_under_asan_ubsan": ["devsim_asan"],
    "test_twenty_bit_generator_window_build_of_the_device_headers": ["devsim_gw20"],
    "test_host_side_under_asan_ubsan_on_eight_devices": ["hostsim_asan"],
    "test_shard_threads_under_tsan": ["hostsim_tsan"],
    "test_the_harness_fails_when_a_dependency_is_taken_out": ["hostsim_plain"],
}


def pytest_collection_finish(session):
    from tests import _prebuild
    names = []
    for item in session.items:
        if any(m.name == "skip" for m in item.iter_markers()):
            continue
        for n in _PREBUILD_NEEDS.get(item.originalname if hasattr(item, "originalname") else item.name, []):
            if n not in names:
                names.append(n)
    if len(session.items) >= 20:          # a whole-suite run: for one or two selected tests the test builds its artefact itself, nothing is gained by a second path
        _prebuild.start(names)


def pytest_sessionfinish(session, exitstatus):
    from tests import _prebuild
    _prebuild.cleanup()


@pytest.fixture(scope="session")
def kats():
    import json
    return json.loads((ROOT / "tests" / "golden" / "reference_kats.json").read_text())
