"""(round 5) Which path do the library's downloads take?  One host-pointer sign (and verify) of 2^18 items from page-locked arrays; run under AMD_LOG_LEVEL=4 and grep the
runtime's log for "HSA Copy" (copy engine) against blit-kernel launches.  PLUME_NO_TORCH_PRELOAD=1 runs the same on /opt/rocm's libamdhip64 instead of the one torch bundles."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
eng = plume.Engine(0)
print("runtime:", [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][:1], file=sys.stderr)
b = synth.sign_inputs(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
print("=== MARK sign begin", file=sys.stderr, flush=True)
t0 = time.perf_counter()
eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
print(f"=== MARK sign end {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr, flush=True)
eng.close()
