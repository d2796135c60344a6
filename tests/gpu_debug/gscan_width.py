"""Level 2 of the signer's uniform schedule: time of the multiplications by G (k_sign_gmul_uniform<2>) for the library named by PLUME_HIP_LIB (round 4: builds with PLUME_GSCAN_W edited to 4 / 5 / 6 in plume_ec.h; 5 was kept: 7.34 / 6.98 / 7.42 ms per 2^20 signs)."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from tests import synth
n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
e.set_sign_uniform(0)
want = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
e.set_sign_uniform(2)
for rep in range(3):
    got = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
print(e.version(), {k: round(v, 3) for k, v in e.last_stage_times()})
assert all(np.array_equal(got[k], want[k]) for k in want)
