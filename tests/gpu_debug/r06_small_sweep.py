"""Round 6: (a) at which call size equation 1's short form starts to pay now that the scalar stage (half-GCD included) runs in role B of the two-role ingest kernel
(VERDICT r5 next #3), and (b) at which call size a second batch in flight starts to pay (two 2^16 calls side by side measured slower than one after
the other).  One process, one box, interleaved rounds; back-to-back calls on one stream (a) / alternating on two streams (b).
    python3 tests/gpu_debug/r06_small_sweep.py [a|b|ab]"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth

what = sys.argv[1] if len(sys.argv) > 1 else "ab"
dev = torch.device("cuda:0")
os.environ["PLUME_IN_FLIGHT_MIN"] = "0"          # (b) measures the lanes at every size
eng = plume.Engine(0)
print(eng.version())
N = 1 << 18
b = synth.sign_inputs(N)
signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, signed)
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
mb = int(v["off"][-1])
exp = synth.expected_ok(N)
oks = [torch.zeros(N, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def call(n, k=0, st=None):
    eng.verify_batch_device(1, n, t["msgs"], off, mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], oks[k], stream=st)


def b2b(n, reps):
    for _ in range(3):
        call(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(50):
    call(1 << 16)                                  # clocks up
torch.cuda.synchronize()
if "a" in what:
    print("(a) equation 1: long form (mode 0) vs short form (mode 3), ms per call, back to back, min of 3 interleaved rounds; stage times of one call each")
    for log2n in (12, 13, 14, 15, 16, 17, 18):
        n = 1 << log2n
        rows = {0: [], 3: []}
        for rnd in range(3):
            for mode in (0, 3):
                eng.set_eq1_short(mode)
                rows[mode].append(b2b(n, 30 if log2n <= 16 else 12))
                assert np.array_equal(oks[0][:n].cpu().numpy(), exp[:n])
        st = {}
        for mode in (0, 3):
            eng.set_eq1_short(mode); eng.set_stage_timing(True)
            call(n); torch.cuda.synchronize()
            st[mode] = [(k, round(x, 4)) for k, x in eng.last_stage_times()]
            eng.set_stage_timing(False)
        print(f"2^{log2n}: long {min(rows[0]):.4f} ms  short {min(rows[3]):.4f} ms  ({100 * (min(rows[3]) / min(rows[0]) - 1):+.1f} %)   rounds long {np.round(rows[0], 4).tolist()} short {np.round(rows[3], 4).tolist()}")
        print(f"       long  {eng and st[0]}\n       short {st[3]}")
    eng.set_eq1_short(1)
if "b" in what:
    print("(b) batches in flight: one lane / one stream vs two lanes / two streams (calls alternating), ms per call, min of 3 interleaved rounds")
    for log2n in (10, 12, 13, 14, 15, 16, 17, 18):
        n = 1 << log2n
        reps = 30 if log2n <= 16 else 12
        rows = {1: [], 2: []}
        for rnd in range(3):
            eng.set_in_flight(1)
            rows[1].append(b2b(n, reps))
            eng.set_in_flight(2)
            for _ in range(2):
                call(n, 0, streams[0]); call(n, 1, streams[1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call(n, 0, streams[0]); call(n, 1, streams[1])
            torch.cuda.synchronize()
            rows[2].append((time.perf_counter() - t0) / (2 * reps) * 1e3)
            assert np.array_equal(oks[1][:n].cpu().numpy(), exp[:n])
        eng.set_in_flight(1)
        print(f"2^{log2n}: one {min(rows[1]):.4f} ms  two in flight {min(rows[2]):.4f} ms per call ({100 * (min(rows[2]) / min(rows[1]) - 1):+.1f} %)   rounds one {np.round(rows[1], 4).tolist()} two {np.round(rows[2], 4).tolist()}")
eng.close()
