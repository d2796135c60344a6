"""What cutting a 2^20-item verify into the host pipeline's pieces costs WITHOUT the copies: the same pieces as device-resident calls dealt to two lanes on two streams."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import zk_nullifier_sig_amd as plume
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
ref = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, ref)
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
offs = v["off"]
okd = torch.zeros(n, dtype=torch.uint8, device=dev)
st = [torch.cuda.Stream(device=dev) for _ in range(2)]


def pieces(sched, lanes):
    e.set_in_flight(lanes)
    offt = []
    i0 = 0
    for c in sched:
        rel = (offs[i0:i0 + c + 1] - offs[i0]).astype(np.uint64)
        offt.append((i0, c, torch.from_numpy(rel.view(np.int64)).to(dev), int(offs[i0])))
        i0 += c
    assert i0 == n

    def run():
        for k, (i0, c, ot, mb) in enumerate(offt):
            e.verify_batch_device(1, c, t["msgs"][mb:], ot, int(offs[i0 + c] - offs[i0]), t["pk"][i0:], t["nullifier"][i0:], t["c"][i0:], t["s"][i0:], t["r_point"][i0:],
                                  t["hashed_to_curve_r"][i0:], okd[i0:], stream=st[k % lanes])
        torch.cuda.synchronize()
    run(); run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    assert bool((okd.cpu().numpy() == synth.expected_ok(n)).all())
    return sorted(ts)[2] * 1e3


for name, sched in [("one piece", [n]), ("default 64k,192k,512k,256k", [65536, 196608, 524288, 262144]), ("8 x 128k", [131072] * 8), ("16 x 64k", [65536] * 16), ("4 x 256k", [262144] * 4),
                    ("64k,192k,256k x3", [65536, 196608, 262144, 262144, 262144]), ("2 x 512k", [524288] * 2),
                    ("64k,64k,128k,256k,512k", [65536, 65536, 131072, 262144, 524288]), ("64k,64k,128k,256k x3", [65536, 65536, 131072, 262144, 262144, 262144]),
                    ("128k,128k,256k,512k", [131072, 131072, 262144, 524288]), ("64k,192k,256k,512k", [65536, 196608, 262144, 524288]), ("32k,32k,64k,128k,256k,512k", [32768, 32768, 65536, 131072, 262144, 524288]),
                    ("64k,128k,320k,512k", [65536, 131072, 327680, 524288]), ("64k,64k,128k,256k,256k,256k r", [65536, 65536, 131072, 262144, 262144, 262144][::-1])]:
    print(f"{name:32s} one lane {pieces(sched, 1):6.2f} ms   two lanes {pieces(sched, 2):6.2f} ms", flush=True)
