import sys; import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch, time
import zk_nullifier_sig_amd as plume
eng = plume.Engine(0); dev = torch.device("cuda:0")
n = 1 << 20
g = torch.Generator(device="cpu"); g.manual_seed(1)
nul = torch.randint(0, 256, (n, 64), dtype=torch.uint8, generator=g).to(dev)
nul[16::16] = nul[8::16][: nul[16::16].shape[0]]
first = torch.zeros(n, dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(5):
    eng.nullifier_first_occurrence_device(n, nul, None, None, first, cnt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): eng.nullifier_first_occurrence_device(n, nul, None, None, first, cnt)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) / 20 * 1e3, int(cnt.item()))
