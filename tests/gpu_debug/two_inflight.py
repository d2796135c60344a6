"""Round 3: one verify call after the other on one stream against TWO batches in flight (two contexts on two streams, calls alternating), per batch size.
    python tests/gpu_debug/two_inflight.py [log2 sizes ...]"""
import pathlib
import sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch  # noqa: E402
import bench  # noqa: E402
import zk_nullifier_sig_amd as plume  # noqa: E402
eng = plume.Engine(0)
for l in [int(x) for x in sys.argv[1:]] or [16, 18, 20]:
    r = bench.small_batch_entry(eng, torch.device("cuda:0"), log2n=l)
    t = r["two_batches_in_flight"]
    print(f"2^{l}: one call after the other {r['ms_per_batch']} ms = {r['items_per_s'] / 1e6:.2f} M/s; two batches in flight {t['ms_per_batch']} ms per batch = {t['items_per_s'] / 1e6:.2f} M/s")
