"""Round 3: the host-pointer entry points (page-locked caller arrays, H2D + kernels + D2H) through ONE context on device 0 against multi-shard contexts whose shards all sit
on device 0 (plume_init_multi([0, 0]), [0, 0, 0]): the shards' pipelines run side by side, i.e. two or three pieces of the batch are in flight on the GPU.
    python tests/gpu_debug/e2e_shards_on_one_device.py [log2n]"""
import pathlib
import sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
b = synth.sign_inputs(n)
e0 = plume.Engine(0)
signed = e0.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
for ids in (0, [0, 0], [0, 0, 0]):
    eng = e0 if ids == 0 else plume.Engine(ids)
    r = bench.e2e_host_pinned(eng, n, b, signed)
    print(ids, "verify", r["verify_v1"]["ms_per_call"], "ms =", round(r["verify_v1"]["items_per_s"] / 1e6, 2), "M/s; sign", r["sign_v1"]["ms_per_call"], "ms =", round(r["sign_v1"]["items_per_s"] / 1e6, 2), "M/s", flush=True)
    if ids != 0:
        eng.close()
