"""Calibration of rocprofv3's FETCH_SIZE for the multi-scalar kernel's access pattern (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access
pattern before trusting an absolute").  Runs a 2^20-item verify (which leaves the 3 GiB window-table buffer on the context) and then the library's table-gather probe
(plume_microbench kind 9: every lane gathers the five 16-byte quads of one 128-byte table row, pseudo-random rows, a known number of times).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal -o cal -- python3 tests/gpu_debug/fetch_calibration.py
    python3 tests/gpu_debug/fetch_calibration.py --reduce gpurun_out/pmc_cal      (prints bytes reported per gather)
"""
import csv, glob, json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
ITERS = 256


def run():
    import numpy as np
    import zk_nullifier_sig_amd as plume
    from tests import synth
    eng = plume.Engine(0)
    n = 1 << 20
    b = synth.sign_inputs(n)
    s = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    ok = eng.verify_batch(1, b["msgs"], b["off"], s["pk"], s["nullifier"], s["c"], s["s"], s["r_point"], s["hashed_to_curve_r"])
    assert bool(np.all(ok == 1))
    rate = eng.microbench(9, ITERS)
    _, ms = eng.microbench_ticks()
    gathers = rate * ms * 1e-3
    print(json.dumps({"probe": "k_gather_probe", "iters": ITERS, "gathers": round(gathers), "ms": round(ms, 3), "gathers_per_s": rate,
                      "requested_GB_per_s": round(rate * 80 / 1e9, 1), "lines_GB_per_s": round(rate * 128 / 1e9, 1)}))


def reduce(d):
    rows = []
    for f in glob.glob(str(pathlib.Path(d) / "**" / "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    out = {}
    for r in rows:
        if "k_gather_probe" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            lanes = int(r["Grid_Size"])
            out = {"FETCH_SIZE_KB": float(r["Counter_Value"]), "lanes": lanes, "gathers": lanes * ITERS}
    if not out:
        sys.exit("no k_gather_probe row found")
    out["reported_bytes_per_gather"] = round(out["FETCH_SIZE_KB"] * 1024 / out["gathers"], 2)
    out["note"] = "a gather requests 80 bytes (5 x 16) of one 128-byte row; rows are pseudo-random over a 3 GiB buffer (no reuse)"
    # the multi-scalar kernel's own count, if the same directory tree holds it
    print(json.dumps(out))


if __name__ == "__main__":
    reduce(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "--reduce" else run()
