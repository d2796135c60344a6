"""2^16-item device-resident verify (BASELINE config 2), one call after the other: stage times for different table jobs per lane (env PLUME_JOBS_PER_LANE; unset = the library's pick)."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import os, sys, pathlib, time, subprocess, json
ROOT = pathlib.Path(__file__).resolve().parents[2]
if len(sys.argv) > 1:
    sys.path.insert(0, str(ROOT))
    import numpy as np, torch
    import zk_nullifier_sig_amd as plume
    from tests import synth
    n = 1 << int(sys.argv[1])
    b = synth.sign_inputs(n)
    e = plume.Engine(0)
    ref = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, ref)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    acc, ts = {}, []
    for rep in range(24):
        t0 = time.perf_counter()
        e.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
        torch.cuda.synchronize()
        if rep >= 4:
            ts.append(time.perf_counter() - t0)
            for name, ms in e.last_stage_times():
                acc[name] = acc.get(name, 0.0) + ms / 20
    assert bool((ok.cpu().numpy() == synth.expected_ok(n)).all())
    print(json.dumps({"call_ms": round(sorted(ts)[len(ts) // 2] * 1e3, 4), **{k: round(x, 4) for k, x in acc.items()}}))
else:
    for lg in (12, 14, 16, 17, 18, 19):
        for small in (0, 3 << 20):
            env = dict(os.environ, PLUME_TABLES_SMALL_MAX=str(small))
            out = subprocess.run([sys.executable, __file__, str(lg)], env=env, capture_output=True, text=True)
            print(f"2^{lg} small-batch tables {'on' if small else 'off'}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
