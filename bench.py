#!/usr/bin/env python3
"""bench.py — PLUME V1 verifies/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--config 2|3|4] [--log2-batch 20] [--version 1] [--multi-ctx]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the verify pipeline (ingest+h2c -> window tables -> multi-scalar loop -> finalize) over one batch of synthetic
signatures, inputs already resident in HBM, through the C ABI's device-resident entry point (plume_verify_batch_device) on torch's
current stream.  One process per GPU; the path has no cross-device exchange, so the only collectives are the timing barrier, the MAX
over ranks and the gather of the per-rank times (backend nccl = RCCL).
  --scaling weak   (default) every rank verifies its own 2^log2-batch items per step: items [rank*2^k, (rank+1)*2^k) of the BASELINE.md §3 generator
  --scaling strong 2^log2-batch items per step IN TOTAL, contiguous even split [floor(rT/W), floor((r+1)T/W)) over the ranks
  --config 4       BASELINE.json configs[3]: 2^22 V2 verifies in total, even split over the ranks (= --scaling strong --log2-batch 22 --version 2)
  --config 3       BASELINE.json configs[2]: the step is one 2^20 V1 SIGN pass (metric signs/s); --config 2: 2^16 V1 verify
  --multi-ctx      ONE process drives all --gpus N devices through one plume_init_multi context and the HOST-POINTER entry point (page-locked caller arrays,
                   the library shards the batch): the other way to use a node (LABNOTES.md §7).  PCIe-inclusive, so the number is reported as `e2e_multi_ctx`
                   beside a `value` that says so in `config.form`; the driver's --gpus N launch (one process per GPU, inputs resident) is the headline form

Objects on the line besides the contract's keys:
  roofline         the BINDING roof of the dominant kernel (k_verify_msm): integer VALU.  achieved = accounting 32-bit MACs of the two double-base
                   multiplications (4 160 Fp-mult x 72, SURVEY.md §8d, frozen in BASELINE.md §4) / that kernel's HIP-event duration; peak = the
                   v_mad_u64_u32 issue rate measured by a microbenchmark in this run (the spec-derived half-rate figure beside it)
  hbm_view         the same kernel against HBM: algorithmic bytes (353 B/item) / kernel time vs 8 TB/s, and the PMC traffic (table gathers)
  valu_roofline    the whole path against the same roof, plus the measured issue rates
  e2e_host_pinned  N=1: whole-call rates of the HOST-POINTER entry points with page-locked caller arrays (H2D + kernels + D2H), verify and sign
  cpu_baseline     the plain-C oracle (kind "port": rust-k256 cannot be built here) on ALL host cores, first 2^16 items of the same batch, and
                   beside it an optimised CPU leg (GLV + wNAF + dedicated squaring + one inversion per point set) so the oracle can stay simple
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the HIP runtime's hardware-queue pool: with the default 4 the two streams of the in-flight mode can land on ONE queue and run their kernels one after the other (round 6's trace;
# include/plume_hip.h, plume_set_in_flight).  Read at the process's first HIP call; reported in config.form.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Every context this script creates records the per-stage timing events (off by default in the library since 0.5: ~6 us of idle GPU each): roofline.kernel_ms is the
# multi-scalar kernel's duration by HIP events over the timed region, as the contract asks -- so `value` carries their cost (0.2 % at 2^20: five events of ~6 us in a
# step of 18 ms; the library's default, without them, is that much FASTER than `value`); the 2^16 entry also reports the default, event-free call
# (other_workloads.verify_v1_2p16.ms_per_batch).
os.environ.setdefault("PLUME_STAGE_TIMES", "1")
sys.path.insert(0, str(ROOT))

BYTES_PER_ITEM = {1: 353, 2: 225}                 # SURVEY.md §8d / BASELINE.md §4 (in + out)
FPMUL_PER_ITEM = {1: 5460, 2: 5460}               # accounting algorithm, whole verify
FPMUL_MSM_PER_ITEM = 1900 + 2260                  # the two double-base multiplications (dominant kernel)
FPMUL_SIGN_PER_ITEM = 4920                        # accounting algorithm, whole V1 sign
FPMUL_SIGN_HMUL_PER_ITEM = 3168                   # r*H, sk*H: the signer's dominant kernel
BYTES_PER_SIGN = 416
MACS_PER_FPMUL = 72
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8 TB/s spec
MAD_PEAK_REF = 3.5e13                             # v_mad_u64_u32 lane-ops/s this pool's boxes reach: 1024 SIMDs x 64 lanes / 1.83 ns (tests/gpu_debug/instr_rates_r01.txt).  Reported beside the
                                                  # run's own measurement for comparison with rounds 1-3, whose `peak` was max(measured, this); since round 4 `peak` IS the measurement
MAD_PEAK_SPEC = 256 * 4 * 16 * 2.4e9              # half rate of the 32-lane-per-clock VALU: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 3.93e13


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-batch", type=int, default=None, help="weak: items per GPU per step = 2^this; strong: items per step in total (default: the preset's, else 20 = BASELINE)")
    ap.add_argument("--version", type=int, default=1, choices=(1, 2))
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4), help="BASELINE.json preset: 2 = 2^16 V1 verify; 3 = 2^20 V1 SIGN; 4 = 2^22 V2 verify split over the ranks")
    ap.add_argument("--multi-ctx", action="store_true", help="one process, one plume_init_multi context over --gpus devices, host-pointer calls from page-locked arrays")
    ap.add_argument("--in-flight", type=int, default=2, choices=(1, 2, 3, 4),
                    help="batches in flight per GPU: step i goes to context i mod F on stream i mod F (F contexts, F streams).  1 = one call after the other on one stream, the mode "
                         "the per-kernel stage times and the roofline are measured in (with F > 1 a serial pass after the timed region supplies them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the issue-rate probe kernels (profiling runs: they would bury the trace); the roofline then prices against the pool's reference rate and says so")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads (V2 verify, V1 sign, SEC1 ingest, e2e) reported at N=1")
    a = ap.parse_args()
    a.workload = "verify"
    preset_log2 = 20
    if a.config == 4:
        a.scaling, preset_log2, a.version = "strong", 22, 2
    elif a.config == 2:
        preset_log2, a.version = 16, 1
    elif a.config == 3:
        preset_log2, a.version, a.workload = 20, 1, "sign"
    if a.log2_batch is None:                      # an explicit --log2-batch overrides the preset's size (smaller smoke runs of the same workload)
        a.log2_batch = preset_log2
    return a


def mad_peak_probe(eng, runs: int = 5, iters: int = 1 << 19):
    """The roof of `roofline`: the chip-wide v_mad_u64_u32 issue rate, measured in THIS run.  The library's probe kernel (k_microbench kind 0: 8 wavefronts per
    SIMD, 8 independent accumulator chains per lane, 2^19 x 8 multiply-adds per lane = ~60 ms per launch, a 16-iteration warm-up launch in front) is run `runs` times back to
    back after one untimed settling run; the MEDIAN is the peak, the spread is reported.  One rule for every box: no floor, no pool constant (rounds 1-3 took
    max(measured, 3.5e13), which let the same run read 0.52 / 0.58 / 0.62 depending on the box: VERDICT r3 weak #4)."""
    eng.microbench(0, iters)
    vals, ms = [], []
    for _ in range(runs):
        vals.append(eng.microbench(0, iters))
        ms.append(eng.microbench_ticks()[1])
    sv = sorted(vals)
    med = sv[len(sv) // 2]
    return {"median": med, "runs": [round(x, 1) for x in vals], "spread": round((sv[-1] - sv[0]) / med, 4), "ms_per_run": round(sorted(ms)[len(ms) // 2], 2),
            "probe": f"k_microbench kind 0: 8 waves/SIMD x 8 chains/lane x {iters} x 8 v_mad_u64_u32 per lane, median of {runs} after one settling run"}


def pmc_issue(kernel: str, build: str):
    """Executed instruction counts of `kernel` per 2^20-item launch from the newest committed counter summary (SQ_INSTS_VALU: wave-instructions) with the share of multiply-adds
    the ISA listing gives for the kernel's loop bodies (profiles/rNN_isa_mix.txt, weighted by how often each body runs) -- or None."""
    try:
        f = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"))[-1]
        j = json.loads(f.read_text())
        d = j[kernel]
        return {"file": f"profiles/{f.name}", "same_build": j.get("_build") == build, "valu_wave_insts": d["SQ_INSTS_VALU"], "waves": d.get("SQ_WAVES"),
                "mad_share": (j.get("_isa_mad_share") or {}).get(kernel), "clock_ghz": d.get("clock_ghz")}
    except Exception:
        return None


# frozen accounting per stage: Fp-multiplications per V1 / V2 verify, and the kernels each stage launches
STAGE_FPMUL = {"verify_ingest_h2c": 634, "verify_scalars": 0, "tables": 381, "verify_msm": FPMUL_MSM_PER_ITEM, "to_affine": 0, "verify_finalize": 282}
STAGE_KERNELS = {"verify_ingest_h2c": [("plume::k_verify_ingest", 1)], "verify_scalars": [("plume::k_verify_scalars", 1)],
                 "tables": [("plume::k_tab_pass_a", 1), ("plume::k_tab_pass_b", 1), ("plume::k_tab_invert", 1)],
                 "verify_msm": [("plume::k_verify_msm", 1)], "verify_finalize": [("plume::k_verify_finalize", 1)]}


def stage_roofline(stages: dict, n: int, mad_rate: float, build: str, short1: bool = False):
    """Every stage of the step against both roofs: the integer-VALU roof in the frozen accounting's multiply-adds (Fp-mult x 72 x items / stage time / this
    run's probe) and in ISSUED VALU instructions (the committed counter pass: SQ_INSTS_VALU x 64 lanes / time / probe -- every instruction priced as a multiply-add slot), and
    the HBM roof (counter bytes: 2 x FETCH_SIZE + WRITE_SIZE of the stage's kernels / time / 8 TB/s).  The counters are per 2^20-item launch of the committed summary
    (profiles/rNN_pmc_summary.json; `same_build` says whether they were collected from these kernels), scaled to this batch; the times are this run's."""
    try:
        f = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"))[-1]
        j = json.loads(f.read_text())
    except Exception:
        j, f = {}, None
    out = {"_source": f"profiles/{f.name}" if f else None, "_same_build": bool(j) and j.get("_build") == build,
           "_note": "valu_mac_frac: accounted multiply-adds / time / probe; valu_issue_frac: issued VALU wave-instructions x 64 / time / probe (1.0 = every issue slot taken by something "
                    "priced as a multiply-add; plain ops issue faster, so a saturated mixed kernel reads ~0.95-1.0); hbm_frac: (2 x FETCH_SIZE + WRITE_SIZE) / time / 8 TB/s"}
    scale = n / float(1 << 20)
    for name, ms in stages.items():
        if not ms or name not in STAGE_KERNELS:
            continue
        sec = ms * 1e-3
        e = {"ms": ms, "accounted_fp_mult_per_item": STAGE_FPMUL.get(name, 0)}
        if mad_rate and STAGE_FPMUL.get(name):
            e["valu_mac_frac"] = round(STAGE_FPMUL[name] * MACS_PER_FPMUL * n / sec / mad_rate, 4)
        insts = byts = 0.0
        have = True
        res = {}
        for k, times in STAGE_KERNELS[name]:
            if short1 and k == "plume::k_verify_msm":
                k = "plume::k_verify_msm_s"
            d = j.get(k)
            if not d or "SQ_INSTS_VALU" not in d:
                have = False
                break
            insts += times * d["SQ_INSTS_VALU"]
            byts += times * (FETCH_SIZE_FACTOR * d.get("FETCH_SIZE_KB_raw", 0.0) + d.get("WRITE_SIZE_KB_raw", 0.0)) * 1024
            res[k.split("::")[1]] = {"vgpr": d.get("_vgpr"), "scratch_B_per_lane": d.get("_scratch"), "lds_B": d.get("_lds"), "VALUBusy": d.get("VALUBusy")}
        if have:
            if mad_rate:
                e["valu_issue_frac"] = round(insts * 64 * scale / sec / mad_rate, 4)
            e["valu_insts_per_item"] = round(insts * 64 / float(1 << 20), 1)
            e["hbm_bytes_per_item"] = round(byts / float(1 << 20), 1)
            e["hbm_GBps"] = round(byts * scale / sec / 1e9, 1)
            e["hbm_frac"] = round(byts * scale / sec / 1e9 / HBM_PEAK_GBS, 4)
            e["bound"] = "hbm" if e["hbm_frac"] > e.get("valu_issue_frac", 0) else "int-valu issue"
            e["kernels"] = res
        out[name] = e
    return out


def shard_bounds(total: int, rank: int, world: int):
    """contiguous even split [floor(r*T/W), floor((r+1)*T/W)) — SURVEY.md §8e"""
    return (total * rank) // world, (total * (rank + 1)) // world


def plan(scaling: str, log2_batch: int, world: int, rank: int):
    """(items per step over all ranks, this rank's [start, stop)).  weak: every rank owns 2^log2_batch items; strong: 2^log2_batch items in total,
    contiguous even split (BASELINE config 4 = strong, 2^22, V2)"""
    total = (1 << log2_batch) * (world if scaling == "weak" else 1)
    start, stop = shard_bounds(total, rank, world)
    return total, start, stop


FETCH_SIZE_FACTOR = 2.0   # gfx950: FETCH_SIZE tallies a 128-byte request at 64 bytes (MI355X_MICROARCH.md, HBM).  The guide calibrated that on wide coalesced streams and asks for a
                          # calibration "in your own access pattern": profiles/r02_fetch_calibration.json (tests/gpu_debug/fetch_calibration.py, the library's table-gather probe:
                          # 63.9 bytes reported per gather of one 128-byte row) confirms the factor for the multi-scalar kernel's per-lane table gathers.  WRITE_SIZE is taken as reported.


def pmc_traffic(kernel: str, build: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_summary.json: separate --pmc runs of this same bench for FETCH_SIZE and
    WRITE_SIZE), corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2, calibrated on this access pattern) -- or None.  The counters cannot be read
    inside a timed run (gpurun refuses --pmc beside the traces a bench needs, and a counter pass serialises the kernels), so the line carries the bytes of the newest
    committed summary and says which library build they were collected from: `same_build` false means the kernels changed since and the figure is history, not this run."""
    try:
        f = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"))[-1]
        j = json.loads(f.read_text())
        d = j[kernel]
        src = {"file": f"profiles/{f.name}", "launch": "2^20 items", "collected_from_build": j.get("_build"), "this_build": build, "same_build": j.get("_build") == build}
        return int((FETCH_SIZE_FACTOR * d["FETCH_SIZE_KB_raw"] + d["WRITE_SIZE_KB_raw"]) * 1024), src
    except Exception:
        return None, None


def cpu_baseline_sign(version: int, cores: int, visible: int, quota):
    """config 3's CPU leg: the plain-C oracle's signer on all host cores over the first 2^15 items of the same batch, the optimised CPU signer beside it"""
    import numpy as np

    from tests import _cpu_fast as CF
    from tests import _oracle_c as OC
    from tests import synth
    n, n1 = 1 << 15, 1024
    b = synth.sign_inputs(n)
    t0 = time.perf_counter(); want = OC.sign_batch(version, b["msgs"], b["off"], b["sk"], b["r"], nthreads=cores); t_mt = time.perf_counter() - t0
    t0 = time.perf_counter(); OC.sign_batch(version, b["msgs"], b["off"][: n1 + 1], b["sk"][:n1], b["r"][:n1], nthreads=1); t_1 = time.perf_counter() - t0
    out = {"value": round(n / t_mt, 1), "unit": "signs/s", "cores": cores, "cores_visible": visible, "kind": "port", "single_thread_value": round(n1 / t_1, 1),
           "sample": f"first {n} items of the same synthetic V{version} batch, plain-C oracle signer (4x64-bit limbs, 4-bit window, no endomorphism, four generic scalar multiplications and one "
                     f"inversion per encoded point), {cores} threads (os.cpu_count() = {visible}, cgroup CPU quota = {quota if quota else 'none'}); single thread: {n1} items. "
                     f"rust-k256 itself cannot be built here (no rustc/cargo)."}
    try:
        t0 = time.perf_counter(); got = CF.sign_batch(version, b["msgs"], b["off"], b["sk"], b["r"], nthreads=cores); t_f = time.perf_counter() - t0
        t0 = time.perf_counter(); CF.sign_batch(version, b["msgs"], b["off"][: n1 + 1], b["sk"][:n1], b["r"][:n1], nthreads=1); t_f1 = time.perf_counter() - t0
        assert all(np.array_equal(got[k], want[k]) for k in got)
        out["optimized"] = {"value": round(n / t_f, 1), "unit": "signs/s", "cores": cores, "kind": "port (optimised)", "single_thread_value": round(n1 / t_f1, 1),
                            "sample": "same items; oracle/plume_cpu_fast.c: GLV + wNAF (width 8 for G, 5 for H), lazy 4x64-bit limbs, three shared inversions per signature; "
                                      "equal to the plain oracle's bytes on these items (asserted)"}
    except Exception as e:
        out["optimized"] = {"error": str(e)[:200]}
    return out


def cpu_baseline(version: int, sign: bool = False):
    """plain-C oracle on ALL host cores over the first 2^16 items of the same synthetic batch, one core beside it, and the
    optimised CPU leg (oracle/plume_cpu_fast.c) when it is built"""
    import numpy as np  # noqa: F401

    from tests import _oracle_c as OC
    from tests import synth
    visible = os.cpu_count() or 1
    cores, quota = visible, None
    try:                                       # the GPU box is a container: a cgroup CPU quota may grant far fewer cores than it shows
        q, per = (Path("/sys/fs/cgroup/cpu.max").read_text().split() + ["100000"])[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            if q > 0:
                quota = q / float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        except Exception:
            pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.999)))
    if sign:
        return cpu_baseline_sign(version, cores, visible, quota)
    n = 1 << 16
    b = synth.sign_inputs(n)
    signed = OC.sign_batch(version, b["msgs"], b["off"], b["sk"], b["r"], nthreads=cores)
    v = synth.corrupt_for_verify(version, b, signed)
    args = (version, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    t0 = time.perf_counter()
    ok_mt = OC.verify_batch(*args, nthreads=cores)
    t_mt = time.perf_counter() - t0
    n1 = 2048
    a1 = (version, v["msgs"], v["off"][: n1 + 1], v["pk"][:n1], v["nullifier"][:n1], v["c"][:n1], v["s"][:n1],
          v["r_point"][:n1] if version == 1 else None, v["hashed_to_curve_r"][:n1] if version == 1 else None)
    t0 = time.perf_counter()
    OC.verify_batch(*a1, nthreads=1)
    t_1 = time.perf_counter() - t0
    assert list(ok_mt) == list(synth.expected_ok(n))
    out = {"value": round(n / t_mt, 1), "unit": "verifies/s", "cores": cores, "cores_visible": visible, "kind": "port",
           "sample": f"first {n} items of the same synthetic V{version} batch (1/16 corrupted), plain-C oracle (4x64-bit limbs, 4-bit window, no endomorphism, "
                     f"one inversion per encoded point), {cores} threads (os.cpu_count() = {visible}, cgroup CPU quota = {quota if quota else 'none'}, effective parallelism "
                     f"measured = {n / t_mt / (n1 / t_1):.1f} cores); single thread ({n1} items): {n1 / t_1:.1f} verifies/s. "
                     f"rust-k256 itself cannot be built here (no rustc/cargo).",
           "single_thread_value": round(n1 / t_1, 1)}
    try:
        from tests import _cpu_fast as CF
        t0 = time.perf_counter()
        ok_f = CF.verify_batch(*args, nthreads=cores)
        t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        CF.verify_batch(*a1, nthreads=1)
        t_f1 = time.perf_counter() - t0
        assert list(ok_f) == list(ok_mt)
        out["optimized"] = {"value": round(n / t_f, 1), "unit": "verifies/s", "cores": cores, "kind": "port (optimised)", "single_thread_value": round(n1 / t_f1, 1),
                            "sample": "same items; oracle/plume_cpu_fast.c: GLV + width-5 wNAF, 4x64-bit limbs with a dedicated squaring, Jacobian mixed additions, "
                                      "inversion by addition chain, checked item by item against the plain oracle in tests/"}
    except Exception as e:  # the optimised leg is optional
        out["optimized"] = {"error": str(e)[:200]}
    try:                                                   # SURVEY.md §8d's optional third-party leg: OpenSSL's secp256k1 arithmetic under the oracle's hash_to_curve
        from tests import _openssl_leg as OL
        if OL.available():
            m = 1 << 13
            am = (version, v["msgs"], v["off"][: m + 1], v["pk"][:m], v["nullifier"][:m], v["c"][:m], v["s"][:m],
                  v["r_point"][:m] if version == 1 else None, v["hashed_to_curve_r"][:m] if version == 1 else None)
            t0 = time.perf_counter()
            ok_o = OL.verify_batch(*am, nthreads=cores)
            t_o = time.perf_counter() - t0
            q = 256
            a1o = (version, v["msgs"], v["off"][: q + 1], v["pk"][:q], v["nullifier"][:q], v["c"][:q], v["s"][:q],
                   v["r_point"][:q] if version == 1 else None, v["hashed_to_curve_r"][:q] if version == 1 else None)
            t0 = time.perf_counter()
            OL.verify_batch(*a1o, nthreads=1)
            t_o1 = time.perf_counter() - t0
            assert list(ok_o) == list(ok_mt[:m])
            out["openssl"] = {"value": round(m / t_o, 1), "unit": "verifies/s", "cores": cores, "kind": "third party (OpenSSL libcrypto EC_POINT_mul / add / cmp on NID_secp256k1; hash_to_curve and "
                              "SHA-256 from the oracle)", "single_thread_value": round(256 / t_o1, 1), "sample": f"first {m} items of the same batch; verdicts equal the oracle's (asserted)"}
    except Exception as e:
        out["openssl"] = {"error": str(e)[:200]}
    return out


_PROGRESS = {"section": None, "since": None}


def _at(section: str):
    """where the secondary sections are (the watchdog reports it should they stall)"""
    _PROGRESS["section"] = section
    _PROGRESS["since"] = time.time()
    if os.environ.get("PLUME_BENCH_PROGRESS"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {section}", file=sys.stderr, flush=True)


def e2e_host_pinned(eng, n, b, signed_v1):
    """whole-call rate of the host-pointer entry points with page-locked caller arrays: H2D + kernels + D2H, pipelined in pieces (SURVEY §8d 'secondary')"""
    import numpy as np

    from tests import synth
    from zk_nullifier_sig_amd import capi
    out = {}
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    so["status"] = capi.pinned_empty(n)
    eng.set_stage_timing(False)            # what a caller of the host-pointer entry points gets: the library's default, no timing events inside the pieces (restored below)

    def best(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts), sorted(ts)[len(ts) // 2]

    _at("e2e_host_pinned: sign_v1 from page-locked arrays")
    tb, tm = best(lambda: eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
    assert np.array_equal(so["s"], signed_v1["s"]) and not so["status"].any()
    out["sign_v1"] = {"items_per_s": round(n / tm, 1), "ms_per_call": round(tm * 1e3, 3), "best_ms": round(tb * 1e3, 3), "bytes_in": 96 * n, "bytes_out": 321 * n}
    v = synth.corrupt_for_verify(1, b, signed_v1)
    vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    ok = capi.pinned_empty(n)
    _at("e2e_host_pinned: verify_v1 from page-locked arrays")
    tb, tm = best(lambda: eng.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=ok))
    assert np.array_equal(ok, synth.expected_ok(n))
    out["verify_v1"] = {"items_per_s": round(n / tm, 1), "ms_per_call": round(tm * 1e3, 3), "best_ms": round(tb * 1e3, 3), "bytes_in": 352 * n, "bytes_out": n}
    # the same call with pageable caller arrays (the runtime stages them), for comparison
    _at("e2e_host_pinned: verify_v1 from pageable arrays")
    tb, tm = best(lambda: eng.verify_batch(1, v["msgs"], b["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"]), reps=2)
    out["verify_v1_pageable"] = {"items_per_s": round(n / tm, 1), "ms_per_call": round(tm * 1e3, 3)}
    # the same call through a plume_init_multi context of EIGHT shards that all sit on this one GPU (config 4's split in the library's own form), against the
    # one-context call above.  The GPU's work is the same; anything well under 1.0 is host-side serialisation (locks, staging allocation, thread wake-ups, eight pipelines'
    # launches) that eight real GPUs would meet too.  Each shard gets n/8 items = one piece, so nothing is pipelined INSIDE a shard: the eight shards overlap each other instead.
    try:
        import zk_nullifier_sig_amd as plume
        dev_id = eng.device_id if hasattr(eng, "device_id") else 0
        _at("e2e_host_pinned: creating the eight-shard context")
        m = plume.Engine([dev_id] * 8)
        try:
            m.set_stage_timing(False)
            ok8 = capi.pinned_empty(n)
            _at("e2e_host_pinned: verify_v1 through eight shards on this GPU")
            tb8, tm8 = best(lambda: m.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=ok8))
            assert np.array_equal(ok8, synth.expected_ok(n))
            out["verify_v1_eight_shards_on_this_gpu"] = {"items_per_s": round(n / tm8, 1), "ms_per_call": round(tm8 * 1e3, 3), "best_ms": round(tb8 * 1e3, 3), "shards": m.num_shards(),
                                                         "frac_of_one_context": round(tm / tm8, 4), "numa_nodes": m.shard_numa_nodes(),
                                                         "note": "plume_init_multi([d] * 8): eight worker threads, eight workspaces, 8 x 4 staging slots sharing ONE GPU; "
                                                                 "frac_of_one_context = this rate / verify_v1's (same arrays, same run)"}
        finally:
            _at("e2e_host_pinned: closing the eight-shard context")
            m.close()
    except Exception as e:
        out["verify_v1_eight_shards_on_this_gpu"] = {"error": str(e)[:300]}
    eng.set_stage_timing(True)
    out["note"] = ("median of 3 calls after one warm-up; stage-timing events off (the library's default); page-locked arrays from plume_host_alloc; verify: pieces of up to 2^19 items (first 2^16, then x3 per piece) alternating between two lanes of the context, four staging slots (round 4); "
                   "sign: uniform 2^16-item pieces dealt to the two lanes in turn (round 6; one lane with tapered pieces before); pageable arrays: one lane")
    return out


def extras(eng, dev, n, b, signed_v1):
    """secondary workloads of BASELINE.json's configs at N=1, device-resident, 2 timed passes each (not the headline metric)"""
    import numpy as np
    import torch

    from tests import _sec1, synth
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    out = {}

    def timed(fn, reps=2):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    msgs, off, sk, r = t(b["msgs"]), t(b["off"].view(np.int64)), t(b["sk"]), t(b["r"])
    mbytes = int(b["off"][-1])
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    _at("other_workloads: sign_v1 / sign_v2 device-resident")
    for ver in (1, 2):
        dt = timed(lambda: eng.sign_batch_device(ver, n, msgs, off, mbytes, sk, r, None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st))
        out[f"sign_v{ver}"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.last_stage_times()}}
    # V2 verify on the V2 signatures just produced (honest batch); the arkworks verification (verify_non_zk) of the same batch beside it
    _at("other_workloads: verify_v2")
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    dt = timed(lambda: eng.verify_batch_device(2, n, msgs, off, mbytes, o["pk"], o["nullifier"], o["c"], o["s"], None, None, ok))
    assert bool(ok.all())
    out["verify_v2"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.last_stage_times()}}
    _at("other_workloads: verify_non_zk_v2")
    dt = timed(lambda: eng.verify_non_zk_batch_device(2, n, msgs, off, mbytes, o["pk"], o["nullifier"], o["s"], o["r_point"], o["hashed_to_curve_r"], o["c"], ok))
    assert bool((ok == 1).all())
    out["verify_non_zk_v2"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.last_stage_times()}}
    # V1 verify with SEC1-compressed points (decompression on the GPU)
    if signed_v1 is not None:
        _at("other_workloads: verify_v1_sec1_compressed")
        c33 = {k: t(_sec1.compress(signed_v1[k])) for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r")}
        cc, ss = t(signed_v1["c"]), t(signed_v1["s"])
        dt = timed(lambda: eng.verify_batch_sec1_device(1, n, msgs, off, mbytes, c33["pk"], c33["nullifier"], cc, ss, c33["r_point"], c33["hashed_to_curve_r"], ok))
        assert bool(ok.all())
        out["verify_v1_sec1_compressed"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.last_stage_times()}}
    # aggregate random-linear-combination pre-filter over the V1 batch (all-or-nothing, probabilistic -- NOT the headline metric's semantics)
    if signed_v1 is not None:
        _at("other_workloads: aggregate_check_v1")
        s1 = {k: t(signed_v1[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
        rec = torch.zeros(72, dtype=torch.uint8, device=dev)
        seed = os.urandom(32)
        dt = timed(lambda: eng.aggregate_check_device(1, 0, n, msgs, off, mbytes, s1["pk"], s1["nullifier"], s1["c"], s1["s"], s1["r_point"], s1["hashed_to_curve_r"], seed, 0, None, rec), reps=3)
        assert int(rec[0].item()) == 1
        out["aggregate_check_v1"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in eng.last_stage_times()},
                                     "semantics": "all-or-nothing pre-filter: exact per-item hash check + one 5n-point multi-scalar multiplication (bucket method, 16-bit windows)"}
    # nullifier-set post-processing on the nullifiers just produced: first occurrences among 2^20 records, 1/16 of them
    # made repeats of an earlier item; HBM view = 66 algorithmic bytes per item (64-byte record + live flag in, first flag out)
    _at("other_workloads: nullifier_first_occurrence")
    nul = o["nullifier"].clone()
    nul[16::16] = nul[8::16][: nul[16::16].shape[0]]
    first = torch.zeros(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    dt = timed(lambda: eng.nullifier_first_occurrence_device(n, nul, ok, None, first, cnt), reps=5)
    assert int(cnt.item()) == n - (n - 1) // 16 and int(first.sum().item()) == int(cnt.item())
    out["nullifier_first_occurrence"] = {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 4), "algorithmic_GBps": round(66 * n / dt / 1e9, 1),
                                         "hbm_frac": round(66 * n / dt / 1e9 / HBM_PEAK_GBS, 4), "stage_ms": {k: round(v, 4) for k, v in eng.last_stage_times()}}
    return out


def small_batch_entry(eng, dev, log2n=16):
    """BASELINE config 2 (2^16 V1 verify, device-resident) as a secondary entry of the default line: ms per call and the stage times of one call"""
    import numpy as np
    import torch

    from tests import synth
    _at("small_batch_entry: setting up")
    n = 1 << log2n
    b = synth.sign_inputs(n)
    signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    d = {k: t(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off, ok = t(v["off"].view(np.int64)), torch.zeros(n, dtype=torch.uint8, device=dev)
    mb = int(v["off"][-1])
    fn = lambda: eng.verify_batch_device(1, n, d["msgs"], off, mb, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)  # noqa: E731
    reps = 20

    def timed():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    _at(f"small_batch_entry 2^{log2n}: warm-up calls")
    for _ in range(60):                     # the GPU has idled through the host-side set-up above: ~90 ms of the same calls bring its clocks back (a first block of 20 calls right
        fn()                                # after the set-up read 5 % slower than the next one: profiles/r05_bench_line.json of build c8ce7f21, before this warm-up existed)
    torch.cuda.synchronize()
    # three interleaved rounds of: the library's default -- no timing events between the kernels (plume_set_stage_timing; each costs ~6 us of idle GPU, 2 % of a call this
    # size) -- and the same calls with the stage events this script reads everywhere else; the median round of each is reported
    _at(f"small_batch_entry 2^{log2n}: timed rounds, one call after the other")
    offs, ons = [], []
    for _ in range(3):
        eng.set_stage_timing(False)
        try:
            offs.append(timed())
        finally:
            eng.set_stage_timing(True)
        ons.append(timed())
    dt, dt_ev = sorted(offs)[1], sorted(ons)[1]
    assert bool((ok.cpu() == torch.from_numpy(synth.expected_ok(n))).all())
    stages = {k: round(x, 4) for k, x in eng.last_stage_times()}
    # The same calls with TWO batches in flight: two lanes of the context (plume_set_in_flight) on two caller streams, calls alternating (GPU_MAX_HW_QUEUES=8, set at the top of this
    # script, so that the two streams sit on different hardware queues).  A 2^16 batch leaves most SIMDs two wavefronts: another call's kernels fit beside them.  A throughput figure for a
    # server holding several small batches, NOT the latency of one call, which is the entry above.  Stage-timing events off, like the default entry.
    _at(f"small_batch_entry 2^{log2n}: two batches in flight on two streams")
    ok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    fn2 = lambda o, st: eng.verify_batch_device(1, n, d["msgs"], off, mb, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], o, stream=st)  # noqa: E731
    torch.cuda.synchronize()
    eng.set_in_flight(2)
    eng.set_stage_timing(False)
    try:
        for _ in range(4):
            fn2(ok, s1); fn2(ok2, s2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn2(ok, s1); fn2(ok2, s2)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / (2 * reps)
    finally:
        eng.set_in_flight(1)
        eng.set_stage_timing(True)
    assert bool((ok2.cpu() == torch.from_numpy(synth.expected_ok(n))).all()) and bool((ok.cpu() == torch.from_numpy(synth.expected_ok(n))).all())
    return {"items_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 4), "ms_per_batch_with_stage_events": round(dt_ev * 1e3, 4), "stage_ms": stages,
            "stage_events": "ms_per_batch: the library's default, no timing events inside the call; stage_ms and ms_per_batch_with_stage_events: plume_set_stage_timing(1), as everywhere else in this line",
            "workload": (f"BASELINE.json configs[1]: " if log2n == 16 else "") + f"2^{log2n} V1 verifies per call, inputs resident in HBM, {reps} calls back to back, median of 3 rounds",
            "rounds_ms": {"default": [round(x * 1e3, 4) for x in offs], "with_stage_events": [round(x * 1e3, 4) for x in ons]},
            "msm_kernel": eng.last_msm_kernel(),
            "two_batches_in_flight": {"items_per_s": round(n / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 4),
                                      "note": "plume_set_in_flight(2), two caller streams on different hardware queues (GPU_MAX_HW_QUEUES=8), calls alternating: throughput with two small batches in "
                                              "flight, not one call's latency"}}


def multi_ctx_main(a):
    """--multi-ctx: ONE process, one plume_init_multi context over a.gpus devices, whole batches through the host-pointer entry point from page-locked arrays"""
    import numpy as np

    import zk_nullifier_sig_amd as plume
    from tests import synth
    from zk_nullifier_sig_amd import capi
    g, ver = a.gpus, a.version
    total = (1 << a.log2_batch) * (g if a.scaling == "weak" else 1)
    eng = plume.Engine(list(range(g)))
    b = synth.sign_inputs(total)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(ver, b, signed)
    keys = ("msgs", "pk", "nullifier", "c", "s") + (("r_point", "hashed_to_curve_r") if ver == 1 else ())
    vp = {k: capi.pinned_copy(v[k]) for k in keys}
    off = capi.pinned_copy(v["off"])
    ok = capi.pinned_empty(total)
    step = lambda: eng.verify_batch(ver, vp["msgs"], off, vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp.get("r_point"), vp.get("hashed_to_curve_r"), out=ok)  # noqa: E731
    for _ in range(a.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    elapsed = time.perf_counter() - t0                       # the host-pointer call returns when its results are in the caller's array
    assert np.array_equal(ok, synth.expected_ok(total)), "verify results differ from the expected corruption pattern"
    value = total * a.steps / elapsed
    line = {"metric": f"PLUME verifies/sec (secp256k1 V{ver}) at batch=2^{a.log2_batch}" + (" per GPU" if a.scaling == "weak" else " total") + ", host-pointer calls (PCIe-inclusive)",
            "value": round(value, 1), "unit": "verifies/s", "n_gpus": g, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{total} PLUME V{ver} verifies per step from page-locked HOST arrays ({BYTES_PER_ITEM[ver] - 1} B in, 1 B out per item over PCIe), 1/16 corrupted",
                       "form": "ONE process, one plume_init_multi context: the library splits every array evenly and contiguously over the devices, one worker thread + three streams per device; "
                               "NOT the inputs-resident form of the headline metric (bench.py --gpus N without --multi-ctx)",
                       "global_items_per_step": total, "parallelism": f"in-library shard x{g}, no collective", "shards": eng.num_shards()},
            "e2e_multi_ctx": {"items_per_s": round(value, 1), "ms_per_call": round(1e3 * elapsed / a.steps, 3), "bytes_in_per_call": (BYTES_PER_ITEM[ver] - 1) * total, "bytes_out_per_call": total}}
    print(json.dumps(line), flush=True)
    eng.close()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.multi_ctx:
        if world > 1:
            sys.exit("--multi-ctx is the one-process form: run it bare (python bench.py --gpus N --multi-ctx), not under torch.distributed.run")
        return multi_ctx_main(a)
    if a.gpus > 1 and world == 1:
        # convenience: launched bare with --gpus N -> re-launch under torch.distributed.run as a CHILD (nothing has touched the GPU yet)
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
               str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    import zk_nullifier_sig_amd as plume
    from tests import synth

    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (this library has no CPU fallback)")
    local_dev = local_rank % ndev          # a launcher that narrows every rank's view to its own GPU leaves each rank with device 0
    torch.cuda.set_device(local_dev)
    dev = torch.device(f"cuda:{local_dev}")
    use_dist = world > 1 or os.environ.get("PLUME_BENCH_FORCE_DIST") == "1"   # the knob runs the RCCL code path at world size 1 (tests/test_gpu_round2.py)
    # The only collectives are the timing barrier, the MAX and the gather of the per-rank records.  RCCL refuses two ranks on one device, so when the launcher gives this
    # node more ranks than GPUs (the one-GPU test box running `--gpus 2`: tests/test_gpu_round4.py) they go over gloo on host tensors instead; the data path has no
    # collective either way.
    timing_backend = os.environ.get("PLUME_BENCH_TIMING_BACKEND") or ("gloo" if world > ndev else "nccl")
    tdev = dev if timing_backend == "nccl" else torch.device("cpu")
    if use_dist:
        if timing_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    eng = plume.Engine(local_dev)
    ver, sign = a.version, a.workload == "sign"
    total, start, stop = plan(a.scaling, a.log2_batch, world, rank)             # items per step over all ranks, this rank's slice
    n = stop - start                                                           # this rank's items per step
    eng.set_chunk(max(n, 1 << 20))
    # F batches in flight: F lanes of the context (each with its own workspace) on F streams, steps dealt out in turn.  Every kernel of a 2^20 batch fills the chip, yet the
    # memory-bound table passes and the ramps / tails of one batch's kernels do fit beside the issue-bound multi-scalar kernel of another: 20.1 vs 20.8 ms per batch on one box
    # (tests/gpu_debug/two_inflight.py) -- with NO slicing cost, which is what sank the in-library sub-batch pipeline (LABNOTES.md §6).
    F = max(1, a.in_flight)
    eng.set_in_flight(F)                   # plume_set_in_flight: the context's device-resident calls go in turn to F lanes (workspace, streams, events each; fixed tables shared)
    engines = [eng] * F
    streams = [None] if F == 1 else [torch.cuda.Stream(device=dev) for _ in range(F)]

    # ---- synthetic shard of this rank, signed on the GPU (setup, untimed), corrupted 1/16 as BASELINE.md §3
    b = synth.sign_inputs(n, start=start)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    if sign:
        d = {k: t(b[k]) for k in ("msgs", "sk", "r")}
        d["off"] = t(b["off"].view(np.int64))
        outs = [{k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]} for _ in range(F)]
        statuses = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(F)]
        o, status = outs[0], statuses[0]
        msgs_bytes = int(b["off"][-1])

        def step(i=0):
            k = i % F
            engines[k].sign_batch_device(ver, n, d["msgs"], d["off"], msgs_bytes, d["sk"], d["r"], None, outs[k]["pk"], outs[k]["nullifier"], outs[k]["c"], outs[k]["s"], outs[k]["r_point"],
                                         outs[k]["hashed_to_curve_r"], statuses[k], stream=streams[k])
    else:
        v = synth.corrupt_for_verify(ver, b, signed, start=start)
        d = {k: t(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s")}
        d["off"] = t(v["off"].view(np.int64))
        d["r_point"] = t(v["r_point"]) if ver == 1 else None
        d["hashed_to_curve_r"] = t(v["hashed_to_curve_r"]) if ver == 1 else None
        oks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(F)]
        ok = oks[0]
        msgs_bytes = int(v["off"][-1])
        expected = torch.from_numpy(synth.expected_ok(n, start)).to(dev)

        def step(i=0):
            k = i % F
            engines[k].verify_batch_device(ver, n, d["msgs"], d["off"], msgs_bytes, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], oks[k], stream=streams[k])

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(F):                      # every lane allocates its workspace on its first call: keep that out of the timed region whatever --warmup is (untimed, not counted as warm-up)
        step(i)
    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    stage_acc = {}
    clk_acc = []                            # the multi-scalar kernel's shader clock, sampled inside the kernel (plume_last_msm_clock), one value per serial call
    for i in range(a.steps):
        step(i)
        if F == 1 and a.steps <= 64:  # per-stage HIP-event times (events are recorded on the launch stream inside the library; the library's default launch order is strictly serial)
            torch.cuda.current_stream().synchronize()
            for name, ms in eng.last_stage_times():
                stage_acc[name] = stage_acc.get(name, 0.0) + ms
            if not sign:
                g = eng.last_msm_clock_ghz()
                if g:
                    clk_acc.append(g)
    fence()
    elapsed_rank = time.perf_counter() - t0
    stage_div = a.steps
    in_flight_info = None
    if F > 1:
        # what the kernels of the timed region's LAST call took (its predecessor's multi-scalar kernel ran beside its first stages; its own multi-scalar kernel finishes with the
        # machine nearly to itself -- calls in the middle of the stream are stretched more, up to +0.4 ms on that kernel by tests with two contexts)
        infl = {name: ms for name, ms in eng.last_stage_times()}        # the lane of the last call
        eng.set_in_flight(1)                                             # the serial pass: one lane, one stream
        # ... and the serial pass the per-kernel figures come from: the same calls, one after the other on one stream
        S = max(2, min(a.steps, 5))
        step(0); torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(S):
            eng.verify_batch_device(ver, n, d["msgs"], d["off"], msgs_bytes, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok) if not sign else \
                eng.sign_batch_device(ver, n, d["msgs"], d["off"], msgs_bytes, d["sk"], d["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], status)
            torch.cuda.current_stream().synchronize()
            for name, ms in eng.last_stage_times():
                stage_acc[name] = stage_acc.get(name, 0.0) + ms
            if not sign:
                g = eng.last_msm_clock_ghz()
                if g:
                    clk_acc.append(g)
        serial_s = (time.perf_counter() - ts) / S
        stage_div = S
        in_flight_info = {"batches_in_flight": F, "stage_ms_in_flight": {k: round(x, 4) for k, x in infl.items()},
                          "serial": {"steps": S, "ms_per_step": round(serial_s * 1e3, 3), ("signs_per_s" if sign else "verifies_per_s"): round(n / serial_s, 1),
                                     "note": "the same calls one after the other on one stream, after the timed region: stage_ms and roofline.kernel_ms are this pass's"}}
    if sign:
        # the timed passes' outputs: equal to the setup pass's (host-pointer entry point) and, on a sample, to the CPU oracle's
        for oo, st_ in zip(outs, statuses):
            for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"):
                assert np.array_equal(oo[k].cpu().numpy(), signed[k]), f"sign outputs differ between passes: {k}"
            assert not bool(st_.any())
        if rank == 0:
            from tests import _oracle_c as OC
            m = 256
            want = OC.sign_batch(ver, b["msgs"], b["off"][: m + 1], b["sk"][:m], b["r"][:m], nthreads=8)
            assert all(np.array_equal(signed[k][:m], want[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")), "sign outputs differ from the CPU oracle"
    else:
        for okk in oks[: max(1, min(F, a.steps + a.warmup))]:
            assert bool((okk == expected).all()), "verify results differ from the expected corruption pattern"

    tmax = torch.tensor([elapsed_rank], dtype=torch.float64, device=tdev)
    per_rank = [elapsed_rank]
    # this rank's record: which device, which slice of the batch, and whether its verdicts were the corruption pattern (asserted above: a rank that got here passed)
    rank_rec = {"rank": rank, "local_rank": local_rank, "device": local_dev, "slice": [start, stop], "items_per_step": n, "elapsed_s": round(elapsed_rank, 6),
                ("outputs_match_setup_pass" if sign else "verdicts_match_pattern"): True}
    ranks = [rank_rec]
    if use_dist:
        gathered = [torch.zeros(1, dtype=torch.float64, device=tdev) for _ in range(world)]
        dist.all_gather(gathered, tmax.clone())
        per_rank = [float(g.item()) for g in gathered]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_rec)
    elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / a.steps
        value = total * a.steps / elapsed
        stages = {k: round(vv / stage_div, 4) for k, vv in stage_acc.items()}
        what = f"2^{a.log2_batch} per GPU" if a.scaling == "weak" else f"2^{a.log2_batch} in total, even split over {world} GPU(s)"
        op, unit = ("sign", "signs/s") if sign else ("verify", "verifies/s")
        cfg_idx = {None: 1, 2: 1, 3: 2, 4: 3}[a.config]
        line = {
            "metric": f"PLUME {'signs' if sign else 'verifies'}/sec (secp256k1 V{ver}) at batch=2^{a.log2_batch}" + (" per GPU" if a.scaling == "weak" else " total"), "value": round(value, 1), "unit": unit,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic", "world_size": world, "timing_backend": timing_backend if use_dist else None,
            "config": {"workload": f"BASELINE.json configs[{cfg_idx}]" + ("/metric" if cfg_idx == 1 and a.log2_batch == 20 else "") + f": batch {what}, PLUME V{ver} {op} (secp256k1 + SHA-256), 32-byte messages, "
                                   + ("" if sign else "1/16 corrupted, ") + "inputs resident in HBM; Fp arithmetic on 9x29-bit limbs through chains of v_mad_u64_u32 (32x32+64)",
                       "form": "one process per GPU (torch.distributed ranks), device-resident entry point plume_" + op + "_batch_device; " +
                               (f"{F} batches in flight per GPU (plume_set_in_flight): step i goes to lane i mod {F} of the context on stream i mod {F}, each call's launch order strictly serial (sub_batches = 1); "
                                f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} so that the streams sit on different hardware queues; "
                                "stage_ms and roofline from a serial pass after the timed region (in_flight.serial)" if F > 1 else
                                "one call after the other on torch's current stream, launch order strictly serial (sub_batches = 1)") + f"; library {eng.version()}",
                       "items_per_gpu": n, "global_items_per_step": total, "parallelism": f"shard x{world}, no collective on the data path",
                       "version": ver, "items_per_step_total": total,
                       "world_size": world, "collective_backend": ((("nccl (RCCL)" if timing_backend == "nccl" else "gloo (more ranks than GPUs on this node: RCCL refuses two ranks on one device)")
                                                                    + ": barrier + MAX + gather of the timings only") if world > 1 else None)},
            "ranks": ranks,
            "per_rank": {"items_per_step": [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)],
                         ("signs_per_s" if sign else "verifies_per_s"): [round((shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]) * a.steps / per_rank[r], 1) for r in range(world)]},
            "stage_ms": stages,
        }
        # the serial rate next to the headline: `value` is the timed region ({F} batches in flight); stage_ms / roofline.kernel_ms are one call after the other
        ser = in_flight_info["serial"] if in_flight_info else None
        line["value_serial"] = round(total / n * ser["signs_per_s" if sign else "verifies_per_s"], 1) if ser else line["value"]
        line["ms_per_step_serial"] = ser["ms_per_step"] if ser else line["ms_per_step"]
        if in_flight_info:
            line["in_flight"] = in_flight_info
        if stages:
            # the kernel the roofline is about is known beforehand: the multi-scalar kernel (k_verify_msm[_s]; the signer: k_sign_hmul), 75-78 % of a step at every batch size.
            # (Round 5: picking the LONGEST stage instead dropped `roofline` from a 2^16 line once, when a hiccup in front of the first kernel made that stage read longer.)
            want = "sign_hmul" if sign else "verify_msm"
            dom = want if stages.get(want) else max(stages, key=stages.get)
            dom_s = stages[dom] * 1e-3
            dom_parts = None
            if sign and dom == "sign_hmul" and stages.get("sign_hdbl"):
                # round 4: 64 of each multiplication's 128 doublings moved out of k_sign_hmul into k_sign_hdbl (2^64 H, once per item).  They belong to the accounted r*H, sk*H,
                # so the signer's roofline is priced on both launches -- k_sign_hmul alone would read too high against the frozen accounting
                dom_parts = {"sign_hmul": stages["sign_hmul"], "sign_hdbl": stages["sign_hdbl"]}
                dom_s += stages["sign_hdbl"] * 1e-3
            bytes_item = BYTES_PER_SIGN if sign else BYTES_PER_ITEM[ver]
            hbm_achieved = bytes_item * n / dom_s / 1e9
            # HBM bytes per launch from the committed PMC passes (2^20-item launch, scaled to this batch)
            ran = None if sign else eng.last_msm_kernel()                      # what the context's last verify call (the serial pass's) LAUNCHED: plume_last_msm_kernel
            short1 = ran == "k_verify_msm_s"                                   # equation 1 in its short form (csrc/plume_eis.h)
            dom_kernel = ran if (ran and dom == "verify_msm") else "k_" + dom
            tb, tsrc = pmc_traffic("plume::" + dom_kernel, eng.version())
            traffic_bytes = int(tb * (n / float(1 << 20))) if tb else None
            try:
                if a.no_probe:
                    raise RuntimeError("--no-probe")
                probe = mad_peak_probe(eng)
                mad_measured = probe["median"]
                add_rate = eng.microbench(4, 1 << 18)
                fpmul_rate = eng.microbench(5, 1 << 13)
                fpsqr_rate = eng.microbench(6, 1 << 13)
                other = {name: round(eng.microbench(k, 1 << 17), 1) for k, name in
                         ((1, "v_addc_co_u32"), (2, "v_mul_lo_u32"), (3, "v_mad_u32_u24"), (7, "v_fma_f64"), (8, "v_lshl_add_u64"))}
            except Exception as e:  # measurement extras must not kill the bench line
                probe, mad_measured, add_rate, fpmul_rate, fpsqr_rate, other = None, None, None, None, None, {"error": str(e)}
            # ONE denominator: the peak is this run's own measurement (median of five >= 50 ms probe launches), as SURVEY.md §8d prescribes; the pool's reference
            # rate and the clock-independent spec figure are printed beside it, never substituted for it.  Only a failed probe falls back to the reference (and says so).
            mad_rate = mad_measured or MAD_PEAK_REF
            msm_ms = stages.get("sign_hmul" if sign else "verify_msm")
            dom_fpmul = {"verify_msm": FPMUL_MSM_PER_ITEM, "sign_hmul": FPMUL_SIGN_HMUL_PER_ITEM}.get(dom)
            if dom_fpmul:
                msm = dom_fpmul * MACS_PER_FPMUL * n / dom_s
                line["roofline"] = {"bound": "int-valu", "kernel": dom_kernel + (" + k_sign_hdbl" if dom_parts else ""), "kernel_ms": round(dom_s * 1e3, 4),
                                    **({"kernel_ms_parts": dom_parts} if dom_parts else {}), "achieved": round(msm, 1), "peak": round(mad_rate, 1),
                                    "unit": "32-bit MAC/s", "frac": round(msm / mad_rate, 4),
                                    "peak_rule": "this run's v_mad_u64_u32 probe (median)" if mad_measured else "no probe in this run (--no-probe or a failed probe): the pool's reference rate",
                                    "peak_probe": probe, "peak_ref_pool": MAD_PEAK_REF, "frac_of_ref_pool": round(msm / MAD_PEAK_REF, 4),
                                    "peak_spec_half_rate": MAD_PEAK_SPEC, "frac_of_spec_half_rate": round(msm / MAD_PEAK_SPEC, 4),
                                    "traffic": traffic_bytes, "traffic_source": tsrc,
                                    "traffic_unit": "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE; the factor 2 of gfx950's FETCH_SIZE calibrated on this kernel's access pattern, "
                                                    "profiles/r02_fetch_calibration.json); read from the committed counter passes, not measured in this run: see traffic_source.same_build",
                                    "equation1_form": ("short (csrc/plume_eis.h): k G - upsilon pk - (tau - 1) R along 64 doublings + 15 comb additions; the accounting below stays the frozen "
                                                       "long-form count, so `achieved` prices the work the REFERENCE's equation asks for, not the instructions executed") if short1 else
                                                      ("long" if not sign else None),
                                    "accounting": (f"{dom_fpmul} Fp-mult per {op} in this kernel " + ("(r*H, sk*H: 2 x 1584)" if sign else "(s*G - c*pk: 1900, s*H - c*nul: 2260)") +
                                                   f" x {MACS_PER_FPMUL} MACs x {n} items per launch (SURVEY.md §8d, frozen in BASELINE.md §4); the 9x29-limb code issues 111 multiply-adds per "
                                                   f"Fp-mult (81 products + 22 fold + 8 column hand-offs), 75 per squaring; the accounting stays on the frozen 72")}
                # what the SIMDs actually issued (committed counter pass of the same build): all VALU slots and the multiply-adds among them, against the same peak.  `frac` counts
                # the ACCOUNTING's multiply-adds; these two say how full the issue ports are and how many multiply-adds the code spends per accounted one.
                line["roofline"]["frac_is"] = ("reference-equivalent work: the frozen accounting's multiply-adds (what the reference's two equations ask for) over kernel time over the probe's "
                                               "rate -- NOT the share of issue slots taken (issue_frac.valu_slots) nor the multiply-adds executed (issue_frac.executed_macs)")
                if clk_acc:
                    # box-independent: the kernel's duration in shader cycles.  The clock is sampled INSIDE the kernel in this run (one workgroup in 32 adds its lifetime in shader cycles
                    # and in constant-rate wall-clock ticks to two counters: plume_last_msm_clock), so a box that runs the kernel at a lower clock shows the same cycle count
                    ghz = sorted(clk_acc)[len(clk_acc) // 2]
                    line["roofline"]["clock_ghz_in_kernel_this_run"] = round(ghz, 4)
                    line["roofline"]["cycles_per_item"] = round(dom_s * ghz * 1e9 / n, 3)
                    line["roofline"]["simd_cycles_per_item"] = round(dom_s * ghz * 1e9 * 1024 / n, 1)
                    line["roofline"]["peak_at_this_runs_kernel_clock"] = round(256 * 4 * 16 * ghz * 1e9, 1)
                    line["roofline"]["frac_at_this_runs_kernel_clock"] = round(msm / (256 * 4 * 16 * ghz * 1e9), 4)
                    line["roofline"]["cycles_note"] = ("cycles_per_item = kernel_ms x the kernel's own shader clock / items (chip-wide cycles per verify); simd_cycles_per_item = x 1024 SIMDs: compare with "
                                                       "issue_frac.valu_insts_per_item x 4 cycles per multiply-add-class wave instruction / 64 lanes")
                iss = pmc_issue("plume::" + dom_kernel, eng.version())
                if iss and iss.get("valu_wave_insts") and dom_parts:
                    iss2 = pmc_issue("plume::k_sign_hdbl", eng.version())        # the signer's roofline spans both launches: so do its instruction counts
                    if iss2 and iss2.get("valu_wave_insts"):
                        iss["valu_wave_insts"] += iss2["valu_wave_insts"]
                    else:
                        iss = None
                if iss and iss.get("valu_wave_insts"):
                    lane_ops = iss["valu_wave_insts"] * 64.0 * (n / float(1 << 20))
                    share = iss.get("mad_share")
                    line["roofline"]["issue_frac"] = {
                        "valu_slots": round(lane_ops / dom_s / mad_rate, 4),
                        "executed_macs": round(lane_ops * share / dom_s / mad_rate, 4) if share else None,
                        "executed_macs_per_accounted_mac": round(lane_ops * share / (dom_fpmul * MACS_PER_FPMUL * n), 3) if share else None,
                        "valu_wave_insts_per_launch": iss["valu_wave_insts"], "valu_insts_per_lane": round(iss["valu_wave_insts"] / iss["waves"], 1) if iss.get("waves") else None,
                        "valu_insts_per_item": round(iss["valu_wave_insts"] * 64.0 / float(1 << 20), 1),
                        "mad_share_of_valu": share, "clock_ghz_in_kernel": iss.get("clock_ghz"),
                        "peak_at_kernel_clock": round(256 * 4 * 16 * iss["clock_ghz"] * 1e9, 1) if iss.get("clock_ghz") else None,
                        "frac_at_kernel_clock": round(msm / (256 * 4 * 16 * iss["clock_ghz"] * 1e9), 4) if iss.get("clock_ghz") else None,
                        "source": iss["file"], "same_build": iss["same_build"],
                        "clock_note": "clock_ghz_in_kernel = GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the committed counter pass: the multi-scalar kernel runs at ~2.07 GHz (power), the "
                                      "multiply-add probe that sets `peak` at ~2.2 GHz; peak_at_kernel_clock = 1024 SIMDs x 16 lanes x that clock",
                        "note": "valu_slots: every VALU instruction priced as one multiply-add slot (plain VOP1/VOP2 ops issue in about 0.57 of one: tests/gpu_debug/instr_rates_r03.txt), "
                                "so a kernel that saturates the SIMDs with a mix reads a little above the measured busy fraction"}
            if dom_fpmul and in_flight_info and in_flight_info["stage_ms_in_flight"].get(dom):
                kin = in_flight_info["stage_ms_in_flight"][dom] + (in_flight_info["stage_ms_in_flight"].get("sign_hdbl", 0.0) if dom_parts else 0.0)
                line["roofline"]["kernel_ms_in_timed_region"] = kin
                line["roofline"]["frac_in_timed_region"] = round(dom_fpmul * MACS_PER_FPMUL * n / (kin * 1e-3) / mad_rate, 4)
                line["roofline"]["note"] = (f"kernel_ms / frac: the serial pass (one call after the other).  In the timed region {F} batches are in flight and the kernels of the streams share the "
                                            "SIMDs: launches are stretched while the step gets shorter (kernel_ms_in_timed_region: the LAST call's launch, the one figure the library's stage events of a lane keep)")
            if not sign:
                try:
                    line["stage_roofline"] = stage_roofline(stages, n, mad_rate if mad_measured else None, eng.version(), short1)
                except Exception as e:
                    line["stage_roofline"] = {"error": str(e)[:200]}
            line["hbm_view"] = {"bound": "hbm", "kernel": dom_kernel, "kernel_ms": stages[dom], "achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(hbm_achieved / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": bytes_item * n,
                                "traffic": round(traffic_bytes / dom_s / 1e9, 1) if traffic_bytes else None, "traffic_bytes_per_launch": traffic_bytes,
                                "traffic_over_algorithmic": round(traffic_bytes / (bytes_item * n), 1) if traffic_bytes else None,
                                "traffic_source": tsrc,
                                "traffic_note": ("2 x FETCH_SIZE + WRITE_SIZE: the per-lane gathers of the HBM-resident window tables (one 128-byte row per table addition, counted by gfx950's "
                                                 "FETCH_SIZE as 64 bytes: calibrated with the library's gather probe, profiles/r02_fetch_calibration.json)") if tb else None,
                                "note": "the path is integer-VALU bound (SURVEY.md §8d): see roofline"}
            if mad_measured:
                step_s = sum(stages.values()) * 1e-3
                whole = (FPMUL_SIGN_PER_ITEM if sign else FPMUL_PER_ITEM[ver]) * MACS_PER_FPMUL * n / step_s
                line["valu_roofline"] = {"bound": "int-valu", "unit": "32-bit MAC/s", "peak_v_mad_u64_u32": round(mad_rate, 1), "peak_v_mad_u64_u32_measured_this_run": round(mad_measured, 1),
                                         "peak_spec_half_rate": MAD_PEAK_SPEC, "peak_v_add_u32": round(add_rate, 1),
                                         "fp_mul_per_s": round(fpmul_rate, 1), "fp_sqr_per_s": round(fpsqr_rate, 1), "other_issue_rates_per_s": other,
                                         "achieved_whole_path": round(whole, 1), "frac_whole_path": round(whole / mad_rate, 4),
                                         "msm_kernel_ms": msm_ms,
                                         "accounting": f"{FPMUL_SIGN_PER_ITEM if sign else FPMUL_PER_ITEM[ver]} Fp-mult/{op} x 72 MACs (SURVEY.md §8d, frozen in BASELINE.md §4)"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(ver, sign)
            except Exception as e:
                line["cpu_baseline"] = {"error": str(e)}
        # Everything the contract asks for is in `line` now.  The secondary sections below (other workloads, end-to-end host calls) add to it -- under a watchdog: should one of
        # them ever stall (a GPU call that never returns cannot be interrupted from Python), the line is printed as it stands, says so, and the process ends, instead of the
        # headline being lost with it.  (Round 6: one bench run on one box of the pool sat in a secondary section until its 900 s timeout; five later runs of the same command on other
        # boxes took 12 s each and the cause was never seen again.)
        import threading
        done = threading.Event()

        def _watchdog():
            if not done.wait(float(os.environ.get("PLUME_BENCH_SECONDARY_TIMEOUT", "420"))):
                line["watchdog"] = "the secondary sections (other_workloads / e2e_host_pinned) did not finish in time: this line carries what was measured before them"
                line["watchdog_stalled_in"] = {"section": _PROGRESS["section"], "for_s": round(time.time() - (_PROGRESS["since"] or time.time()), 1)}
                try:                                             # ... and every thread's Python stack on stderr (the stalled call is a ctypes call into the library or a torch synchronize)
                    import faulthandler
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                except Exception:
                    pass
                for _ in range(5):                               # (the main thread may be adding a key just now)
                    try:
                        out = json.dumps(dict(line), default=str)
                        break
                    except RuntimeError:
                        time.sleep(0.05)
                print(out, flush=True)
                os._exit(0)
        threading.Thread(target=_watchdog, daemon=True).start()
        if world == 1 and not a.no_extras and a.scaling == "weak" and a.log2_batch == 20 and not sign:
            try:
                line["other_workloads"] = extras(eng, dev, n, b, signed if ver == 1 else None)
            except Exception as e:
                line["other_workloads"] = {"error": str(e)}
            try:
                line["other_workloads"]["verify_v1_2p16"] = small_batch_entry(eng, dev, 16)
            except Exception as e:
                line["other_workloads"]["verify_v1_2p16"] = {"error": str(e)}
            try:        # a call the machine is nearly empty under: the multi-scalar stage runs as two half chains per task there (k_verify_msm_pair, calls of <= 2^14 items)
                line["other_workloads"]["verify_v1_2p14"] = small_batch_entry(eng, dev, 14)
            except Exception as e:
                line["other_workloads"]["verify_v1_2p14"] = {"error": str(e)}
            if ver == 1:
                try:
                    line["e2e_host_pinned"] = e2e_host_pinned(eng, n, b, signed)
                    # the host-pointer call against the same run's device-resident serial rate (VERDICT r3 #2 asks for >= 0.95; the floor analysis is in DESIGN.md §6)
                    line["e2e_host_pinned"]["verify_v1"]["frac_of_value_serial"] = round(line["e2e_host_pinned"]["verify_v1"]["items_per_s"] / line["value_serial"], 4)
                    sv = (line.get("other_workloads") or {}).get("sign_v1", {}).get("items_per_s")
                    if sv:
                        line["e2e_host_pinned"]["sign_v1"]["frac_of_device_resident_sign"] = round(line["e2e_host_pinned"]["sign_v1"]["items_per_s"] / sv, 4)
                except Exception as e:
                    line["e2e_host_pinned"] = {"error": str(e)}
                finally:
                    eng.set_stage_timing(True)
        done.set()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
