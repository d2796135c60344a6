#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes under gpurun_out/pmc_{sq,sq2,fetch,write}/ into profiles/<round>_pmc_summary.json.
usage: python profiles/summarize_pmc.py r01"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[name]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, c in (("_vgpr", "VGPR_Count"), ("_sgpr", "SGPR_Count"), ("_lds", "LDS_Block_Size"), ("_scratch", "Scratch_Size")):
            agg[name][k] = [int(r[c])]
    return agg


rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
for tag in ("sq", "sq2"):
    for k, v in load(f"gpurun_out/pmc_{tag}/{tag}_counter_collection.csv").items():
        if not (k.startswith("plume::k_verify") or k.startswith("plume::k_tab")):
            continue
        d = out.setdefault(k, {})
        for c, x in v.items():
            if c == "_dur_ns":
                d.setdefault("_dur_ns_max", max(x))
            else:
                d[c] = max(x) if k.startswith("plume::k_tab") else sum(x) / len(x)      # the table kernels also run for the setup signer's smaller launches: the verify launch is the largest
# the signer's own counter pass (bench.py --config 3; collect.sh's pmc_sqs): its kernels only -- the table kernels keep the verify run's figures.  The run's launches differ in
# size (the setup, the piece pipeline of the extras): the 2^20-item launch is the one with the most wavefronts.
try:
    for k, v in load("gpurun_out/pmc_sqs/sqs_counter_collection.csv").items():
        if not k.startswith("plume::k_sign"):
            continue
        d = out.setdefault(k, {})
        big = max(range(len(v["SQ_WAVES"])), key=lambda i: v["SQ_WAVES"][i]) if v.get("SQ_WAVES") else 0
        for c, x in v.items():
            if c == "_dur_ns":
                d["_dur_ns_max"] = max(x)
            elif c.startswith("_"):
                d[c] = x[0]
            else:
                d[c] = x[big] if big < len(x) else max(x)
except OSError:
    pass
fe, wr = load("gpurun_out/pmc_fetch/fetch_counter_collection.csv"), load("gpurun_out/pmc_write/write_counter_collection.csv")
for k in out:
    if k in fe:
        out[k]["FETCH_SIZE_KB_raw"] = max(fe[k]["FETCH_SIZE"])
    if k in wr:
        out[k]["WRITE_SIZE_KB_raw"] = max(wr[k]["WRITE_SIZE"])
for k, d in out.items():
    if isinstance(d, dict) and "SQ_ACTIVE_INST_VALU" in d and d.get("GRBM_GUI_ACTIVE"):
        # rocprof's derived VALUBusy: 4 cycles per wave64 VALU instruction, 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        d["VALUBusy"] = round(d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8), 3)
for k, d in out.items():
    if isinstance(d, dict) and d.get("GRBM_GUI_ACTIVE") and d.get("_dur_ns_max"):
        d["clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / 8 / d["_dur_ns_max"], 3)       # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the counter pass's own launch duration


def isa_mad_share(path):
    """share of v_mad_u64_u32 among the VALU instructions the multi-scalar kernels execute, from the ISA mix of their loop bodies (profiles/isa_mix.py output): the two largest
    blocks of a kernel's section are the mixed-addition body and the doubling body; a verify lane runs 128 doublings and ~98 additions (78 / 132 slots per equation, 15/16 non-zero),
    a signer lane (round 4: chains of 64 doublings) 64 doublings and ~64 additions"""
    share, cur, blocks = {}, None, {}
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return share
    for ln in lines + ["# end: 0 basic blocks"]:
        if ln.startswith("# ") and "basic blocks" in ln:
            if cur and len(blocks) >= 2:
                top = sorted(blocks.values(), key=lambda t: -t[0])[:2]          # (mad64, VALU): addition body first, doubling body second
                # k_verify_msm_s (round 5): equation 1's lanes walk 64 doublings and ~79 additions (4 x 17 slots, 15/16 non-zero, + 15 from the comb), equation 2's 128 and ~124
                w_add, w_dbl = (64.0, 64.0) if "sign" in cur else ((101.5, 96.0) if cur.endswith("_msm_s") else (98.5, 128.0))
                mad = w_add * top[0][0] + w_dbl * top[1][0]
                valu = w_add * top[0][1] + w_dbl * top[1][1]
                share["plume::" + cur] = round(mad / valu, 4)
            cur, blocks = ln[2:].split(":")[0].strip(), {}
        elif cur and ln[:1] in "._e" and len(ln.split()) >= 12 and ln.split()[1].isdigit():
            f = ln.split()
            blocks[f[0]] = (int(f[1]), int(f[-2]))
    return share


out["_isa_mad_share"] = isa_mad_share(f"profiles/{rnd}_isa_mix.txt")
try:
    out["_build"] = open(f"gpurun_out/build_{rnd}.txt").read().strip()      # plume_version() of the library the counters were collected from (bench.py compares it with its own)
except OSError:
    pass
out["_notes"] = {
    "collection": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv, separate passes (sq, sq2, FETCH_SIZE, WRITE_SIZE); "
                  "command: python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras",
    "units": "FETCH_SIZE / WRITE_SIZE in KB as reported (raw). MI355X_MICROARCH.md: FETCH_SIZE tallies a 128-byte request at 64 bytes on gfx950 (x2 for wide coalesced streams); "
             "calibrated for the multi-scalar kernel's access pattern (five 16-byte quads of one 128-byte table row per lane) in profiles/r02_fetch_calibration.json: 63.9 bytes "
             "reported per row, i.e. the same factor 2, which bench.py applies; WRITE_SIZE is uncalibrated and taken as reported",
    "k_tab*": "max over launches (the setup signer launches the table passes with a third of the jobs)"}
json.dump(out, open(f"profiles/{rnd}_pmc_summary.json", "w"), indent=1)
for k, d in out.items():
    if not k.startswith("_") and isinstance(d, dict):
        print(k, {c: "%.3g" % x for c, x in sorted(d.items()) if c in ("FETCH_SIZE_KB_raw", "WRITE_SIZE_KB_raw", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE", "_vgpr", "_scratch", "_dur_ns_max")})
