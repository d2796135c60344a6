#!/usr/bin/env python3
"""ISA-level instruction mix of the hot loop bodies of a gfx950 kernel, from hipcc's assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S zk-nullifier-sig_amd/csrc/plume_kernels.hip -o /tmp/k.s
    python profiles/isa_mix.py /tmp/k.s k_verify_msm [min_instructions_per_block]

Splits the kernel into basic blocks (labels), classifies every instruction, and prints the blocks with at least `min` instructions
(the doubling body and the mixed-addition body of the multi-scalar loop are the two big ones).  Classes follow the measured issue costs
(tests/gpu_debug/instr_rates_r01.txt): v_mad_u64_u32 and every other VOP3-encoded VALU op ~1.8 ns per wave-instruction per SIMD, carry-less
VOP1/VOP2 ops ~1.05 ns, s_nop 0.5 ns.
"""
import collections
import re
import sys

VOP3_ONLY = ("v_mad_", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_alignbit", "v_bfe_", "v_bfi_", "v_and_or", "v_or3", "v_xad", "v_add3", "v_lshl_add", "v_lshl_or",
             "v_add_lshl", "v_mul_lo", "v_mul_hi", "v_perm", "v_cndmask_b32_e64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_cmp", "v_readlane",
             "v_writelane", "v_med3", "v_min3", "v_max3", "v_fma", "v_div", "v_mbcnt", "v_xor3", "v_sad", "v_lerp", "v_cvt_pk")


def classify(ins):
    op = ins.split()[0]
    if op == "v_mad_u64_u32":
        return "mad64"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("v_"):
        if op.endswith("_e64") or op.startswith(VOP3_ONLY):
            return "valu_vop3"
        return "valu_plain"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_ZN\d*plume\d+{kernel}E.*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    blocks, cur, name, nfall = [], collections.Counter(), "entry", 0
    ops = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        c = classify(t)
        cur[c] += 1
        ops[name][t.split()[0]] += 1
        if c == "branch":                      # a block ends at its branch: what follows a loop's back edge is a new (unnamed) block
            blocks.append((name, cur))
            nfall += 1
            cur, name = collections.Counter(), f"{name}+{nfall}"
    blocks.append((name, cur))
    tot = collections.Counter()
    for _, c in blocks:
        tot.update(c)
    cols = ["mad64", "valu_vop3", "valu_plain", "s_nop", "salu", "vmem", "lds", "scratch", "waitcnt", "branch"]
    print(f"# {kernel}: {len(blocks)} basic blocks, {sum(tot.values())} instructions in the listing (static count)")
    print("block".ljust(14) + "".join(c.rjust(11) for c in cols) + "   VALU  est_ns/wave")
    for name, c in blocks:
        n = sum(c.values())
        if n < mn:
            continue
        valu = c["mad64"] + c["valu_vop3"] + c["valu_plain"]
        est = 1.8 * (c["mad64"] + c["valu_vop3"]) + 1.05 * c["valu_plain"] + 0.5 * c["s_nop"]
        print(name.ljust(14) + "".join(str(c[k]).rjust(11) for k in cols) + f"{valu:7d} {est:9.0f}")
        top = ", ".join(f"{o} {k}" for o, k in ops[name].most_common(14))
        print("    " + top)


if __name__ == "__main__":
    main()
