#!/usr/bin/env python3
"""Generate plume_fe_asm.h: hand-scheduled gfx950 assembly for the Fp multiplication and squaring.

    python gen_fe_asm.py > plume_fe_asm.h

Why assembly: hipcc's code for the 8x32-limb product spends more issue slots on zero-extension moves (the 64-bit addend
of v_mad_u64_u32 needs a {limb, 0} register pair, 48 v_mov per multiplication) and on hazard padding (~36 s_nop) than
is necessary.  Here every running limb lives in the LOW register of a pair whose HIGH register is zeroed once per
call, products go to a separate set of pairs, and the carry chain of row i is interleaved with the multiply-adds of
row i+1, so consecutive v_addc are always separated by a v_mad_u64_u32.

Register plan (physical, clobbered): T[k] = v[TB+2k : TB+2k+1], k = 0..15  (limb k of the 512-bit product, high half 0)
                                     P[j] = v[PB+2j : PB+2j+1], j = 0..7   (products of the current row)
Operands: %0..%7 result limbs (early-clobber), %8..%15 a, %16..%23 b (squaring: %8..%15 a only).
"""
TB, PB = 96, 128          # 32 + 16 VGPRs: v96..v143
SCARRY = "s[92:93]"       # dummy carry-out of v_mad_u64_u32
S977 = "s94"


def T(k):
    return f"v[{TB + 2 * k}:{TB + 2 * k + 1}]"


def Tlo(k):
    return f"v{TB + 2 * k}"


def Thi(k):
    return f"v{TB + 2 * k + 1}"


def P(j):
    return f"v[{PB + 2 * j}:{PB + 2 * j + 1}]"


def Plo(j):
    return f"v{PB + 2 * j}"


def Phi(j):
    return f"v{PB + 2 * j + 1}"


def mul_wide(ins, a, b):
    """t = a * b (16 limbs) into Tlo(0..15).  a, b: lists of 8 operand strings."""
    # zero the high halves of the pairs that serve as 64-bit addends (positions 1..14; 0 and 15 never do)
    for k in range(1, 15):
        ins.append(f"v_mov_b32 {Thi(k)}, 0")
    # row 0: products with a zero addend
    for j in range(8):
        ins.append(f"v_mad_u64_u32 {P(j)}, {SCARRY}, {a[0]}, {b[j]}, 0")
    # rows: chain of row i interleaved with the products of row i+1
    for i in range(8):
        # chain of row i writes t[i .. i+8]; product (i+1, j) needs t[i+1+j], i.e. chain step j+1
        steps = []
        steps.append(f"v_mov_b32 {Tlo(i)}, {Plo(0)}")
        for j in range(1, 8):
            op = "v_add_co_u32" if j == 1 else "v_addc_co_u32"
            tail = "" if j == 1 else ", vcc"
            steps.append(f"{op} {Tlo(i + j)}, vcc, {Plo(j)}, {Phi(j - 1)}{tail}")
        steps.append(f"v_addc_co_u32 {Tlo(i + 8)}, vcc, 0, {Phi(7)}, vcc")
        # interleave: after chain step j+1 is issued, product (i+1, j) may go — but it overwrites P[j], whose lo/hi are read by
        # chain steps j and j+1, so product (i+1, j) is placed after chain step j+1.
        nxt = []
        if i < 7:
            for j in range(8):
                nxt.append(f"v_mad_u64_u32 {P(j)}, {SCARRY}, {a[i + 1]}, {b[j]}, {T(i + 1 + j)}")
        # step s (0..8) issued, then product j = s-1 (needs chain steps j and j+1 done: s >= j+1)
        for s, st in enumerate(steps):
            ins.append(st)
            if i < 7 and s >= 1:
                ins.append(nxt[s - 1])


def reduce_wide(ins, r):
    """Tlo(0..15) -> r[0..7] weakly reduced (same algebra as fe_reduce_wide in plume_field.h)."""
    ins.append(f"s_movk_i32 {S977}, 0x3d1")
    # Q_i = hi_i * 977 + {lo_i, 0}  (T[i].hi is still zero for i = 1..7; T[0].hi was never zeroed: zero it now)
    ins.append(f"v_mov_b32 {Thi(0)}, 0")
    for i in range(8):
        ins.append(f"v_mad_u64_u32 {P(i)}, {SCARRY}, {Tlo(8 + i)}, {S977}, {T(i)}")
    # u_i = Q_i.lo + Q_{i-1}.hi (+c), u_8 = Q_7.hi + c      -> Tlo(0..7), u8 in Thi(0)
    ins.append(f"v_mov_b32 {Tlo(0)}, {Plo(0)}")
    for i in range(1, 8):
        op = "v_add_co_u32" if i == 1 else "v_addc_co_u32"
        tail = "" if i == 1 else ", vcc"
        ins.append(f"{op} {Tlo(i)}, vcc, {Plo(i)}, {Phi(i - 1)}{tail}")
    U8, U9 = Thi(0), Thi(1)
    ins.append(f"v_addc_co_u32 {U8}, vcc, 0, {Phi(7)}, vcc")
    # u += hi << 32 : u_i += hi_{i-1} for i = 1..7, u_8 += hi_7, u_9 = carry
    for i in range(1, 8):
        op = "v_add_co_u32" if i == 1 else "v_addc_co_u32"
        tail = "" if i == 1 else ", vcc"
        ins.append(f"{op} {Tlo(i)}, vcc, {Tlo(i)}, {Tlo(8 + i - 1)}{tail}")
    ins.append(f"v_addc_co_u32 {U8}, vcc, {U8}, {Tlo(15)}, vcc")
    ins.append(f"v_mov_b32 {U9}, 0")
    ins.append(f"v_addc_co_u32 {U9}, vcc, 0, {U9}, vcc")
    # top = u8 + u9 * 2^32 (< 2^34).  top * PC = top*977 + (top << 32):
    #   M = u8*977 + {0, u9*977}  (64-bit), added at limbs 0,1 ; then u8 at limb 1 and u9 at limb 2
    M = P(0)
    ins.append(f"v_mul_u32_u24 {Phi(1)}, {S977}, {U9}")          # u9 * 977 (u9 is 0 or 1)
    ins.append(f"v_mov_b32 {Plo(1)}, 0")
    ins.append(f"v_mad_u64_u32 {M}, {SCARRY}, {U8}, {S977}, {P(1)}")
    ins.append(f"v_add_co_u32 {r[0]}, vcc, {Tlo(0)}, {Plo(0)}")
    ins.append(f"v_addc_co_u32 {r[1]}, vcc, {Tlo(1)}, {Phi(0)}, vcc")
    for i in range(2, 8):
        ins.append(f"v_addc_co_u32 {r[i]}, vcc, 0, {Tlo(i)}, vcc")
    C1 = Plo(2)
    ins.append(f"v_mov_b32 {C1}, 0")
    ins.append(f"v_addc_co_u32 {C1}, vcc, 0, {C1}, vcc")          # carry of chain 1
    ins.append(f"v_add_co_u32 {r[1]}, vcc, {r[1]}, {U8}")
    ins.append(f"v_addc_co_u32 {r[2]}, vcc, {r[2]}, {U9}, vcc")
    for i in range(3, 8):
        ins.append(f"v_addc_co_u32 {r[i]}, vcc, 0, {r[i]}, vcc")
    ins.append(f"v_addc_co_u32 {C1}, vcc, 0, {C1}, vcc")          # + carry of chain 2  (sum is 0 or 1)
    # fold carry: r += C1 * PC, and once more for the (tiny) second wrap -- exactly fe_fold_carry
    K = Plo(3)
    ins.append(f"v_mul_u32_u24 {K}, {S977}, {C1}")
    ins.append(f"v_add_co_u32 {r[0]}, vcc, {r[0]}, {K}")
    ins.append(f"v_addc_co_u32 {r[1]}, vcc, {r[1]}, {C1}, vcc")
    for i in range(2, 8):
        ins.append(f"v_addc_co_u32 {r[i]}, vcc, 0, {r[i]}, vcc")
    ins.append(f"v_mov_b32 {C1}, 0")
    ins.append(f"v_addc_co_u32 {C1}, vcc, 0, {C1}, vcc")
    ins.append(f"v_mul_u32_u24 {K}, {S977}, {C1}")
    ins.append(f"v_add_co_u32 {r[0]}, vcc, {r[0]}, {K}")
    ins.append(f"v_addc_co_u32 {r[1]}, vcc, {r[1]}, {C1}, vcc")


def emit(name, ins, ninputs):
    body = "\\n\\t\"\n        \"".join(ins)
    outs = ", ".join(f'"=&v"(r.v[{i}])' for i in range(8))
    if ninputs == 16:
        inps = ", ".join(f'"v"(a.v[{i}])' for i in range(8)) + ", " + ", ".join(f'"v"(b.v[{i}])' for i in range(8))
        sig = f"PLUME_DEV void {name}(fe& r, const fe& a, const fe& b)"
    else:
        inps = ", ".join(f'"v"(a.v[{i}])' for i in range(8))
        sig = f"PLUME_DEV void {name}(fe& r, const fe& a)"
    clob = ", ".join(f'"v{k}"' for k in range(TB, PB + 16)) + ', "s92", "s93", "s94", "vcc"'
    return f"""{sig} {{
    asm volatile(
        "{body}"
        : {outs}
        : {inps}
        : {clob});
}}
"""


def main():
    r = [f"%{i}" for i in range(8)]
    a = [f"%{8 + i}" for i in range(8)]
    b = [f"%{16 + i}" for i in range(8)]
    ins = []
    mul_wide(ins, a, b)
    reduce_wide(ins, r)
    mul_src = emit("fe_mul_asm", ins, 16)
    nm = sum(1 for x in ins if x.startswith("v_mad_u64"))
    ins2 = []
    mul_wide(ins2, a, a)          # squaring through the same schedule (operands repeated): correct, not yet specialised
    reduce_wide(ins2, r)
    sqr_src = emit("fe_sqr_asm", ins2, 8)
    print(f"""// GENERATED by gen_fe_asm.py — do not edit.  gfx950 assembly for the secp256k1 field multiplication.
// {len(ins)} instructions per multiplication ({nm} v_mad_u64_u32, {len(ins) - nm} plain), no s_nop, no zero-extension moves in the rows.
#pragma once
#include "plume_field.h"
#if defined(__HIP_DEVICE_COMPILE__)
#define PLUME_DEV __device__ __forceinline__
namespace plume {{
{mul_src}
{sqr_src}
}}  // namespace plume
#else
// host pass of hipcc / plain host builds: same results through the portable C++ (the assembly exists for the device only)
namespace plume {{
PLUME_HD void fe_mul_asm(fe& r, const fe& a, const fe& b) {{ fe_mul(r, a, b); }}
PLUME_HD void fe_sqr_asm(fe& r, const fe& a) {{ fe_sqr(r, a); }}
}}  // namespace plume
#endif
""")


if __name__ == "__main__":
    main()
