#!/usr/bin/env python3
"""Generate plume_fe_mul.inc: the secp256k1 field multiplication and squaring on 9 x 29-bit limbs for gfx950.

    python gen_fe_mul.py > plume_fe_mul.inc          (the generated file is committed; rerun only when this script changes)

Why this shape (measured on MI355X, tests/gpu_debug/instr_rates_r01.txt): a carry-chain step (v_add_co / v_addc_co) costs as
much issue time as a 32x32+64 multiply-add (v_mad_u64_u32), so the fast multiplication is the one with no carry steps at all.
With 29-bit limbs a whole product column (<= 9 products < 2^60.4 each) accumulates inside the 64-bit addend of a CHAIN of
v_mad_u64_u32, and the carry into the next column is simply that chain's initial addend:

    high half   columns 9..16 -> h[0..8]   (h[k] has weight 2^(261+29k); 2^261 = 2^37 + 31264 (mod p))
    low half    column k = sum a_i b_(k-i) + h[k]*31264 + h[k-1]*2^8          (k = 0..8; the fold is two more multiply-adds)
    tail        bits >= 2^256 of column 8 fold as t*(2^32 + 977) into limbs 0..2; h[8]'s 2^8 part lands on column 9 = 2^261 again and
                joins the tail: limb 0 += h[8]*(31264 << 8), limb 1 += h[8] << 16 (so the low half never waits for the last high column)

Round 2: the HIGH columns hand their excess on with one more multiply-add instead of a mask and a 64-bit shift -- h[k] is the column's low 32-bit register as it
stands, the high register (weight 2^32 = 8 * 2^29 relative to the column) opens the next column's chain as `v_mad_u64_u32 acc, hi, 8, 0`, writing a fresh accumulator
pair so that nothing is copied; 32-bit h[] are harmless because the fold multiplies them by 31264 and 256 only.  fe_sqr3 (3a^2) and fe_sqr_d (a^2 and 2a) ride factors
in the squaring's operands for the group law's doubling.

Round 3: (1) the chains that open a multiplication start from the literal 0 (no zeroed register pair); (2) fe_mul_sub / fe_sqr_sub / fe_sqr_sub2 take the group law's
"product minus something" into the product's own fold: the unreduced difference M*p - s joins the low columns as one more multiply-add per limb, so the subtraction needs no
carry pass of its own (18 instructions instead of 47) and the result is as tight as any product.  What an instruction costs in the kernels is the SUM of the single-kind
issue costs (multiply-add 1.83 ns, other VOP3 1.77, plain VOP1/VOP2 1.05, s_nop 0 per wave-instruction per SIMD -- DESIGN.md section 6), so trading 29 plain / VOP3
instructions for 9 multiply-adds pays.

110 multiply-adds (81 products + 18 fold + 7 hand-offs + 4 tail) + 9 v_lshrrev_b64 + 12 v_and_b32 per multiplication (squaring: 74 multiply-adds; round 1: 103 + 16 + 20, squaring 67).  Each chain is ONE asm
statement with compiler-allocated registers: hipcc keeps scheduling and register allocation but can neither re-associate the
chain (that costs a 64-bit add per column) nor strength-reduce the fold constants into shift/add pairs, and it inserts no
hazard nops inside a statement.  Host builds (tests/devsim) compile the same column algorithm as plain C++.
"""


SQR_DIAG = "a.v[{i}]"   # second operand of the diagonal products of a squaring ("t3[{i}]" in fe_sqr3, "d[{i}]" in fe_sqr2)
SQR_CROSS = "a.v[{i}]"  # first operand of the cross products of a squaring ("d[{i}]" in fe_sqr2: d_i d_j = 4 a_i a_j)


def col_terms(k, sqr, pair=("a", "b")):
    t = []
    x, y = pair
    for i in range(9):
        j = k - i
        if j < 0 or j > 8:
            continue
        if not sqr:
            t.append((f"{x}.v[{i}]", f"{y}.v[{j}]"))
        elif i < j:
            t.append((SQR_CROSS.format(i=i), f"d[{j}]"))
        elif i == j:
            t.append((f"a.v[{i}]", SQR_DIAG.format(i=i)))
    return t


ZERO_START = True   # the two chains that open a multiplication (column 9, column 0) start from the literal 0 instead of a zeroed register pair; --no-zero-start for the A/B build
HICARRY = True   # high-half hand-off by one multiply-add (carry = high word * 8) instead of mask + 64-bit shift; --no-hicarry for the A/B build
A_CONS = "v"   # constraint of the a.v[i] operands: "s" in fe_mul_k (a is a wave-uniform constant held in SGPRs)


SHIFT_IN_ASM = False  # round 3 experiment (--shift-in-asm): the low columns' 64-bit shift rides at the end of the column's asm statement and writes a FRESH pair, so that the limb
                      # mask need not follow the statement.  It halves the s_nop hipcc pads asm results with (mixed addition body 143 -> 71, doubling 54 -> 24; 110 VGPRs, bit-exact) and
                      # changes NOTHING measurable: multi-scalar kernel 18.69 / 18.58 vs 18.67 ms on one (slow) box.  The nops are absorbed while other wavefronts issue (round 1 saw
                      # the same with its interleaving); the instruction-diet item of VERDICT r2 that named them was chasing a cost the kernel does not pay.  Off: the shipped file is round 2's.


def emit_chain(terms, indent="    ", acc="acc", carry=None, shift_out=None, zero=False):
    """zero: the accumulator starts at 0: the first multiply-add takes the literal 0 as its addend and writes a fresh pair (round 3: no `v_mov_b64 acc, 0` in front
    of the multiplication's two opening chains -- 2 of its 131 instructions).
    carry: name of a 32-bit variable holding the previous column's high word: the chain then STARTS a fresh accumulator with carry * 8 (the
    hand-off multiply-add) instead of continuing in place.
    shift_out: name of a 64-bit variable that receives acc >> 29 from a v_lshrrev_b64 appended to the statement.  The column's own pair stays as it is, so the limb mask
    (a plain v_and on its low half) no longer has to follow the statement directly -- hipcc pads every instruction that reads an asm statement's result right behind it
    with an s_nop -- and becomes filler the scheduler can place between other statements."""
    regs = []
    base = 3 if shift_out else 2

    def idx(name, cons):
        key = (name, cons)
        if key not in regs:
            regs.append(key)
        return regs.index(key) + base

    lines = []
    if carry:
        lines.append(f"v_mad_u64_u32 %0, %1, %{idx(carry, 'v')}, 8, 0")
    cy = "%2" if shift_out else "%1"
    for n, (x, y, ys) in enumerate(terms):
        xc = A_CONS if x.startswith("a.v[") else "v"
        xop = f"%{idx(x, xc)}"
        yop = y if ys == "lit" else f"%{idx(y, 's' if ys else 'v')}"
        lines.append(f"v_mad_u64_u32 %0, {cy}, {xop}, {yop}, " + ("0" if (zero and n == 0) else "%0"))
    if shift_out:
        lines.append("v_lshrrev_b64 %1, 29, %0")
    body = "\\n\\t".join(lines)
    ins = ", ".join(f'"{c}"({n})' for (n, c) in regs)
    host = " ".join(f"{acc} += (uint64_t){x} * {y}{'u' if ys == 'lit' else ''};" for (x, y, ys) in terms)
    if shift_out:
        assert not carry
        return (f"{indent}PLUME_FE_CHAIN_SHIFT({acc}, {shift_out}, \"{body}\", {ins});\n", f"{indent}{host} {shift_out} = {acc} >> 29;\n")
    if carry:
        return (f"{indent}PLUME_FE_CHAIN_NEW({acc}, \"{body}\", {ins});\n", f"{indent}{acc} = (uint64_t){carry} * 8u; {host}\n")
    if zero:
        assert not shift_out
        return (f"{indent}PLUME_FE_CHAIN_NEW({acc}, \"{body}\", {ins});\n", f"{indent}{acc} = 0; {host}\n")
    return (f"{indent}PLUME_FE_CHAIN({acc}, \"{body}\", {ins});\n", f"{indent}{host}\n")


def gen(name, sqr, two=False, scale3=False, expose_d=False, scale2=False, sub=None):
    """sub (1 or 2): r = a*b + sub * (M*p - s) for a template parameter M: the group law's "product minus something, then a carry pass" in ONE fold -- the unreduced
    difference w[k] = M*p[k] - s[k] (one v_sub per limb) joins low column k as the multiply-add w[k] * sub, so the subtraction costs 18 instructions instead of the 47
    of fe_sub_lazy + fe_carry, and the result is as tight as any product.
    two: r = a*b + c*d with ONE fold -- the two products share their column sums (each column: two chain statements).

    Statement order: the high-half chain (columns 9..16, accumulator acch) and the low-half chain (columns 0..8, accumulator acc)
    are INTERLEAVED, the low half lagging by two columns (column k needs h[k] and h[k-1]): every asm statement is followed by an
    independent one, so the mask / shift that reads its result is no longer the very next instruction and hipcc does not have to
    pad with s_nop (it places one after every inline-asm statement whose result the next instruction reads)."""
    dev, host = [], []

    def both(s):
        dev.append(s)
        host.append(s)

    def stmts_high(k):
        out = [emit_chain([(x, y, False) for (x, y) in col_terms(k, sqr)], acc="acch", carry=(f"hw{k - 1}" if (HICARRY and k > 9) else None), zero=(ZERO_START and k == 9))]
        if two:
            out.append(emit_chain([(x, y, False) for (x, y) in col_terms(k, False, ("c", "e"))], acc="acch"))
        return out

    def mask_high(k):
        if HICARRY:
            # h[k-9] = the column's low 32 bits as they are (a free sub-register, no mask); the rest, acch >> 32, has weight 2^32 = 8 * 2^29 relative to
            # this column and enters the next one as hi * 8: ONE multiply-add instead of v_and + v_lshrrev_b64.  The fold multiplies h[] by 31264 and
            # 256 only, so 32-bit h[] leave the column bounds where they were (h * 31264 < 2^47).
            # The hand-off instruction opens the next column's asm statement (emit_chain(carry=...)), which writes a FRESH accumulator pair: the old
            # pair's halves stay where they are as h[] and the carry, no copies.
            if k < 16:
                return f"    h[{k - 9}] = (uint32_t)acch; const uint32_t hw{k} = (uint32_t)(acch >> 32);\n"
            return "    h[7] = (uint32_t)acch; h[8] = (uint32_t)(acch >> 32);   // h[8] counts units of 8 * 2^(261+232): constants K0H, K2H, K3H\n    PLUME_FE_ASSERT((acch >> 32) < (1ull << 24));\n"
        if k < 16:
            return f"    h[{k - 9}] = (uint32_t)acch & PLUME_FE_MASK; acch >>= 29;\n"
        return "    h[7] = (uint32_t)acch & PLUME_FE_MASK; h[8] = (uint32_t)(acch >> 29);\n    PLUME_FE_ASSERT((acch >> 29) < (1ull << 27));\n"

    def stmts_low(k):
        out = []
        z = ZERO_START and k == 0 and not SHIFT_IN_ASM
        if two:
            out.append(emit_chain([(x, y, False) for (x, y) in col_terms(k, False, ("c", "e"))], zero=z))
            z = False
        t = [(x, y, False) for (x, y) in col_terms(k, sqr)]
        t.append((f"h[{k}]", "K0H" if (HICARRY and k == 8) else "K0", True))
        if k > 0:
            t.append((f"h[{k - 1}]", "K1", True))
        if sub:
            t.append((f"w[{k}]", str(sub), "lit"))
        out.append(emit_chain(t, shift_out=(f"accn{k}" if (SHIFT_IN_ASM and k < 8) else None), zero=z))
        return out

    def mask_low(k):
        if SHIFT_IN_ASM:
            return f"    l[{k}] = (uint32_t)acc & PLUME_FE_MASK; acc = accn{k};\n"
        return f"    l[{k}] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29;\n"

    def put(pairs):
        for d, h in pairs:
            dev.append(d)
            host.append(h)

    both("    uint32_t h[9], l[9];     // l: result limbs (r may alias a or b)\n    uint64_t acc = 0, acch = 0;\n")
    if SHIFT_IN_ASM:
        both("    uint64_t accn0, accn1, accn2, accn3, accn4, accn5, accn6, accn7;\n")
    if sub:
        both("    uint32_t w[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) { PLUME_FE_ASSERT(s.v[i] <= (uint32_t)M * fe_p(i)); w[i] = (uint32_t)M * fe_p(i) - s.v[i]; }\n")
    if sqr and scale3:
        # r = 3 a^2: the cross products take 6 a_j, the diagonal ones 3 a_i; the column sums are three times a squaring's (27 T^2 < 2^63 for tight a)
        both("    uint32_t d[9], t3[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) { t3[i] = a.v[i] + u32_dbl(a.v[i]); d[i] = u32_dbl(t3[i]); }\n")
    elif sqr:
        both("    uint32_t d[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) d[i] = u32_dbl(a.v[i]);\n")
        if expose_d:
            both("    PLUME_UNROLL for (int i = 0; i < 9; i++) dbl.v[i] = d[i];\n")
    # H9, mH9, H10, then per step j: L(j), mH(j+10), H(j+11), mL(j)
    put(stmts_high(9)); both(mask_high(9)); put(stmts_high(10))
    for j in range(0, 9):
        put(stmts_low(j))
        if j + 10 <= 16:
            both(mask_high(j + 10))
        if j + 11 <= 16:
            put(stmts_high(j + 11))
        if j < 8:
            both(mask_low(j))
    both("""    // acc = column 8 (weight 2^232): bits >= 24 are multiples of 2^256 -> t = t0 + t1 * 2^29, times (2^32 + 977).  h[8] (weight 2^(261+232))
    // joins here: its 31264 part went to column 8 above, its 2^8 part is 2^261 again: limb 0 += h[8] * (31264 << 8), limb 1 += h[8] << 16
    l[8] = (uint32_t)acc & 0x00FFFFFFu;
    acc >>= 24;
    const uint32_t t0 = (uint32_t)acc & PLUME_FE_MASK, t1 = (uint32_t)(acc >> 29);
    acc = l[0];
""")
    k2, k3 = ("K2H", "K3H") if HICARRY else ("K2", "K3")
    dev.append(f'    PLUME_FE_CHAIN(acc, "v_mad_u64_u32 %0, %1, %2, %3, %0\\n\\tv_mad_u64_u32 %0, %1, %4, %5, %0", "v"(t0), "s"(K4), "v"(h[8]), "s"({k2}));\n')
    host.append(f"    acc += (uint64_t)t0 * K4; acc += (uint64_t)h[8] * {k2};\n")
    both("    l[0] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29;\n    acc += l[1] + t1 * 977u;\n")
    dev.append(f'    PLUME_FE_CHAIN(acc, "v_mad_u64_u32 %0, %1, %2, %3, %0\\n\\tv_mad_u64_u32 %0, %1, %4, %5, %0", "v"(t0), "s"(K5), "v"(h[8]), "s"({k3}));\n')
    host.append(f"    acc += (uint64_t)t0 * K5; acc += (uint64_t)h[8] * {k3};\n")
    both("    l[1] = (uint32_t)acc & PLUME_FE_MASK;\n    l[2] += (uint32_t)(acc >> 29) + (t1 << 3);\n    PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = l[i];\n")
    sig = f"PLUME_HD void {name}(fe& r, const fe& a)" if sqr else f"PLUME_HD void {name}(fe& r, const fe& a, const fe& b)"
    check = "    PLUME_FE_ASSERT(fe_mul_inputs_ok(a, a));\n" if sqr else "    PLUME_FE_ASSERT(fe_mul_inputs_ok(a, b));\n"
    if scale3 or scale2:
        check = "    PLUME_FE_ASSERT(fe_is_tight(a));\n"
    if expose_d and scale2:
        sig = f"// r = 2 a^2, dbl = 2a (group law: 2Y^2 and Z' = (2Y) Z from one squaring); dbl must not alias a\nPLUME_HD void {name}(fe& r, fe& dbl, const fe& a)"
    elif expose_d:
        sig = f"// r = a^2, dbl = 2a: the doubled limbs the squaring forms anyway (group law: Z' = (2Y) Z); dbl must not alias a\nPLUME_HD void {name}(fe& r, fe& dbl, const fe& a)"
    if two:
        sig = f"PLUME_HD void {name}(fe& r, const fe& a, const fe& b, const fe& c, const fe& e)"
        check = "    PLUME_FE_ASSERT(fe_muladd_inputs_ok(a, b, c, e));\n"
    if sub:
        sig = "template <int M>\n" + (f"PLUME_HD void {name}(fe& r, const fe& a, const fe& s)" if sqr else f"PLUME_HD void {name}(fe& r, const fe& a, const fe& b, const fe& s)")
        check += "    static_assert(M >= 1 && M <= 7, \"M*p limbs must fit 32 bits\");\n"
    consts = "    const uint32_t K0 = 31264u, K1 = 256u, K2 = 31264u << 8, K3 = 65536u, K4 = 977u, K5 = 8u;\n"
    if HICARRY:
        consts = "    const uint32_t K0 = 31264u, K1 = 256u, K0H = 31264u << 3, K2H = 31264u << 11, K3H = 65536u << 3, K4 = 977u, K5 = 8u;\n"
    return f"{sig} {{\n{check}{consts}#if defined(__HIP_DEVICE_COMPILE__)\n{''.join(dev)}#else\n{''.join(host)}#endif\n}}\n"


def main():
    import sys
    global HICARRY
    if "--no-hicarry" in sys.argv:
        HICARRY = False
    global ZERO_START
    if "--no-zero-start" in sys.argv:
        ZERO_START = False
    global SHIFT_IN_ASM
    if "--shift-in-asm" in sys.argv:
        SHIFT_IN_ASM = True
    print("""// GENERATED by gen_fe_mul.py -- do not edit (see that script for the design and the measurements behind it).
// Included by plume_field.h inside namespace plume.
// one chain of multiply-adds on `acc`; the carry-out pair of v_mad_u64_u32 is a dead SGPR pair the compiler picks
#define PLUME_FE_CHAIN(ACC, TEXT, ...) do { uint64_t cy_; asm(TEXT : "+v"(ACC), "=&s"(cy_) : __VA_ARGS__); } while (0)
// the same, but the chain's first instruction starts a fresh accumulator (previous column's high word * 8: a 29-bit limb step is 2^29 = 2^32 / 8)
#define PLUME_FE_CHAIN_NEW(ACC, TEXT, ...) do { uint64_t cy_; asm(TEXT : "=&v"(ACC), "=&s"(cy_) : __VA_ARGS__); } while (0)
// a low column: the chain, then NEXT = ACC >> 29 into a fresh pair (ACC keeps the column: its low half is masked into the limb whenever the scheduler likes)
#define PLUME_FE_CHAIN_SHIFT(ACC, NEXT, TEXT, ...) do { uint64_t cy_; asm(TEXT : "+v"(ACC), "=&v"(NEXT), "=&s"(cy_) : __VA_ARGS__); } while (0)
""")
    print(gen("fe_mul", False))
    print(gen("fe_sqr", True))
    print(gen("fe_sqr_d", True, expose_d=True))
    global SQR_DIAG
    SQR_DIAG = "t3[{i}]"
    print("// r = 3 a^2 for a TIGHT a, tight result: the factor rides in the operands (group law: E = 3 X^2 without the tripling and its carry pass)")
    print(gen("fe_sqr3", True, scale3=True))
    global SQR_CROSS
    SQR_DIAG, SQR_CROSS = "d[{i}]", "d[{i}]"
    print("// r = 2 a^2 for a TIGHT a, tight result, at the cost of a plain squaring: the cross products are d_i d_j = 4 a_i a_j, the diagonal ones a_i d_i = 2 a_i^2 (d = 2a, which every")
    print("// squaring forms anyway); column sums 18 T^2 < 2^63.  The group law's doubling takes 2Y^2 and 8Y^4 = 2 (2Y^2)^2 from these instead of doubling afterwards.")
    print(gen("fe_sqr2", True, scale2=True))
    print(gen("fe_sqr2_d", True, scale2=True, expose_d=True))
    SQR_DIAG, SQR_CROSS = "a.v[{i}]", "a.v[{i}]"
    global A_CONS
    A_CONS_SAVE = A_CONS
    A_CONS = "v"
    print("// r = a*b + c*e with one fold: the 17 column sums of both products accumulate in the same chains (bound: fe_muladd_inputs_ok)")
    print(gen("fe_muladd", False, True))
    print("// r = a*b - s = a*b + (M*p - s): the difference is formed limbwise (needs s[i] <= M*p[i]: M = 2 for a tight s) and joins the low columns, so the result is as tight as a")
    print("// product's and the subtraction needs no carry pass of its own (group law: H = U2 - X1, r = S2 - Y1)")
    print(gen("fe_mul_sub", False, sub=1))
    print("// r = a^2 - s (group law: X3 = r^2 - (2V + H^3), M = 4)")
    print(gen("fe_sqr_sub", True, sub=1))
    print("// r = a^2 - 2s (doubling: X' = E^2 - 2 * 4XY^2, M = 2 for a tight s)")
    print(gen("fe_sqr_sub2", True, sub=2))
    A_CONS = "s"
    print("// a is a compile-time constant (curve / isogeny coefficients): its limbs stay in SGPRs instead of occupying 9 VGPRs each")
    print(gen("fe_mul_k", False))
    print("#undef PLUME_FE_CHAIN")
    print("#undef PLUME_FE_CHAIN_NEW")
    print("#undef PLUME_FE_CHAIN_SHIFT")


if __name__ == "__main__":
    main()
