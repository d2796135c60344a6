#!/usr/bin/env python3
"""Generate plume_fe_mul.inc: the secp256k1 field multiplication and squaring on 9 x 29-bit limbs for gfx950.

    python gen_fe_mul.py > plume_fe_mul.inc          (the generated file is committed; rerun only when this script changes)

Why this shape (measured on MI355X, tests/gpu_debug/instr_rates_r01.txt): a carry-chain step (v_add_co / v_addc_co) costs as
much issue time as a 32x32+64 multiply-add (v_mad_u64_u32), so the fast multiplication is the one with no carry steps at all.
With 29-bit limbs a whole product column (<= 9 products < 2^60.4 each) accumulates inside the 64-bit addend of a CHAIN of
v_mad_u64_u32, and the carry into the next column is simply that chain's initial addend.  Order of one multiplication (round 4):

    high half   columns 9..16 -> h[0..8]   (h[k] has weight 2^(261+29k); 2^261 = 2^37 + 31264 (mod p)).  h[k] is the column's low 32-bit register as it
                stands; the high register (weight 2^32 = 8 * 2^29 relative to the column) opens the next column's chain as `v_mad_u64_u32 acc, hi, 8, 0`
                (a fresh accumulator pair, nothing copied).  32-bit h[] are harmless: the fold multiplies them by 31264 and 256 only.
    top first   column 7 WITHOUT its carry-in (E7), handed on the same way (x7 = its low 32 bits stay behind), then column 8 (E8).  Bits >= 24 of E8 are
                multiples of 2^256 = 2^32 + 977 (mod p).  E8's high word takes h[8] << 8 by a plain add first (h[8] * 8 * (2^37 + 31264) * 2^232: the 2^40 part lands
                exactly there -- which is why limb 8 of a multiplication input must stay <= 2^26), then the excess is cut in two: tlo = bits 24..31 of the low word
                (weight 2^256), thi = the whole high word (weight 2^264).  Both go into columns 0 and 1 as four more multiply-adds BEFORE those columns run, with the
                constants the emitted code names: limb 0 += tlo * K4 + thi * K4H (K4 = 977, K4H = 977 << 8), limb 1 += tlo * 8 + thi * K11 (K11 = 2048 = 2^(32 + 8 - 29));
                h[8] itself joins column 8 as h[8] * K0H (31264 << 3).
    low half    column k = sum a_i b_(k-i) + h[k]*31264 + h[k-1]*2^8 (+ the terms above), k = 0..6: one v_and (limb) and one v_lshrrev_b64 (carry) each
    fix-up      column 6's carry + x7 (one multiply-add by the literal 1) -> limb 7 and a remainder < 2^8 that joins limb 8: tight, no second pass.

Rounds 1-3 ran the low half 0..8 in order and folded column 8's excess at the end, which touched limbs 0..2 a second time (two more masks, two shifts, a 64-bit
add, moves): 140 VALU instructions per multiplication against 132 now (111 multiply-adds + 21 others; squaring 112 -> 104, 75 + 29) -- the counts of
profiles/r04_isa_mix.txt (tests/test_abi_cpu.py holds the committed plume_fe_mul.inc to this script's output).

fe_sqr3 (3a^2), fe_sqr2 (2a^2), fe_sqr_d / fe_sqr2_d (also return 2a) ride factors in the squaring's operands for the group law's doubling;
fe_mul_sub / fe_sqr_sub / fe_sqr_sub2 take the group law's "product minus something" into the product's own fold: the unreduced difference M*p - s joins the low
columns as one more multiply-add per limb, so the subtraction needs no carry pass of its own (18 instructions instead of 47) and the result is as tight as any product.

Each chain is ONE asm statement with compiler-allocated registers: hipcc keeps scheduling and register allocation but can neither re-associate the
chain (that costs a 64-bit add per column) nor strength-reduce the fold constants into shift/add pairs, and it inserts no
hazard nops inside a statement.  Host builds (tests/devsim) compile the same column algorithm as plain C++.
(The switches of earlier rounds' A/B builds -- mask + shift hand-off, zeroed start registers, the shift inside the asm statement -- are gone; LABNOTES.md has their numbers.)
"""


SQR_DIAG = "a.v[{i}]"   # second operand of the diagonal products of a squaring ("t3[{i}]" in fe_sqr3, "d[{i}]" in fe_sqr2)
SQR_CROSS = "a.v[{i}]"  # first operand of the cross products of a squaring ("d[{i}]" in fe_sqr2: d_i d_j = 4 a_i a_j)
H8_BY_ADD = True
A_CONS = "v"   # constraint of the a.v[i] operands: "s" in fe_mul_k (a is a wave-uniform constant held in SGPRs)


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


def emit_chain(terms, indent="    ", acc="acc", carry=None, zero=False):
    """One column's multiply-adds as one asm statement (and the same sum as plain C++ for the host build).
    terms: (x, y, kind) with kind False: y is a VGPR value, True: y is a constant in an SGPR, "lit": y is an inline literal.
    zero: the accumulator starts at 0: the first multiply-add takes the literal 0 as its addend and writes a fresh pair.
    carry: name of a 32-bit variable holding the previous column's high word: the chain then STARTS a fresh accumulator with carry * 8 (the
    hand-off multiply-add) instead of continuing in place."""
    regs = []

    def idx(name, cons):
        key = (name, cons)
        if key not in regs:
            regs.append(key)
        return regs.index(key) + 2

    lines = []
    if carry:
        lines.append(f"v_mad_u64_u32 %0, %1, %{idx(carry, 'v')}, 8, 0")
    for n, (x, y, ys) in enumerate(terms):
        xc = A_CONS if x.startswith("a.v[") else "v"
        xop = f"%{idx(x, xc)}"
        yop = y if ys == "lit" else f"%{idx(y, 's' if ys else 'v')}"
        lines.append(f"v_mad_u64_u32 %0, %1, {xop}, {yop}, " + ("0" if (zero and n == 0) else "%0"))
    body = "\\n\\t".join(lines)
    ins = ", ".join(f'"{c}"({n})' for (n, c) in regs)
    host = " ".join(f"{acc} += (uint64_t){x} * {y}{'u' if ys == 'lit' else ''};" for (x, y, ys) in terms)
    if carry:
        return (f"{indent}PLUME_FE_CHAIN_NEW({acc}, \"{body}\", {ins});\n", f"{indent}{acc} = (uint64_t){carry} * 8u; {host}\n")
    if zero:
        return (f"{indent}PLUME_FE_CHAIN_NEW({acc}, \"{body}\", {ins});\n", f"{indent}{acc} = 0; {host}\n")
    return (f"{indent}PLUME_FE_CHAIN({acc}, \"{body}\", {ins});\n", f"{indent}{host}\n")


def gen(name, sqr, two=False, scale3=False, expose_d=False, scale2=False, sub=None):
    """sub (1 or 2): r = a*b + sub * (M*p - s) for a template parameter M: the group law's "product minus something, then a carry pass" in ONE fold -- the unreduced
    difference w[k] = M*p[k] - s[k] (one v_sub per limb) joins low column k as the multiply-add w[k] * sub, so the subtraction costs 18 instructions instead of the 47
    of fe_sub_lazy + fe_carry, and the result is as tight as any product.
    two: r = a*b + c*d with ONE fold -- the two products share their column sums (each column: two chain statements)."""
    dev, host = [], []

    def both(s):
        dev.append(s)
        host.append(s)

    def put(pairs):
        for d, h in pairs:
            dev.append(d)
            host.append(h)

    def prods(k, zero=False, carry=None, acc="acc", extra=()):
        """the statements of column k: [c*e products,] a*b products + extra terms"""
        out = []
        if two:
            out.append(emit_chain([(x, y, False) for (x, y) in col_terms(k, False, ("c", "e"))], acc=acc, zero=zero, carry=carry))
            zero, carry = False, None
        t = [(x, y, False) for (x, y) in col_terms(k, sqr)] + list(extra)
        out.append(emit_chain(t, acc=acc, zero=zero, carry=carry))
        return out

    def low_extra(k):
        t = [(f"h[{k}]", "K0H" if k == 8 else "K0", True)]
        if k > 0:
            t.append((f"h[{k - 1}]", "K1", True))
        if k == 0:
            t += [("thi", "K4H", True), ("tlo", "K4", True)] + ([] if H8_BY_ADD else [("h[8]", "K2H", True)])
        if k == 1:
            t += [("thi", "K11", True), ("tlo", "8", "lit")] + ([] if H8_BY_ADD else [("h[8]", "K3H", True)])
        if k == 6:
            t += [("x7", "K29", True)]
        if sub:
            t.append((f"w[{k}]", str(sub), "lit"))
        return t

    both("    uint32_t h[9], l[9];     // l: result limbs (r may alias a or b)\n    uint64_t acc = 0;\n")
    if sub:
        both("    uint32_t w[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) { PLUME_FE_ASSERT(s.v[i] <= (uint32_t)M * fe_p(i)); w[i] = (uint32_t)M * fe_p(i) - s.v[i]; }\n")
    if sqr and scale3:
        # r = 3 a^2: the cross products take 6 a_j, the diagonal ones 3 a_i; the column sums are three times a squaring's (27 T^2 < 2^63 for tight a)
        both("    uint32_t d[9], t3[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) { t3[i] = a.v[i] + u32_dbl(a.v[i]); d[i] = u32_dbl(t3[i]); }\n")
    elif sqr:
        both("    uint32_t d[9];\n    PLUME_UNROLL for (int i = 0; i < 9; i++) d[i] = u32_dbl(a.v[i]);\n")
        if expose_d:
            both("    PLUME_UNROLL for (int i = 0; i < 9; i++) dbl.v[i] = d[i];\n")
    # ---- high half: columns 9..16, each handing its high word (x 8) to the next
    for k in range(9, 17):
        put(prods(k, zero=(k == 9), carry=(f"hw{k - 1}" if k > 9 else None)))
        if k < 16:
            both(f"    h[{k - 9}] = (uint32_t)acc; const uint32_t hw{k} = (uint32_t)(acc >> 32);\n")
        else:
            both("    h[7] = (uint32_t)acc; h[8] = (uint32_t)(acc >> 32);   // h[8] counts units of 8 * 2^(261+232): constants K0H, K2H, K3H\n    PLUME_FE_ASSERT((acc >> 32) < (1ull << 24));\n")
    # ---- columns 7 and 8 before the others (column 7 without its carry-in): the excess of column 8 over 2^256 is known before limbs 0..2 are formed
    put(prods(7, zero=True, extra=low_extra(7)))
    both("    const uint32_t x7 = (uint32_t)acc, hw7 = (uint32_t)(acc >> 32);   // x7: what stays in column 7 (joined by column 6's carry at the end)\n")
    put(prods(8, carry="hw7", extra=low_extra(8)))
    both("    // acc = column 8 (weight 2^232) short of what column 6's carry will still push up (< 2^8, added to the limb at the end).  h[8] (units of 8 * 2^(261+232)) folds to\n"
         "    // h[8] * 8 * (2^37 + 31264) * 2^232 = h[8] * (2^40 + 250112) * 2^232: both parts land on this column, the 2^40 part as a plain add on the accumulator's high word.\n"
         "    // Bits >= 24 of the column are multiples of 2^256 = 2^32 + 977 (mod p): the high word whole (thi, weight 2^264) and bits 24..31 of the low word (tlo)\n"
         "    PLUME_FE_ASSERT((acc >> 32) + ((uint64_t)h[8] << 8) < (1ull << 32));\n"
         "    const uint32_t thi = (uint32_t)(acc >> 32) + (h[8] << 8), tlo = (uint32_t)acc >> 24;\n    l[8] = (uint32_t)acc & 0x00FFFFFFu;\n" if H8_BY_ADD else
         "    const uint32_t thi = (uint32_t)(acc >> 32), tlo = (uint32_t)acc >> 24;\n    l[8] = (uint32_t)acc & 0x00FFFFFFu;\n")
    # ---- columns 0..6 (column 6 also takes x7 * 2^29: its carry is then column 7 whole)
    for k in range(0, 7):
        put(prods(k, zero=(k == 0), extra=low_extra(k)))
        both(f"    l[{k}] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29;\n")
    both("    PLUME_FE_ASSERT(acc < (1ull << 37));\n    l[7] = (uint32_t)acc & PLUME_FE_MASK;\n    l[8] += (uint32_t)(acc >> 29);\n    PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = l[i];\n")
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
    consts = "    const uint32_t K0 = 31264u, K1 = 256u, K0H = 31264u << 3, K4 = 977u, K4H = 977u << 8, K29 = 1u << 29, K11 = 2048u;" + ("" if H8_BY_ADD else " const uint32_t K2H = 31264u << 11, K3H = 65536u << 3;") + "\n"
    return f"{sig} {{\n{check}{consts}#if defined(__HIP_DEVICE_COMPILE__)\n{''.join(dev)}#else\n{''.join(host)}#endif\n}}\n"


def main():
    print("""// GENERATED by gen_fe_mul.py -- do not edit (see that script for the design and the measurements behind it).
// Included by plume_field.h inside namespace plume.
// one chain of multiply-adds on `acc`; the carry-out pair of v_mad_u64_u32 is a dead SGPR pair the compiler picks
#define PLUME_FE_CHAIN(ACC, TEXT, ...) do { uint64_t cy_; asm(TEXT : "+v"(ACC), "=&s"(cy_) : __VA_ARGS__); } while (0)
// the same, but the chain's first instruction starts a fresh accumulator (0, or the previous column's high word * 8: a 29-bit limb step is 2^29 = 2^32 / 8)
#define PLUME_FE_CHAIN_NEW(ACC, TEXT, ...) do { uint64_t cy_; asm(TEXT : "=&v"(ACC), "=&s"(cy_) : __VA_ARGS__); } while (0)
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


if __name__ == "__main__":
    main()
