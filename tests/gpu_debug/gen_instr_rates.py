#!/usr/bin/env python3
"""Generate instr_rates.hip: sustained issue cost of the gfx950 VALU instructions the field arithmetic is (or could be) made of.

    python gen_instr_rates.py > instr_rates.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 instr_rates.hip -o instr_rates

Every kernel runs ITERS trips of one asm block of 32 instructions (8 independent chains x 4, unless the kind says otherwise) and
the host prints ns and shader-clock ticks per wave-instruction per SIMD at 1, 2, 4 and 8 waves per SIMD.
"""
KINDS = [
    # name, type of the 8 chain variables, template for chain i (uses {c} = chain operand, %8 %9 = inputs a, b)
    ("v_add_u32", "u32", "v_add_u32 {c}, %8, {c}"),
    ("v_mov_b32", "u32", "v_mov_b32 {c}, %8"),
    ("v_and_b32", "u32", "v_and_b32 {c}, %8, {c}"),
    ("v_xor_b32", "u32", "v_xor_b32 {c}, %8, {c}"),
    ("v_lshrrev_b32", "u32", "v_lshrrev_b32 {c}, 3, {c}"),
    ("v_alignbit_b32", "u32", "v_alignbit_b32 {c}, {c}, %8, 29"),
    ("v_add3_u32", "u32", "v_add3_u32 {c}, {c}, %8, %9"),
    ("v_lshl_add_u32", "u32", "v_lshl_add_u32 {c}, {c}, 3, %8"),
    ("v_and_or_b32", "u32", "v_and_or_b32 {c}, {c}, %8, %9"),
    ("v_bfe_u32", "u32", "v_bfe_u32 {c}, {c}, 3, 29"),
    ("v_add_co_u32 (writes vcc)", "u32", "v_add_co_u32 {c}, vcc, %8, {c}"),
    ("v_addc_co_u32 (vcc in+out)", "u32", "v_addc_co_u32 {c}, vcc, %8, {c}, vcc"),
    ("v_addc_co_u32 e64 (sgpr pair carry)", "u32", "v_addc_co_u32 {c}, s[20:21], %8, {c}, s[20:21]"),
    ("v_cndmask_b32 (vcc)", "u32", "v_cndmask_b32 {c}, {c}, %8, vcc"),
    ("v_cndmask_b32 e64 (sgpr pair mask)", "u32", "v_cndmask_b32 {c}, {c}, %8, s[20:21]"),
    ("v_cndmask_b32 (vcc), dst != src", "u32", None),
    ("v_cmp_lt_u32 + v_cndmask (vcc)", "u32", None),
    ("v_cmp_lt_u32 + 7 v_cndmask (vcc)", "u32", None),
    ("v_addc_co_u32 (writes vcc) + 7 v_cndmask (vcc)", "u32", None),
    ("v_bfi_b32", "u32", "v_bfi_b32 {c}, %8, {c}, %9"),
    ("v_sub_u32", "u32", "v_sub_u32 {c}, %8, {c}"),
    ("v_lshlrev_b32", "u32", "v_lshlrev_b32 {c}, 1, {c}"),
    ("v_or_b32", "u32", "v_or_b32 {c}, %8, {c}"),
    ("v_mad_u64_u32 (8 chains)", "u64", "v_mad_u64_u32 {c}, s[20:21], %8, %9, {c}"),
    ("v_mad_u64_u32 (1 chain)", "u64", "v_mad_u64_u32 %0, s[20:21], %8, %9, %0"),
    ("v_mad_u64_u32 (2 chains)", "u64", None),
    ("v_mad_u64_u32 sgpr operand", "u64", "v_mad_u64_u32 {c}, s[20:21], %8, s22, {c}"),
    ("v_mul_lo_u32", "u32", "v_mul_lo_u32 {c}, %8, {c}"),
    ("v_mul_hi_u32", "u32", "v_mul_hi_u32 {c}, %8, {c}"),
    ("v_mad_u32_u24", "u32", "v_mad_u32_u24 {c}, %8, %9, {c}"),
    ("v_mul_u32_u24", "u32", "v_mul_u32_u24 {c}, %8, {c}"),
    ("v_mul_hi_u32_u24", "u32", "v_mul_hi_u32_u24 {c}, %8, {c}"),
    ("v_lshl_add_u64", "u64", "v_lshl_add_u64 {c}, {c}, 0, %[e]"),
    ("v_lshrrev_b64", "u64", "v_lshrrev_b64 {c}, 1, {c}"),
    ("v_lshlrev_b64", "u64", "v_lshlrev_b64 {c}, 1, {c}"),
    ("v_fma_f32", "f32", "v_fma_f32 {c}, {c}, %8, %9"),
    ("v_pk_fma_f32", "f64bits", "v_pk_fma_f32 {c}, {c}, %[e], %[e]"),
    ("v_fma_f64", "f64", "v_fma_f64 {c}, {c}, %[m], %[k]"),
    ("v_mul_f64", "f64", "v_mul_f64 {c}, {c}, %[m]"),
    ("v_add_f64", "f64", "v_add_f64 {c}, {c}, %[k]"),
    ("v_cvt_f64_u32 + back", "u32", None),
    ("v_dot4_u32_u8", "u32", "v_dot4_u32_u8 {c}, %8, %9, {c}"),
    ("s_nop 0", "u32", "s_nop 0"),
    ("s_nop 3", "u32", "s_nop 3"),
    ("mad + addc alternating", "mix", None),
    ("mad + 2 plain alternating", "mix", None),
]


def block(kind):
    name, ty, t = kind
    lines = []
    if name == "v_mad_u64_u32 (2 chains)":
        for r in range(16):
            lines += ["v_mad_u64_u32 %0, s[20:21], %8, %9, %0", "v_mad_u64_u32 %1, s[20:21], %8, %9, %1"]
    elif name == "v_cndmask_b32 (vcc), dst != src":
        for r in range(4):
            for i in range(8):
                lines.append(f"v_cndmask_b32 %{i}, %8, %9, vcc")
    elif name == "v_cmp_lt_u32 + v_cndmask (vcc)":
        for r in range(2):
            for i in range(8):
                lines += [f"v_cmp_lt_u32 vcc, %8, %{i}", f"v_cndmask_b32 %{i}, %{i}, %9, vcc"]
    elif name == "v_cmp_lt_u32 + 7 v_cndmask (vcc)":
        # what a compiled select of a multi-limb value looks like: one compare, then a run of selects on its VCC
        for r in range(4):
            lines.append(f"v_cmp_lt_u32 vcc, %8, %{r}")
            for i in range(7):
                lines.append(f"v_cndmask_b32 %{(r + i + 1) % 8}, %{(r + i + 1) % 8}, %9, vcc")
    elif name == "v_addc_co_u32 (writes vcc) + 7 v_cndmask (vcc)":
        for r in range(4):
            lines.append(f"v_addc_co_u32 %{r}, vcc, %8, %{r}, vcc")
            for i in range(7):
                lines.append(f"v_cndmask_b32 %{(r + i + 1) % 8}, %{(r + i + 1) % 8}, %9, vcc")
    elif name == "v_cvt_f64_u32 + back":
        # 16 x (u32 -> f64 -> u32) on 8 chains, temp pair v[100:101] .. per chain
        for r in range(2):
            for i in range(8):
                lines += [f"v_cvt_f64_u32 v[{100 + 2 * i}:{101 + 2 * i}], %{i}", f"v_cvt_u32_f64 %{i}, v[{100 + 2 * i}:{101 + 2 * i}]"]
    elif name == "mad + addc alternating":
        # 16 mads on chains 0..3 (64-bit) interleaved with 16 addc on the low halves of chains 4..7
        for r in range(4):
            for i in range(4):
                lines += [f"v_mad_u64_u32 %{i}, s[20:21], %8, %9, %{i}", f"v_addc_co_u32 %{4 + i}, vcc, %8, %{4 + i}, vcc"]
    elif name == "mad + 2 plain alternating":
        for r in range(3):
            for i in range(4):
                lines += [f"v_mad_u64_u32 %{i}, s[20:21], %8, %9, %{i}", f"v_add_u32 %{4 + i}, %8, %{4 + i}", f"v_xor_b32 %{4 + i}, %9, %{4 + i}"]
        lines = lines[:32]
    else:
        for r in range(4):
            for i in range(8):
                lines.append(t.format(c=f"%{i}"))
    return lines


def main():
    out = []
    out.append("// GENERATED by gen_instr_rates.py -- do not edit.\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <cstring>\n")
    for k, kind in enumerate(KINDS):
        name, ty, t = kind
        cty = {"u32": "uint32_t", "u64": "uint64_t", "f32": "float", "f64": "double", "f64bits": "uint64_t", "mix": "uint64_t"}[ty]
        lines = block(kind)
        body = "\\n\\t\"\n            \"".join(lines)
        # %L / %H select halves of a 64-bit operand
        body = body.replace("%L", "%L").replace("%H", "%H")
        init = "; ".join(f"{cty} c{i} = ({cty})(tid + {i})" for i in range(8))
        if ty == "mix":
            init = "; ".join(f"uint64_t c{i} = tid + {i}" for i in range(4)) + "; " + "; ".join(f"uint32_t c{i} = tid + {i}" for i in range(4, 8))
        if ty == "f64bits":
            init = "; ".join(f"uint64_t c{i} = 0x3F8000003F800000ull + tid + {i}" for i in range(8))
        clob = '"vcc", "s20", "s21", "s22"' + (", " + ", ".join(f'"v{r}"' for r in range(100, 116)) if "cvt" in name else "")
        out.append(f"""__global__ __launch_bounds__(256) void k{k}(int iters, uint32_t* sink) {{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t a = tid * 2654435761u + 12345u, b = (tid ^ 0x9E3779B9u) | 1u;
    uint64_t e = 0x3F8000013F800001ull; double m = 1.0000001, kk = 1e-9;
    {init};
    asm volatile("s_mov_b32 s22, 0x3d1\\n\\ts_mov_b64 s[20:21], 0" ::: "s20", "s21", "s22");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {{
        asm volatile("{body}"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
            : "v"(a), "v"(b), [e] "v"(e), [m] "v"(m), [k] "v"(kk)
            : {clob});
    }}
    const uint64_t t1 = __builtin_readcyclecounter();
    uint64_t x = 0; {{ uint64_t w; w = 0; memcpy(&w, &c0, sizeof(c0)); x ^= w; w = 0; memcpy(&w, &c1, sizeof(c1)); x ^= w; w = 0; memcpy(&w, &c2, sizeof(c2)); x ^= w; w = 0; memcpy(&w, &c3, sizeof(c3)); x ^= w;
      w = 0; memcpy(&w, &c4, sizeof(c4)); x ^= w; w = 0; memcpy(&w, &c5, sizeof(c5)); x ^= w; w = 0; memcpy(&w, &c6, sizeof(c6)); x ^= w; w = 0; memcpy(&w, &c7, sizeof(c7)); x ^= w; }}
    if (tid == 0) {{ sink[64] = (uint32_t)(t1 - t0); sink[65] = (uint32_t)((t1 - t0) >> 32); }}
    if (x == 0x123456789ull) sink[tid & 63] = (uint32_t)x;
}}
""")
    names = ", ".join('"' + k[0] + '"' for k in KINDS)
    launches = "\n".join(f"            case {k}: hipLaunchKernelGGL(k{k}, dim3(blocks), dim3(256), 0, 0, iters, sink); break;" for k in range(len(KINDS)))
    out.append(f"""int main() {{
    uint32_t* sink; (void)hipMalloc(&sink, 4096); (void)hipMemset(sink, 0, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {{{names}}};
    const int nk = {len(KINDS)}, iters = 40000;
    printf("%-40s %s\\n", "instruction (32 per trip)", "ns per wave-instr per SIMD  [ticks]  at 1 / 2 / 4 / 8 waves per SIMD");
    for (int k = 0; k < nk; k++) {{
        printf("%-40s", names[k]);
        for (int wps = 1; wps <= 8; wps *= 2) {{
            const int blocks = 256 * wps;          // 256 CUs x 4 SIMDs, one 256-thread block = 4 waves = 1 wave per SIMD of a CU
            float best = 1e30f; uint64_t ticks = 0;
            for (int rep = 0; rep < 3; rep++) {{
                (void)hipEventRecord(e0);
                switch (k) {{
{launches}
                }}
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) {{ best = ms; uint32_t t[2]; (void)hipMemcpy(t, sink + 64, 8, hipMemcpyDeviceToHost); ticks = ((uint64_t)t[1] << 32) | t[0]; }}
            }}
            const double instr_per_simd = 32.0 * iters * wps;
            printf("  %6.3f [%5.2f]", best * 1e6 / instr_per_simd, (double)ticks / (32.0 * iters));
        }}
        printf("\\n");
    }}
    return 0;
}}
""")
    print("\n".join(out))


if __name__ == "__main__":
    main()
