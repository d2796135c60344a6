import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as p
e=p.Engine(0)
names={0:"v_mad_u64_u32",1:"v_addc_co_u32",2:"v_mul_lo_u32",3:"v_mad_u32_u24",4:"v_add_u32",7:"v_fma_f64",8:"v_lshl_add_u64",5:"fe_mul",6:"fe_sqr"}
for k,nm in names.items():
    it=(1<<18) if k not in (5,6) else (1<<13)
    e.microbench(k,it//4)
    r=e.microbench(k,it); t,ms=e.microbench_ticks()
    ops_lane=(8 if k not in (5,6) else 2)*it
    # 8 waves per SIMD: per-SIMD issue cycles per wave-instruction = ticks / (ops_lane*8 waves)
    print(f"{nm:16s} {r:.3e} ops/s  kernel {ms:.3f} ms  wave0 ticks {t:.0f}  ticks/op/wave {t/ops_lane:.2f}  (x1/8 if 8 waves share a SIMD: {t/ops_lane/8:.2f})  tick rate {t/ms/1e3:.1f} MHz")
