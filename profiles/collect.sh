#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repo root):
#     gpurun -- 'bash profiles/collect.sh r03'
# then, back in the container:  cp gpurun_out/prof_r03/r03_kernel_{stats,trace}.csv profiles/ ; tail -1 gpurun_out/bench_r03.log > profiles/r03_bench_line.json ;
#                               python profiles/summarize_pmc.py r03
# Counters go in their own passes, each with --kernel-trace only (never with sys/hip/hsa traces).
R=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "import zk_nullifier_sig_amd as p; print(p.Engine(0).version())" > gpurun_out/build_$R.txt 2>/dev/null
python3 bench.py > gpurun_out/bench_$R.log 2> gpurun_out/bench_$R.err
# (the profiled runs use --in-flight 1: one call after the other, so that per-kernel durations and counters mean one kernel on the machine; the unprofiled line above is the default, two batches in flight)
# the metric workload alone (every k_verify_* / k_tables launch is a 2^20-item V1 launch, so the per-kernel averages are comparable with bench.py's stage_ms) ...
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o $R -- python3 bench.py --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench_${R}_prof.log 2>&1
# ... and with the secondary workloads (signer, V2, SEC1, verify_non_zk, nullifier set, host-pointer pipeline)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}x -o ${R}x -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${R}x_prof.log 2>&1
for spec in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tg=${spec%%:*}; cn=${spec#*:}
  rocprofv3 --pmc $cn --kernel-trace --output-format csv -d gpurun_out/pmc_$tg -o $tg -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_$tg.log 2>&1
done
tail -1 gpurun_out/bench_$R.log | cut -c1-400
head -8 gpurun_out/prof_$R/${R}_kernel_stats.csv | cut -c1-130
