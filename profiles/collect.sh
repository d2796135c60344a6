#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repo root):
#     gpurun -- 'bash profiles/collect.sh r04'
# then, back in the container:  bash profiles/finish.sh r04       (copies the summaries into profiles/, regenerates the ISA mix, summarises the counter passes)
# Counters go in their own passes, each with --kernel-trace only (never with sys/hip/hsa traces).  The profiled runs take --no-probe: the issue-rate probe kernels
# (k_microbench) would otherwise be half of the trace.
R=${1:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "import zk_nullifier_sig_amd as p; print(p.Engine(0).version())" > gpurun_out/build_$R.txt 2>/dev/null
# (the profiled runs use --in-flight 1: one call after the other, so that per-kernel durations and counters mean one kernel on the machine; the unprofiled lines above are the default, two batches in flight)
# the metric workload alone (every k_verify_* / k_tab_* launch is a 2^20-item V1 launch, so the per-kernel averages are comparable with bench.py's stage_ms) ...
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o $R -- python3 bench.py --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-probe > gpurun_out/bench_${R}_prof.log 2>&1
# ... the signer (BASELINE config 3) ...
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}s -o ${R}s -- python3 bench.py --config 3 --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-probe > gpurun_out/bench_${R}s_prof.log 2>&1
# ... and with the secondary workloads (V2, SEC1, verify_non_zk, nullifier set, host-pointer pipeline)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}x -o ${R}x -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-probe > gpurun_out/bench_${R}x_prof.log 2>&1
#  the modes the headline and config 4 actually use: the DEFAULT run (two batches in flight) under the kernel trace, so that roofline.kernel_ms_in_timed_region
# has a profile behind it, and one GPU's share of config 4 under the 8-way split (2^19 V2 verifies per step) with its own bench line
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}i -o ${R}i -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe > gpurun_out/bench_${R}i_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}c4 -o ${R}c4 -- python3 bench.py --config 4 --gpus 1 --log2-batch 19 --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-probe > gpurun_out/bench_${R}c4_prof.log 2>&1
# the host-pointer pipeline's own timeline, no profiler attached (PLUME_HOST_TRACE: timing events on the library's three streams)
PLUME_HOST_TRACE=1 python3 tests/gpu_debug/host_trace.py 20 both > gpurun_out/host_trace_$R.txt 2>&1
for spec in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tg=${spec%%:*}; cn=${spec#*:}
  rocprofv3 --pmc $cn --kernel-trace --output-format csv -d gpurun_out/pmc_$tg -o $tg -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-probe > gpurun_out/pmc_$tg.log 2>&1
done
# the signer's counters (summarize_pmc.py folds its k_sign_* kernels into the same profiles/<round>_pmc_summary.json)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sqs -o sqs -- python3 bench.py --config 3 --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-probe > gpurun_out/pmc_sqs.log 2>&1
# Round 5: the counters are summarised ON THE BOX before the unprofiled runs, so that the committed bench lines carry the traffic / instruction counts of THIS build
# (`traffic_source.same_build`, `stage_roofline._same_build` true) from one invocation
bash profiles/finish.sh $R > gpurun_out/finish_on_box_$R.log 2>&1
python3 bench.py > gpurun_out/bench_$R.log 2> gpurun_out/bench_$R.err
python3 bench.py --config 3 > gpurun_out/bench_${R}_sign.log 2> gpurun_out/bench_${R}_sign.err
python3 bench.py --config 4 --gpus 1 --log2-batch 19 --no-cpu-baseline > gpurun_out/bench_${R}_c4share.log 2> gpurun_out/bench_${R}_c4share.err
tail -1 gpurun_out/bench_$R.log | cut -c1-400
head -8 gpurun_out/prof_$R/${R}_kernel_stats.csv | cut -c1-130
