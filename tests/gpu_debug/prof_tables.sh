export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3c/kt.log 2>&1
for spec in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tg=${spec%%:*}; cn=${spec#*:}
  rocprofv3 --pmc $cn --kernel-trace --output-format csv -d gpurun_out/r3c/pmc_$tg -o $tg -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3c/pmc_$tg.log 2>&1
done
find gpurun_out/r3c -name "*kernel_stats.csv" | head -1 | xargs head -25 | cut -c1-150
