#!/bin/bash
# round 5: the three bench lines again with the final bench.py (same library build as the committed counters), then the soak run
R=r05
python3 bench.py > gpurun_out/bench_$R.log 2> gpurun_out/bench_$R.err
python3 bench.py --config 3 > gpurun_out/bench_${R}_sign.log 2> gpurun_out/bench_${R}_sign.err
python3 bench.py --config 4 --gpus 1 --log2-batch 19 --no-cpu-baseline > gpurun_out/bench_${R}_c4share.log 2> gpurun_out/bench_${R}_c4share.err
tail -c 600 gpurun_out/bench_$R.log
mkdir -p gpurun_out/r05r
timeout 1500 python3 tests/gpu_debug/soak.py 6 18 > gpurun_out/r05r/soak.txt 2>&1; tail -8 gpurun_out/r05r/soak.txt
