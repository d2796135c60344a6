#!/bin/bash
# round 6, final lines (bench.py sets GPU_MAX_HW_QUEUES=8) + the sub-batch pipeline once more now that two streams really run side by side
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06p
export TMPDIR=/tmp
for rep in 1 2; do
  for sb in 1 2 4; do
    PLUME_SUB_BATCHES=$sb timeout 300 python3 bench.py --in-flight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sub_batches=$sb rep$rep verify', d['value'], d['ms_per_step'])"
    PLUME_SUB_BATCHES=$sb timeout 300 python3 bench.py --config 3 --in-flight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sub_batches=$sb rep$rep sign', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r06p/sub_batches.txt
python3 bench.py > gpurun_out/r06p/bench_r06.log 2> gpurun_out/r06p/bench_r06.err; echo "bench rc=$?"
python3 bench.py --config 3 > gpurun_out/r06p/bench_r06_sign.log 2> gpurun_out/r06p/bench_r06_sign.err
python3 bench.py --config 4 --gpus 1 --log2-batch 19 --no-cpu-baseline > gpurun_out/r06p/bench_r06_c4share.log 2> gpurun_out/r06p/bench_r06_c4share.err
python3 bench.py --config 2 > gpurun_out/r06p/bench_r06_c2.log 2> gpurun_out/r06p/bench_r06_c2.err
tail -n 1 gpurun_out/r06p/bench_r06.log | cut -c1-300
