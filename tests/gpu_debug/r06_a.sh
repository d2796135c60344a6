#!/bin/bash
# round 6, first GPU call: the new tests, the consumers of the regenerated goldens, the small-call sweeps, a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > gpurun_out/r06a/pytest_round6.log 2>&1; echo "round6 rc=$?" | tee -a gpurun_out/r06a/rc.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu -k "golden or edge or non_zk or sec1 or fuzz" > gpurun_out/r06a/pytest_goldens.log 2>&1; echo "goldens rc=$?" | tee -a gpurun_out/r06a/rc.txt
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "half_chains or short_form" > gpurun_out/r06a/pytest_r5.log 2>&1; echo "r5 rc=$?" | tee -a gpurun_out/r06a/rc.txt
timeout 900 python3 tests/gpu_debug/r06_small_sweep.py ab > gpurun_out/r06a/small_sweep.txt 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/r06a/rc.txt
timeout 900 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r06a/rc.txt
tail -3 gpurun_out/r06a/pytest_round6.log gpurun_out/r06a/pytest_goldens.log gpurun_out/r06a/pytest_r5.log
cat gpurun_out/r06a/small_sweep.txt | tail -40
