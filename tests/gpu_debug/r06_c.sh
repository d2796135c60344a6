#!/bin/bash
# round 6: where does bench.py hang?  (faulthandler dumps the Python stack when timeout sends SIGABRT); then the base-8 pricing experiments
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06c
export TMPDIR=/tmp
timeout -s ABRT 150 python3 -X faulthandler bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r06c/bench_min.json 2> gpurun_out/r06c/bench_min.err; echo "bench_min rc=$?" | tee -a gpurun_out/r06c/rc.txt
timeout -s ABRT 300 python3 -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06c/bench_extras.json 2> gpurun_out/r06c/bench_extras.err; echo "bench_extras rc=$?" | tee -a gpurun_out/r06c/rc.txt
timeout -s ABRT 300 python3 -X faulthandler bench.py --steps 3 --warmup 1 --no-extras > gpurun_out/r06c/bench_cpu.json 2> gpurun_out/r06c/bench_cpu.err; echo "bench_cpu rc=$?" | tee -a gpurun_out/r06c/rc.txt
bash tests/gpu_debug/r06_b.sh
for f in bench_min bench_extras bench_cpu; do echo "== $f"; tail -n 25 gpurun_out/r06c/$f.err | cut -c1-300; done
