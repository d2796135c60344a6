# A/B in one process-sequence on one box: alternate the in-tree build (A) and libplume_hip_b.so (B)
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_b.so; else unset PLUME_HIP_LIB; fi
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep$rep', d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done
