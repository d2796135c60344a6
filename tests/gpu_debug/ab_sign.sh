# signer A/B on one box: the in-tree build (A) against libplume_hip_<x>.so:   bash tests/gpu_debug/ab_sign.sh head
for rep in 1 2 3; do
  for v in A "$@"; do
    if [ $v = A ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep$rep', d['value'], d.get('value_serial'), d['ms_per_step'], d['stage_ms'])"
  done
done
