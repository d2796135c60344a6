for L in 3 6 9 12; do
  PLUME_JOBS_PER_LANE=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L', d['value'], d['ms_per_step'], d['stage_ms'])"
done
