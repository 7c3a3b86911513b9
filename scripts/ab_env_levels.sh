# in-situ A/B of environment settings: per-block times (scripts/small_levels.py, RCOT_GEMM_PREC as given or fp32) and the iteration, interleaved, ONE gpurun call
#   bash scripts/ab_env_levels.sh "A=0" "A=1"
for rep in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg (small_levels)"
    env RCOT_GEMM_PREC=fp32 $cfg python scripts/small_levels.py 30 2>/dev/null
  done
done
for rep in 1 2 3; do
  for cfg in "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'])"
  done
done
