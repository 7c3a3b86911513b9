# round 6: a library build (RCOT_LIB=$1, e.g. the build of HEAD under build_variants/) against the tree's, inside ONE gpurun call:
# per-block times (exact fp32) and the iteration, interleaved (boxes differ by up to 4 %: only compare within a call)
BASE=$1
for rep in 1 2; do
  for lib in $BASE rcot_amd/librcot_hip.so; do
    echo "== $lib (small_levels, fp32)"
    RCOT_LIB=$PWD/$lib RCOT_GEMM_PREC=fp32 python scripts/small_levels.py 30 2>/dev/null
  done
done
for rep in 1 2 3; do
  for lib in $BASE rcot_amd/librcot_hip.so; do
    RCOT_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
  done
done
