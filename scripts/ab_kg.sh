# A/B of the eight-wavefront k-group form of the 64x64 exact-fp32 kernel (RCOT_XX_KG=0: round-4 dispatch), one gpurun call
OUT=${1:-gpurun_out/ab_kg.txt}
: > $OUT
for kg in 0 1; do
  echo "== RCOT_XX_KG=$kg: products (scripts/bench_bwd3.py, fp32, cold operands)" >> $OUT
  RCOT_XX_KG=$kg BWD3_PRECS=fp32 X3_SHAPES=6,7,8,9,10 python scripts/bench_bwd3.py 2>/dev/null | grep -v "^$" >> $OUT
  echo "== RCOT_XX_KG=$kg: blocks (scripts/small_levels.py, fp32)" >> $OUT
  RCOT_XX_KG=$kg RCOT_GEMM_PREC=fp32 python scripts/small_levels.py 30 2>/dev/null >> $OUT
done
for kg in 0 1 0 1; do
  echo "== RCOT_XX_KG=$kg: whole iteration" >> $OUT
  RCOT_XX_KG=$kg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done
cat $OUT
