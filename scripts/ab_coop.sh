# A/B of the cooperative operand split of the pixel-reduction kernel (RCOT_NT_COOP: bit 0 = bf16x6, bit 1 = bf16x3), one gpurun call:
# kernel tests of the split arithmetics, the level-1 / 64x64 products cold, the blocks of the unit, the whole iteration
OUT=${1:-gpurun_out/ab_coop.txt}
: > $OUT
echo "== tests (RCOT_NT_COOP=3)" >> $OUT
RCOT_NT_COOP=3 timeout 900 python -m pytest tests/test_x3_gpu.py -x -q 2>&1 | tail -4 >> $OUT
for c in 0 3; do
  echo "== RCOT_NT_COOP=$c: products (cold operands)" >> $OUT
  RCOT_NT_COOP=$c BWD3_PRECS=x6,x3 X3_SHAPES=0,1,2,3,4,5 python scripts/bench_bwd3.py 2>/dev/null | grep -v "^$" >> $OUT
done
for c in 0 1; do
  echo "== RCOT_NT_COOP=$c: blocks bf16x6" >> $OUT
  RCOT_NT_COOP=$c RCOT_GEMM_PREC=bf16x6 python scripts/small_levels.py 30 2>/dev/null >> $OUT
done
for c in 0 1 0 1; do
  echo -n "RCOT_NT_COOP=$c bf16x6 ms/iteration: " >> $OUT
  RCOT_NT_COOP=$c RCOT_GEMM_PREC=bf16x6 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done
for c in 0 2 0 2; do
  echo -n "RCOT_NT_COOP=$c bf16x3 ms/iteration: " >> $OUT
  RCOT_NT_COOP=$c RCOT_GEMM_PREC=bf16x3 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done
cat $OUT
