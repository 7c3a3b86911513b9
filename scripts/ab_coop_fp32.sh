# A/B of the exact-fp32 in-place cooperative LayerNorm of the pixel-reduction kernel (RCOT_NT_COOP bit 2), one gpurun call
OUT=${1:-gpurun_out/ab_coop_fp32.txt}
: > $OUT
echo "== kernel tests with RCOT_NT_COOP=7" >> $OUT
RCOT_NT_COOP=7 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_x3_gpu.py -x -q 2>&1 | tail -3 >> $OUT
for c in 3 7; do
  echo "== RCOT_NT_COOP=$c: products fp32 (cold operands)" >> $OUT
  RCOT_NT_COOP=$c BWD3_PRECS=fp32 X3_SHAPES=0,1,4 python scripts/bench_bwd3.py 2>/dev/null | grep -v "^$" >> $OUT
done
for c in 3 7 3 7; do
  echo -n "RCOT_NT_COOP=$c fp32 ms/iteration: " >> $OUT
  RCOT_NT_COOP=$c python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done
cat $OUT
