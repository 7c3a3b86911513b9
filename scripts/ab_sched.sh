# A/B of the side-stream schedule (one gpurun call): weight gradients with / behind their data gradient x block close joined / deferred
OUT=${1:-gpurun_out/ab_sched.txt}
: > $OUT
for prec in fp32 bf16x6; do
for rep in 1 2; do
for wa in 0 1; do for dc in 0 1; do
  echo -n "$prec RCOT_WGRAD_AFTER=$wa RCOT_DEFER_CLOSE=$dc ms/iteration: " >> $OUT
  RCOT_WGRAD_AFTER=$wa RCOT_DEFER_CLOSE=$dc RCOT_GEMM_PREC=$prec python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done; done; done; done
cat $OUT
