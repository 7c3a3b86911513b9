# round 5 A/B of the launch-cutting changes (LayerNorm statistics inside the exact-fp32 projection kernel, dV / dQ / dK from one launch) on the per-block times (scripts/small_levels.py, launch-plan replay), one gpurun call:
#   old = RCOT_XX_LN_COMP=0 RCOT_MULTI=0 (round 4's schedule), new = defaults
# usage: bash scripts/ab_r05.sh [out file]
OUT=${1:-gpurun_out/ab_r05.txt}
: > $OUT
for prec in fp32 bf16x6 bf16x3; do
  echo "== $prec, round-4 schedule" >> $OUT
  RCOT_GEMM_PREC=$prec RCOT_XX_LN_COMP=0 RCOT_MULTI=0 python scripts/small_levels.py 30 2>/dev/null >> $OUT
  echo "== $prec, round-5 schedule" >> $OUT
  RCOT_GEMM_PREC=$prec python scripts/small_levels.py 30 2>/dev/null >> $OUT
done
cat $OUT
