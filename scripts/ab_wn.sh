OUT=gpurun_out/r05g/ab_wn.txt
mkdir -p gpurun_out/r05g; : > $OUT
for prec in bf16x6 bf16x3; do
  echo "== $prec, RCOT_X3P_WN=2 (round-4 tile rule)" >> $OUT
  RCOT_GEMM_PREC=$prec RCOT_X3P_WN=2 python scripts/small_levels.py 30 2>/dev/null >> $OUT
  echo "== $prec, occupancy rule" >> $OUT
  RCOT_GEMM_PREC=$prec python scripts/small_levels.py 30 2>/dev/null >> $OUT
done
cat $OUT
