# A/B of a build variant of csrc/gemm_glds.hip (round 5: -DXX_DUAL=0, one accumulator per wavefront in the 64x64 kernels; earlier: ring depth) vs the
# default build, one gpurun call
OUT=${1:-gpurun_out/ab_ring.txt}
: > $OUT
for lib in build_variants/librcot_nodual.so rcot_amd/librcot_hip.so; do
  for kg in 0 1; do
    echo "== $lib RCOT_XX_KG=$kg: products (fp32, cold operands)" >> $OUT
    RCOT_LIB=$PWD/$lib RCOT_XX_KG=$kg BWD3_PRECS=fp32 X3_SHAPES=6,7,8,9,10 python scripts/bench_bwd3.py 2>/dev/null | grep -v "^$" >> $OUT
    echo "== $lib RCOT_XX_KG=$kg: blocks (fp32)" >> $OUT
    RCOT_LIB=$PWD/$lib RCOT_XX_KG=$kg RCOT_GEMM_PREC=fp32 python scripts/small_levels.py 30 2>/dev/null >> $OUT
  done
done
for lib in build_variants/librcot_nodual.so rcot_amd/librcot_hip.so build_variants/librcot_nodual.so rcot_amd/librcot_hip.so; do
  echo "== $lib: whole iteration (default switches)" >> $OUT
  RCOT_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT
done
cat $OUT
