# Everything profiles/ holds for one round, in ONE gpurun call (comparisons inside one box): bench lines of the three workloads, replay-only
# kernel statistics of the three arithmetics, the frac check, per-block times, the level-1 block's counter traffic, the block time line.
#   bash scripts/profile_round.sh r05
R=${1:-r05}
O=gpurun_out/$R
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for prec in fp32 bf16x6 bf16x3; do
  TAG=_${R}_$prec BENCH_ARGS="--prec $prec" bash scripts/rocprof_step.sh > /dev/null 2>&1
  mv gpurun_out/kstats_${R}_$prec.txt $O/kernel_stats_$prec.txt
done
python scripts/frac_check.py $O/bench_cfg2.json $O/kernel_stats_fp32.txt > $O/frac_check.txt 2>&1
for prec in fp32 bf16x6 bf16x3; do
  RCOT_GEMM_PREC=$prec python scripts/small_levels.py 30 2>/dev/null > $O/small_levels_$prec.txt
done
python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
PMC_SCRIPT=pmc_block.py PMC_BWD=1 TAG=_${R}_block RCOT_GEMM_PREC=bf16x3 bash scripts/rocprof_traffic.sh > /dev/null 2>&1
mv gpurun_out/pmc_traffic_${R}_block.txt $O/pmc_traffic_block_fwdbwd_C96_128.txt
RCOT_GEMM_PREC=fp32 TAG=_${R}_fp32 bash scripts/block_trace.sh > /dev/null 2>&1
mv gpurun_out/block_trace_${R}_fp32.txt $O/block_trace_fp32.txt
cat $O/frac_check.txt; tail -c 400 $O/bench_cfg2.err; head -3 $O/kernel_stats_fp32.txt | cut -c1-400; cat $O/small_levels_fp32.txt
