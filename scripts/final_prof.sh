R=r05g; O=gpurun_out/$R; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for prec in fp32 bf16x6 bf16x3; do
  TAG=_${R}_$prec BENCH_ARGS="--prec $prec" bash scripts/rocprof_step.sh > /dev/null 2>&1
  mv gpurun_out/kstats_${R}_$prec.txt $O/kernel_stats_$prec.txt
done
python scripts/frac_check.py $O/bench_cfg2.json $O/kernel_stats_fp32.txt > $O/frac_check.txt 2>&1
cat $O/frac_check.txt; tail -c 500 $O/bench_cfg2.err
