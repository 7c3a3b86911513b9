# GPU clocks / power while the bench runs (is the sustained MFMA peak the spec-sheet peak?): samples rocm-smi every ~0.2 s next to
# `python bench.py --steps 60`, and once idle before it.  Output: gpurun_out/clocks_under_load.txt
OUT=${1:-gpurun_out/clocks_under_load.txt}
{
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
} > $OUT
python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/clk_bench.json 2> /tmp/clk_bench.err &
BP=$!
sleep 25            # (imports, network construction, warm-up and plan recording take ~20 s)
echo "== under load (bench.py fp32 steps running)" >> $OUT
for i in $(seq 1 20); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.2
done
wait $BP
tail -c 300 /tmp/clk_bench.err >> $OUT
cat $OUT
