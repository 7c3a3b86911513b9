# GPU clocks / power while the bench runs: samples rocm-smi every ~0.3 s next to `python bench.py --steps 400` (exact fp32, then bf16x3),
# and once idle before it.  Output: gpurun_out/clocks_under_load.txt
OUT=${1:-gpurun_out/clocks_under_load.txt}
smp() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; }
{ echo "== idle (sclk level | socket power W)"; smp; } > $OUT
for prec in fp32 bf16x3; do
  python bench.py --prec $prec --steps 400 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/clk_bench.json 2> /tmp/clk_bench.err &
  BP=$!
  # wait for the timed loop: the bench logs "warmup done" when it starts
  for i in $(seq 1 120); do grep -q "warmup done" /tmp/clk_bench.err 2>/dev/null && break; sleep 0.5; done
  sleep 3
  echo "== under load: bench.py --prec $prec, timed steps running" >> $OUT
  for i in $(seq 1 16); do
    kill -0 $BP 2>/dev/null || break
    smp >> $OUT
    sleep 0.3
  done
  wait $BP
  grep "timed" /tmp/clk_bench.err >> $OUT
done
cat $OUT
