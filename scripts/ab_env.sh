# in-situ A/B: ms/step of the default bench for each environment setting given (quote each: "A=1 B=2"); two runs each
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'])"
  done
done
