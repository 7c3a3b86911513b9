# A/B of library builds in situ: ms/step of the default bench (side stream off) for each RCOT_LIB given
for lib in "$@"; do
  for rep in 1 2; do
    RCOT_LIB=$lib RCOT_OVERLAP=0 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
  done
done
