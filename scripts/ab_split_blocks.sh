for t in ${TARGETS:-768 512 384 256}; do
  echo "== RCOT_SPLIT_BLOCKS=$t"
  RCOT_SPLIT_BLOCKS=$t SHAPES="4,128,64;4,176,32;4,80,128" timeout 100 python scripts/bench_conv_mprnet.py 2>&1 | tail -3
  RCOT_SPLIT_BLOCKS=$t timeout 200 python scripts/bench_mprnet.py 20 --hip-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['hip'])"
done
