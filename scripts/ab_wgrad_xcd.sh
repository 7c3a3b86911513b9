for v in 1 0 1 0; do
  echo "== RCOT_WGRAD_XCD=$v"
  RCOT_WGRAD_XCD=$v SHAPES="4,80,128;4,128,64;8,128,64;8,256,32;8,512,16" timeout 100 python scripts/bench_conv_mprnet.py 2>&1 | tail -5 | sed 's/fwd.*| wgrad/| wgrad/'
done
for v in 1 0; do RCOT_WGRAD_XCD=$v PMC_SCRIPT=pmc_mprnet_conv.py TAG=_mprnet_xcd$v timeout 300 bash scripts/rocprof_traffic.sh 2>&1 | grep wgrad; done
bash scripts/ab_env.sh "RCOT_WGRAD_XCD=1" "RCOT_WGRAD_XCD=0" "RCOT_WGRAD_XCD=1" "RCOT_WGRAD_XCD=0" 2>&1 | tail -8
for v in 1 0 1 0; do RCOT_WGRAD_XCD=$v timeout 200 python scripts/bench_mprnet.py 20 --hip-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"mprnet xcd=$v\", d[\"hip\"][\"ms_per_iteration\"], d[\"hip\"][\"tnet_fwd_bwd_ms\"])"; done
