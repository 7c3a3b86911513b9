# round 6: LayerNorm backward as the epilogue of the data gradient (RCOT_DGRAD_LN=0|1), one gpurun call: kernel test, network fixtures,
# per-block times and the iteration, exact fp32
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -x -k "dgrad_with_layernorm" 2>&1 | tail -5
python -m pytest tests/test_network_gpu.py tests/test_configs_gpu.py -q -x 2>&1 | tail -3
for v in 0 1; do
  echo "== RCOT_DGRAD_LN=$v"
  RCOT_DGRAD_LN=$v RCOT_GEMM_PREC=fp32 python scripts/small_levels.py 30 2>/dev/null
done
bash scripts/ab_env.sh "RCOT_DGRAD_LN=0" "RCOT_DGRAD_LN=1"
