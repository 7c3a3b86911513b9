# per-dispatch time line of one TransformerBlock fwd+bwd at every level -> gpurun_out/block_trace${TAG}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bt_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/bt_prof -o run -- python $GRAFT_REPO_ROOT/scripts/block_trace.py run > $GRAFT_REPO_ROOT/gpurun_out/block_trace_run${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/block_trace.py report $(find /tmp/bt_prof -name '*kernel_trace.csv' | head -1) > gpurun_out/block_trace${TAG}.txt
tail -5 gpurun_out/block_trace${TAG}.txt
