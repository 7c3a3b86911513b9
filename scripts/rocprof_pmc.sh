# per-kernel SQ counters of a TINY script (counters serialise every dispatch; never point this at a whole step)
# usage: PMC_SCRIPT=pmc_x3.py [PMC="SQ_... ..."] [KPAT="gemm_x3"] bash scripts/rocprof_pmc.sh  -> gpurun_out/pmc_report${TAG}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE}
timeout 240 rocprofv3 --kernel-trace --pmc $PMC -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o run -- python $GRAFT_REPO_ROOT/scripts/${PMC_SCRIPT:-pmc_gemm.py} > $GRAFT_REPO_ROOT/gpurun_out/pmc_run.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_report.py gpurun_out/pmc $KPAT > gpurun_out/pmc_report${TAG}.txt 2> gpurun_out/pmc_report${TAG}.err
rm -rf gpurun_out/pmc
cat gpurun_out/pmc_report${TAG}.txt | head -${HEAD:-60}
