# per-kernel SQ counters of a TINY script (counters serialise every dispatch; never point this at a whole step)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o run -- python $GRAFT_REPO_ROOT/scripts/${PMC_SCRIPT:-pmc_gemm.py} > $GRAFT_REPO_ROOT/gpurun_out/pmc_run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/pmc/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
open("gpurun_out/pmc_tables.txt","w").write("\n".join(tabs))
for t in tabs:
    if "counter" in t.lower() and "view" not in t.lower():
        cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
        print(t, cols[:14])
PY
