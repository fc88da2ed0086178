# quick SQ counter pass of the batch demodulator at the bench batch (valu instructions per frame, VALU / LDS busy); PMC_EXTRA=--fast for the fast mode
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_x -o p -- python $GRAFT_REPO_ROOT/bench.py --captures 3584 --steps 1 --warmup 0 --no-cpu-baseline --no-extras $PMC_EXTRA > $OUT/pmc_x.log 2>&1
python - <<'PY'
import csv,glob,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
acc={}
for f in glob.glob(root+"/pmc_x/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "oct" not in k: continue
        acc[r["Counter_Name"]]=max(acc.get(r["Counter_Name"],0), float(r["Counter_Value"]))
print(acc)
fr=3584*20000
sc=acc["GRBM_GUI_ACTIVE"]/8*1024
print("valu/frame",acc["SQ_INSTS_VALU"]/fr,"salu/frame",acc.get("SQ_INSTS_SALU",0)/fr,"valu_busy",acc["SQ_ACTIVE_INST_VALU"]*4/sc,"lanes",acc["SQ_THREAD_CYCLES_VALU"]/acc["SQ_ACTIVE_INST_VALU"],"lds_busy",acc["SQ_ACTIVE_INST_LDS"]*4/sc)
PY
