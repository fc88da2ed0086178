# where the wavefronts' cycles go (SQ wait / active breakdown) for the batch demodulator at the bench batch; PMC_ENV="WENET_RX_OCT=7 WENET_RX_OCT_ND=1" to force a variant
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | md5sum | cut -c1-6)
  env $PMC_ENV rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_w_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --captures 3584 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_w_$n.log 2>&1
done
python - <<'PY'
import csv,glob,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
acc={}
for f in glob.glob(root+"/pmc_w_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "oct" not in k: continue
        acc[r["Counter_Name"]]=max(acc.get(r["Counter_Name"],0), float(r["Counter_Value"]))
wc=acc.get("SQ_WAVE_CYCLES",1)
for k,v in sorted(acc.items()): print(f"{k:28s} {v:.4g}  {v/wc:.3f} of wave cycles")
fr=3584*20000
print("per frame and capture: valu",acc.get("SQ_INSTS_VALU",0)/fr,"salu",acc.get("SQ_INSTS_SALU",0)/fr,"lds",acc.get("SQ_INSTS_LDS",0)/fr,"vmem rd",acc.get("SQ_INSTS_VMEM_RD",0)/fr,"vmem wr",acc.get("SQ_INSTS_VMEM_WR",0)/fr)
PY
