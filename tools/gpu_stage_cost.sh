# development aid: instruction counts of the pipelined demod kernel with one stage left out at a time (WENET_RX_DBG_SKIP:
# 1 chain, 2 estimator, 4 mix+integrate waves, 8 timing/decision wave; inside the D waves: 16 mix, 32 integrate, 64 sample staging).  Results of those runs are garbage by construction;
# the differences to the full run are the stages' shares.  Needs a development build of the library:
#   make -C wenet_amd/csrc clean all EXTRA=-DWR_DBG_SKIP   (production builds compile the switch out)
cd /tmp && export TMPDIR=/tmp
for skip in ${2:-0 1 2 4 8 15 16 32 64}; do
  export WENET_RX_DBG_SKIP=$skip
  rm -rf /tmp/pmc_st
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_st -o s -- python $GRAFT_REPO_ROOT/bench.py --captures ${1:-768} --steps 1 --warmup 0 --no-cpu-baseline --no-single-stream > /tmp/pmc_st.log 2>&1
  python - <<PY
import csv, glob
acc = {}
for f in glob.glob("/tmp/pmc_st/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "demod_pipe" in r.get("Kernel_Name", "") or "demod_tri" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] = max(acc.get(r["Counter_Name"], 0), float(r["Counter_Value"]))
fr = ${1:-768} * 20000.0          # v2: 10 s x 96000 symbols/s / 48 symbols per modem frame
print("skip=$skip", " ".join(f"{k}={v / fr:.0f}/frame" for k, v in sorted(acc.items())))
PY
done
