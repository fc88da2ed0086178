# development aid: LDS activity / bank-conflict counters of our kernels
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_lds -o s -- python $GRAFT_REPO_ROOT/bench.py --captures ${1:-768} --steps 1 --warmup 0 --no-cpu-baseline --no-single-stream > /tmp/pmc_lds.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/pmc_lds/*counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:36]
        if "wenet_demod" not in k and "wenet_decode" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "max=%.4g" % max(v))
PY
tail -2 /tmp/pmc_lds.log | cut -c1-200
