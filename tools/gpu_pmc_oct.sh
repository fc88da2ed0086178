# SQ counters of the demod kernel at a bench-like batch (separate rocprofv3 --pmc run, no trace domains).  usage: gpu_pmc_oct.sh [captures] [seconds] [extra bench args]
cd /tmp && export TMPDIR=/tmp
B=${1:-3584}; S=${2:-2}; shift; shift
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_oct -o s -- python $GRAFT_REPO_ROOT/bench.py --captures $B --seconds $S --steps 1 --warmup 0 --no-cpu-baseline --no-single-stream "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_oct.log 2>&1
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for f in glob.glob(f"{root}/pmc_oct/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = {}
    for r in rows:
        k = r.get("Kernel_Name", "")[:40]
        if "demod" not in k and "decode" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "n=%d" % len(v), "max=%.4g" % max(v))
PY
tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/pmc_oct.log
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_oct
