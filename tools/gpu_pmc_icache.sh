# development aid: instruction-cache and branch counters of the demod kernel
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_ic -o s -- python $GRAFT_REPO_ROOT/bench.py --captures ${1:-768} --steps 1 --warmup 0 --no-cpu-baseline --no-extras ${2:-} > /tmp/pmc_ic.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/pmc_ic/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:36]
        if "wenet_demod" not in k and "wenet_decode" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "n=%d" % len(v), "max=%.4g" % max(v))
PY
tail -3 /tmp/pmc_ic.log | cut -c1-300
