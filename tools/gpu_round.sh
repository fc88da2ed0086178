# development aid: one GPU-box round = bench + rocprofv3 kernel stats + PMC pass (outputs under gpurun_out/)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=${1:-512}
python bench.py --captures $B --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_b$B.json; cat gpurun_out/bench_b$B.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof/r01_kernel_stats.csv | cut -c1-200
# HBM traffic: separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), small batch to keep it short
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --captures 64 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --captures 64 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{root}/{tag}/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        acc = {}
        for r in rows:
            k = r.get("Kernel_Name", "")[:40]
            if "wenet" not in k: continue
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            print(tag, k, c, "n=%d" % len(v), "mean=%.1f" % (sum(v) / len(v)), "max=%.1f" % max(v))
PY
