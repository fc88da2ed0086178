# development aid (round 4): rocprofv3 kernel stats of the decode step at the bench batch with short captures
# usage: gpu_decode_prof.sh [seconds] [extra bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=${1:-2}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dec -o k -- python $GRAFT_REPO_ROOT/bench.py --seconds $S --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/prof_dec.log 2>&1
python - <<'PY'
import csv, os
f = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_dec/k_kernel_stats.csv"
for r in csv.DictReader(open(f)):
    if "wenet" in r["Name"] and "tx" not in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
