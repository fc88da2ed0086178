# Round-3 measurement set on the GPU box: headline profile round, BASELINE config 4 (batch + single stream), v1, batch sizes.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_profile_round.sh 3584 r03 > gpurun_out/r03_round.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile_round.sh 1024 r03c4 --config 4fsk --max-iter 50 > gpurun_out/r03c4_round.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --config v1 --captures 3584 --no-extras 2>/dev/null | tail -1 > gpurun_out/r03_bench_v1_b3584.json
for B in 16 256 768 1536 2048; do
  python bench.py --captures $B --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r03_bench_b$B.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03*_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["kernel_ms"], d["roofline"]["kernel"], d["roofline"].get("frac"), d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -30 gpurun_out/r03_round.log
