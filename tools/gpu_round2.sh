# Round-2 measurement set on the GPU box: headline profile round, BASELINE config 4, v1, batch sizes.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_profile_round.sh 3584 r02 > gpurun_out/r02_round.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --config 4fsk --captures 1024 --max-iter 50 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_config4_b1024.json
python bench.py --config v1 --captures 3584 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r02_bench_v1_b3584.json
for B in 16 256 512 768 1536 2048; do
  python bench.py --captures $B --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r02_bench_b$B.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["kernel_ms"], d["roofline"]["kernel"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -40 gpurun_out/r02_round.log
