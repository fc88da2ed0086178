# batch sizes that are not a multiple of the 3584 resident captures (VERDICT r03 task 2): bench lines under gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in "$@"; do
  python bench.py --captures $B --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> gpurun_out/r04_bench_b$B.err | tail -1 > gpurun_out/r04_bench_b$B.json
  python - $B <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r04_bench_b{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms"], d["packets_valid_per_step_rank0"])
PY
done
