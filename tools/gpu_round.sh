# development aid: one GPU-box round = smoke + bench + rocprofv3 kernel stats (outputs under gpurun_out/)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=${1:-64}
python bench.py --captures $B --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_b$B.json; cat gpurun_out/bench_b$B.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --captures $B --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
head -12 $GRAFT_REPO_ROOT/gpurun_out/prof/*kernel_stats.csv
