#!/bin/bash
# after tools/gpu_round4.sh (+ tools/gpu_pmc_lds.sh) on the GPU box: copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
for f in r04_allout.txt r04_bench_b1536.json r04_bench_b16.json r04_bench_b2048.json r04_bench_b256.json r04_bench_b3584.json r04_bench_b3584_100ppm.json r04_bench_b3600.json \
         r04_bench_b4000.json r04_bench_b5000.json r04_bench_b7168.json r04_bench_b768.json r04_bench_config4_b1.json r04_bench_v1_b3584.json r04_cli_times.txt \
         r04_config3_sweep3584_v2.md r04_host_feed.txt r04_kernel_stats_b3584.csv r04_pmc_b3584.json r04_soak.txt r04c4_kernel_stats_b1024.csv r04c4_pmc_b1024.json r04_lds_counters.txt r04c4_lds_counters.txt r04_soak_live.txt r04_determinism.txt; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
cp gpurun_out/r04c4_bench_b1024.json profiles/r04_bench_config4_b1024.json
sed -i 's#gpurun_out/r04#profiles/r04#g' profiles/r04_bench_*.json
python tools/isa_summary.py > profiles/r04_isa_summary.txt 2>/dev/null
python tools/check_docs.py | tail -5
