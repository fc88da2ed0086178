# Round-3 measurement set on the GPU box: headline profile round, BASELINE config 4 (batch + single stream), v1, batch sizes, soaks.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_profile_round.sh 3584 r03 > gpurun_out/r03_round.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile_round.sh 1024 r03c4 --config 4fsk --max-iter 50 > gpurun_out/r03c4_round.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --config v1 --captures 3584 --no-extras 2>/dev/null | tail -1 > gpurun_out/r03_bench_v1_b3584.json
python bench.py --config 4fsk --captures 1 --max-iter 50 --no-extras 2>/dev/null | tail -1 > gpurun_out/r03_bench_config4_b1.json
for B in 16 256 768 1536 2048; do
  python bench.py --captures $B --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r03_bench_b$B.json
done
{
  echo "# Parity soak, round 3 (tools/soak.py, tools/soak_short.py): random captures (Eb/N0 3-15 dB, clock error 0 or +-1500 ppm, 1-13 packets) through the GPU chain,"
  echo "# soft decisions and packets compared bit for bit with the CPU oracle.  Host-fed batches run in time slices (forced short here: WENET_RX_SLICE_SAMPLES)."
  echo "## batch demodulator, 7 captures per workgroup, v2 + v1, slices of 20 000 samples"
  WENET_RX_OCT=7 WENET_RX_SLICE_SAMPLES=20000 python tools/soak.py 500 31 2>&1 | tail -2
  echo "## batch demodulator, 4 per workgroup, one launch per capture set"
  WENET_RX_OCT=4 WENET_RX_NO_SLICES=1 python tools/soak.py 400 32 2>&1 | tail -2
  echo "## default kernel choice (pipelined kernels), slices of 50 000 samples"
  WENET_RX_SLICE_SAMPLES=50000 python tools/soak.py 200 33 2>&1 | tail -2
  echo "## 4-FSK Ts 32: four captures + chain wave + sum wave per workgroup; one capture + two duty waves"
  WENET_RX_OCT=4 WENET_RX_OCT_ND=2 python tools/soak.py 120 34 4fsk 2>&1 | tail -2
  WENET_RX_OCT=1 WENET_RX_OCT_ND=2 WENET_RX_OCT_HLP=0 WENET_RX_SLICE_SAMPLES=200000 python tools/soak.py 60 35 4fsk 2>&1 | tail -2
  echo "## 4-FSK Ts 32: one capture + three tone helpers + two duty waves (the single-stream form)"
  WENET_RX_OCT=1 WENET_RX_OCT_ND=2 WENET_RX_OCT_HLP=1 WENET_RX_SLICE_SAMPLES=300000 python tools/soak.py 80 36 4fsk 2>&1 | tail -2
  WENET_RX_OCT=1 WENET_RX_OCT_ND=2 WENET_RX_OCT_HLP=1 WENET_RX_NO_SLICES=1 python tools/soak.py 40 37 4fsk 2>&1 | tail -2
  echo "## captures of 0..6 frames"
  WENET_RX_OCT=7 python tools/soak_short.py 2>&1 | tail -1
} > gpurun_out/r03_soak.txt 2>&1
python tools/sweep.py --config v2 --n 3584 --bins 17 --check-cpu 6 > gpurun_out/r03_config3_sweep3584_v2.md 2>&1
for c in v1 v2; do python tools/robustness.py --config $c > gpurun_out/r03_robustness_$c.md 2>&1; done
python tools/host_feed.py 768 10 > gpurun_out/r03_host_feed.txt 2>&1; python tools/host_feed.py 256 10 >> gpurun_out/r03_host_feed.txt 2>&1; python tools/host_feed.py 3584 10 >> gpurun_out/r03_host_feed.txt 2>&1
{ echo "# cycle stamps of the batch demodulator (instrumented build, tools/prof_build.sh; ~10 % slower than the product), per frame"
  echo "## 3584 captures x 2 s, seven captures + one duty wave per workgroup, two workgroups per CU"; WENET_RX_LIB=tools/prof_build/libwenet_rx.so python tools/gpu_oct_prof.py 3584 2 7 v2 2>&1 | grep -v amdgpu.ids | head -9
  echo "## 1792 captures: one workgroup per CU"; WENET_RX_LIB=tools/prof_build/libwenet_rx.so python tools/gpu_oct_prof.py 1792 2 7 v2 2>&1 | grep -v amdgpu.ids | head -2
  echo "## product build, the same two"; python tools/gpu_oct_prof.py 3584 2 7 v2 2>&1 | grep kernel; python tools/gpu_oct_prof.py 1792 2 7 v2 2>&1 | grep kernel
} > gpurun_out/r03_oct_stamps.txt 2>&1
python tools/gpu_allout.py v2 3584 2 8 > gpurun_out/r03_allout.txt 2>&1; python tools/gpu_allout.py 4fsk 1024 2 8 >> gpurun_out/r03_allout.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03*_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["kernel_ms"], d["roofline"]["kernel"], d["roofline"].get("frac"), d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/r03_soak.txt; cat gpurun_out/r03_host_feed.txt; cat gpurun_out/r03_allout.txt
