# round 5: reproducibility of decoder variants with per-wavefront canaries (tools/gpu_repro.py); usage: gpu_r05_repro.sh "<variant> ..." [passes] [captures] [seconds]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
VARS=${1:-"c_bf96 c_t7st c_t7 c_prod"}; P=${2:-300}; B=${3:-3584}; S=${4:-2}
(hostname; cat /sys/class/drm/card*/device/unique_id 2>/dev/null; rocm-smi --showuniqueid --showbus --showserial 2>&1 | grep -v "^=\|^$") > gpurun_out/r05_box.txt 2>&1
for v in $VARS; do
  WENET_RX_LIB=tools/variants/$v/libwenet_rx.so timeout 900 python tools/gpu_repro.py $B $S $P > gpurun_out/r05_repro_$v.txt 2>&1
  echo "== $v"; head -2 gpurun_out/r05_repro_$v.txt; tail -1 gpurun_out/r05_repro_$v.txt
done
