# plain (no canary) reproducibility runs of library variants; usage: gpu_r05_plain_repro.sh "<variant|product> ..." [passes] [captures] [seconds]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
VARS=${1:-product}; P=${2:-300}; B=${3:-3584}; S=${4:-2}
for v in $VARS; do
  if [ $v = product ]; then L=""; else L=tools/variants/$v/libwenet_rx.so; fi
  WENET_RX_LIB=$L timeout 1500 python tools/gpu_repro.py $B $S $P > gpurun_out/r05_plain_$v.txt 2>&1
  echo "== $v"; tail -2 gpurun_out/r05_plain_$v.txt | cut -c1-200
done
