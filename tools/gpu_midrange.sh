# development aid (round 5): mid-range batches (768..3584 captures) through forced workgroup shapes of the batch demodulator; needs tools/variants/nd2 (-DWO_SMALL_ND2)
# usage: gpu_midrange.sh "<captures> ..." [seconds]
cd $GRAFT_REPO_ROOT
S=${2:-4}
for B in ${1:-"1024 1536 1792 2048 2560"}; do
  for cfg in "default:" "g7nd1:WENET_RX_OCT=7" "g4nd1:WENET_RX_OCT=4" "g6nd2:WENET_RX_OCT=6 WENET_RX_OCT_ND=2" "g7nd2:WENET_RX_OCT=7 WENET_RX_OCT_ND=2" "g4nd2:WENET_RX_OCT=4 WENET_RX_OCT_ND=2" "g5nd2:WENET_RX_OCT=5 WENET_RX_OCT_ND=2" "g3nd2:WENET_RX_OCT=3 WENET_RX_OCT_ND=2"; do
    n=${cfg%%:*}; e=${cfg#*:}
    r=$(env $e WENET_RX_LIB=tools/variants/nd2/libwenet_rx.so python bench.py --captures $B --seconds $S --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('demod %.1f ms  %s  value %.1f G/s' % (d['kernel_ms']['demod'], d['roofline']['kernel'][:48], d['value']/1e3))
except Exception as ex: print('failed', ex)")
    echo "B=$B $n: $r"
  done
done
