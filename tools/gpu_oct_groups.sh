# development aid: captures per workgroup of the batch demodulator (WENET_RX_OCT) against how many workgroups a CU then holds; v2, 2 s captures
cd $GRAFT_REPO_ROOT
run() { python bench.py --captures $1 --seconds 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); B=d['config']['captures_per_gpu']; ns=d['config']['samples_per_capture']; print('  demod %.2f ms = %.1f G samples/s demod-only' % (d['kernel_ms']['demod'], B*ns/d['kernel_ms']['demod']/1e6))"; }
for spec in "7 3584" "4 3072" "4 2048" "3 3072" "2 2560" "5 2560" "6 3072"; do set -- $spec; echo "G=$1, $2 captures"; WENET_RX_OCT=$1 run $2; done
