# development aid (round 4, VERDICT r03 task 5): the parked window of the 4-FSK / Ts-32 geometry (2 W + 2 of 32 integrator outputs per symbol and tone), config 4 at 1024 captures x 2 s, 8 dB
cd $GRAFT_REPO_ROOT
run() { python bench.py --config 4fsk --captures 1024 --seconds 2 --max-iter 50 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); B=d['config']['captures_per_gpu']; ns=d['config']['samples_per_capture']; print('  demod %.2f ms = %.1f G samples/s demod-only, packets %d' % (d['kernel_ms']['demod'], B*ns/d['kernel_ms']['demod']/1e6, d['packets_valid_per_step_rank0']))"; }
echo "W = 3 (product: 8 of 32 parked)"; run
for w in ${WS:-2 4 5}; do echo "W = $w ($((2*w+2)) of 32 parked)"; WENET_RX_LIB=tools/variants/park_w$w/libwenet_rx.so run; done
