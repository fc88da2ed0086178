# development aid (round 4, VERDICT r03 task 3b): the small geometry (v2: 2-FSK, Ts 10) with a chain wave and a sum wave per workgroup under the one-barrier
# schedule and with the capture waves above the duty waves, against the product's one duty wave.  Needs tools/variants/small_nd2 (tools/variant_build.sh small_nd2 "-DWO_SMALL_ND2" demod_oct wenet_rx).
cd $GRAFT_REPO_ROOT
run() { python bench.py --captures $1 --seconds 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); B=d['config']['captures_per_gpu']; ns=d['config']['samples_per_capture']; print('  demod %.2f ms = %.1f G samples/s demod-only, packets %d' % (d['kernel_ms']['demod'], B*ns/d['kernel_ms']['demod']/1e6, d['packets_valid_per_step_rank0']))"; }
echo "product, 3584 captures (two workgroups of 7 + 1)"; run 3584
echo "one duty wave, 3072 captures (two workgroups of 6 + 1)"; WENET_RX_OCT=6 run 3072
export WENET_RX_LIB=tools/variants/small_nd2/libwenet_rx.so
echo "chain wave + sum wave, 3072 captures (two workgroups of 6 + 2)"; WENET_RX_OCT=6 WENET_RX_OCT_ND=2 run 3072
echo "chain wave + sum wave, 3584 captures (one workgroup of 14 + 2)"; WENET_RX_OCT=14 WENET_RX_OCT_ND=2 run 3584
echo "chain wave + sum wave, 3584 captures (two workgroups of 7 + 2: 18 waves, does not fit 16)"; WENET_RX_OCT=7 WENET_RX_OCT_ND=2 run 3584
