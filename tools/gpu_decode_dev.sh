# development aid (round 4): decode-step time at the bench batch with 2 s captures (a fifth of the packets), kernel stats of the decode kernels
# usage: gpu_decode_dev.sh [seconds] [extra bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=${1:-2}; shift
python bench.py --seconds $S --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('decode ms', d['kernel_ms']['decode'], 'demod', d['kernel_ms']['demod'], 'packets', d['packets_valid_per_step_rank0'], d['packets_found_per_step_rank0'])"
