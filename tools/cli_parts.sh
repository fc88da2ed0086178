# development aid: where the drop-in pipe's wall time goes (needs gpurun_out/c.cu8 made by cli_pipe.py --keep)
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wenet_amd import siggen
from wenet_amd.tx import Tx
cfg = siggen.config_v2(); dev = torch.device("cuda", 0); tx = Tx.from_config(cfg)
nsym = 10 * cfg.Rs; spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(5)
pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
out = torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev)
tx.modulate_device([sym.data_ptr()], [nsym], [out.data_ptr()], 8.0, seeds=[1])
torch.cuda.synchronize(); out.cpu().numpy().tofile("/tmp/c.cu8")
PY
B=wenet_amd/bin
t() { python3 -c "import subprocess,sys,time; t0=time.perf_counter(); subprocess.run(sys.argv[2], shell=True, stderr=subprocess.DEVNULL); print('%.3f s  %s' % (time.perf_counter()-t0, sys.argv[1]))" "$1" "$2"; }
t "fsk_demod file->file" "$B/fsk_demod --cu8 -s 2 960000 96000 /tmp/c.cu8 /tmp/c.sd"
t "fsk_demod file->file (2nd)" "$B/fsk_demod --cu8 -s 2 960000 96000 /tmp/c.cu8 /tmp/c.sd"
t "fsk_demod pipe->pipe" "cat /tmp/c.cu8 | $B/fsk_demod --cu8 -s 2 960000 96000 - - > /tmp/c.sd2"
t "wenet_ldpc file->file" "$B/wenet_ldpc /tmp/c.sd /tmp/c.pk"
t "wenet_ldpc pipe" "cat /tmp/c.sd | $B/wenet_ldpc - - > /tmp/c.pk2"
t "full pipe" "cat /tmp/c.cu8 | $B/fsk_demod --cu8 -s 2 960000 96000 - - | $B/wenet_ldpc - - > /tmp/c.pk3"
t "fsk_demod on empty input (start-up cost)" "$B/fsk_demod --cu8 -s 2 960000 96000 /dev/null /tmp/e.sd"
t "wenet_ldpc on empty input (start-up cost)" "$B/wenet_ldpc /dev/null /tmp/e.pk"
t "fsk_demod file->file --stats=100 2>/dev/null" "$B/fsk_demod --cu8 -s --stats=100 2 960000 96000 /tmp/c.cu8 /tmp/c.sd 2>/dev/null"
t "fsk_demod file->file --stats=100 2>file" "$B/fsk_demod --cu8 -s --stats=100 2 960000 96000 /tmp/c.cu8 /tmp/c.sd 2>/tmp/st.txt"
t "fsk_demod file->file --stats=8" "$B/fsk_demod --cu8 -s --stats=8 2 960000 96000 /tmp/c.cu8 /tmp/c.sd 2>/dev/null"
t "reference fsk_demod --stats=100" "oracle/_ref/fsk_demod --cu8 -s --stats=100 2 960000 96000 /tmp/c.cu8 /tmp/c.sd 2>/dev/null"
t "reference fsk_demod" "oracle/_ref/fsk_demod --cu8 -s 2 960000 96000 /tmp/c.cu8 /tmp/c.sd 2>/dev/null"
ls -la /tmp/st.txt
