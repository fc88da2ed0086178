#!/usr/bin/env python3
"""Development aid: wall time of ONE 10 s capture through (a) the reference pipe, (b) the drop-in two-process pipe, (c) the fused wenet_rx."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wenet_amd import siggen
from wenet_amd.tx import Tx
cfg = siggen.config_v2(); dev = torch.device("cuda", 0); tx = Tx.from_config(cfg)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
nsym = int(secs * cfg.Rs); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(5)
pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
out = torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev)
tx.modulate_device([sym.data_ptr()], [nsym], [out.data_ptr()], 8.0, seeds=[1])
torch.cuda.synchronize()
ref, gpu = os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "wenet_amd", "bin")
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, "c.cu8"); out.cpu().numpy().tofile(f)
    cmds = {"reference pipe (CPU)": f"cat {f} | {ref}/fsk_demod --cu8 -s 2 {cfg.Fs} {cfg.Rs} - - 2>/dev/null | {ref}/wenet_ldpc - - 2>/dev/null",
            "drop-in pipe (2 GPU processes)": f"cat {f} | {gpu}/fsk_demod --cu8 -s 2 {cfg.Fs} {cfg.Rs} - - 2>/dev/null | {gpu}/wenet_ldpc - - 2>/dev/null",
            "fused wenet_rx (1 GPU process)": f"cat {f} | {gpu}/wenet_rx --cu8 -m 2 2 {cfg.Fs} {cfg.Rs} - - 2>/dev/null",
            "fused wenet_rx from a file": f"{gpu}/wenet_rx --cu8 -m 2 2 {cfg.Fs} {cfg.Rs} {f} - 2>/dev/null",
            "start-up only (empty input)": f"{gpu}/wenet_rx --cu8 -m 2 2 {cfg.Fs} {cfg.Rs} /dev/null - 2>/dev/null"}
    outs = {}
    for name, cmd in cmds.items():
        if "reference" in name and not os.path.exists(os.path.join(ref, "fsk_demod")):
            continue
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); o = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE).stdout; ts.append(time.perf_counter() - t0)
        outs[name] = o
        print(f"{name:34s}: median {sorted(ts)[2]:.3f} s (min {min(ts):.3f}) for {secs:g} s of signal, {len(o)} bytes", flush=True)
    vals = [v for k, v in outs.items() if "start-up" not in k]
    print("outputs identical:", all(v == vals[0] for v in vals))
