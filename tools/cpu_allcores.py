#!/usr/bin/env python3
"""Development aid: the reference CPU pipe on ALL host cores (xargs -P $(nproc) shape)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wenet_amd import siggen
from wenet_amd.tx import Tx
cfg = siggen.config_v2(); dev = torch.device("cuda", 0); tx = Tx.from_config(cfg)
ncpu = os.cpu_count() or 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else ncpu
nsym = 10 * cfg.Rs; spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(5)
pay = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), B * nfr, sym.data_ptr())
caps = [torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([sym.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], 8.0, seeds=list(range(B)))
torch.cuda.synchronize()
ref = os.path.join(ROOT, "oracle", "_ref")
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    for i, c in enumerate(caps):
        c.cpu().numpy().tofile(os.path.join(td, f"c{i}.cu8"))
    for P in sorted({1, ncpu // 2, ncpu}):
        if P < 1: continue
        cmd = (f"ls {td}/c*.cu8 | xargs -P {P} -I{{}} sh -c '{ref}/fsk_demod --cu8 -s {cfg.M} {cfg.Fs} {cfg.Rs} {{}} - 2>/dev/null | "
               f"{ref}/wenet_ldpc - - 2>/dev/null | wc -c' > /dev/null")
        t0 = time.perf_counter(); subprocess.run(cmd, shell=True, check=True); dt = time.perf_counter() - t0
        print(f"reference pipe, {B} captures, xargs -P {P} ({ncpu} logical CPUs): {dt:.2f} s = {B * nsym * cfg.Ts / dt / 1e6:.0f} Msamples/s")
