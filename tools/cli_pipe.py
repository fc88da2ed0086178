#!/usr/bin/env python3
"""Development aid: wall time of the drop-in shell pipe (benchmarking/test_demod.py:26-43 shape) vs the reference binaries."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wenet_amd import siggen
from wenet_amd.tx import Tx
cfg = siggen.config_v2()
dev = torch.device("cuda", 0)
tx = Tx.from_config(cfg)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
nsym = int(secs * cfg.Rs); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(5)
pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
out = torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev)
tx.modulate_device([sym.data_ptr()], [nsym], [out.data_ptr()], 8.0, seeds=[1])
torch.cuda.synchronize()
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, "c.cu8"); out.cpu().numpy().tofile(f)
    res = {}
    for name, d in (("reference (CPU)", os.path.join(ROOT, "oracle", "_ref")), ("this build (GPU)", os.path.join(ROOT, "wenet_amd", "bin"))):
        if not os.path.exists(os.path.join(d, "fsk_demod")):
            continue
        for stats in ("", "--stats=100"):
            cmd = f"cat {f} | {d}/fsk_demod --cu8 -s {stats} 2 {cfg.Fs} {cfg.Rs} - - 2>/dev/null | {d}/wenet_ldpc - - 2>/dev/null"
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); o = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE).stdout; best = min(best, time.perf_counter() - t0)
            res[(name, stats)] = o
            print(f"{name:18s} {stats or 'no stats':12s}: {best:6.2f} s for {secs:g} s of signal = {secs / best:6.1f}x real time, {len(o)} bytes")
    vals = list(res.values())
    print("outputs identical:", all(v == vals[0] for v in vals))
