#!/usr/bin/env python3
"""Development aid: PCIe-inclusive rate -- the batch chain fed from HOST buffers (wenet_rx_process with device=0)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = siggen.config_v2()
dev = torch.device("cuda", 0)
tx = Tx.from_config(cfg)
nsym = 10 * cfg.Rs; nsamp = nsym * cfg.Ts
spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(5)
pay = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), B * nfr, sym.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([sym.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], 8.0, seeds=list(range(B)))
torch.cuda.synchronize()
for pinned in (False, True):
    host = [c.cpu().pin_memory().numpy() if pinned else c.cpu().numpy() for c in caps]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(host, "cu8")
    t0 = time.perf_counter(); rx.process(host, "cu8"); dt = time.perf_counter() - t0
    print(f"host-fed ({'pinned' if pinned else 'pageable'}), {B} captures: {B * nsamp / dt / 1e9:.2f} Gsamples/s ({dt * 1e3:.0f} ms; GPU part {rx.last_ms(3):.0f} ms)")
    rx.close()
