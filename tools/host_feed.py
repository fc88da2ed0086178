#!/usr/bin/env python3
"""Host-fed rate (PCIe included): captures in pinned (or, third argument "pageable", ordinary) HOST memory through wenet_rx_process (device = 0).
usage: host_feed.py [captures] [seconds] [pinned|pageable]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
cfg = siggen.config_v2()
dev = torch.device("cuda", 0)
nsym = int(secs * cfg.Rs); nsamp = nsym * cfg.Ts
tx = Tx.from_config(cfg); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
G = min(B, 64)                                                   # a few distinct captures, reused: the host copies are what matters here
_g = torch.Generator(device=dev); _g.manual_seed(2001)          # (seeded: numbers of different runs describe the same batch)
payloads = torch.randint(0, 256, (G * nfr, 256), dtype=torch.uint8, device=dev, generator=_g)
symbols = torch.empty(G * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(payloads.data_ptr(), G * nfr, symbols.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(G)]
tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(G)], [nsym] * G, [c.data_ptr() for c in caps], 8.0, seeds=[7000 + i for i in range(G)])
torch.cuda.synchronize()
kind = sys.argv[3] if len(sys.argv) > 3 else "pinned"
host = [(caps[i % G].cpu().pin_memory() if kind == "pinned" else caps[i % G].cpu().clone()).numpy() for i in range(B)]
del caps
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
for tag, env in (("time slices", None), ("one upload per capture (WENET_RX_NO_SLICES)", "1")):
    if env: os.environ["WENET_RX_NO_SLICES"] = env
    rx.process(host, "cu8")
    t = time.perf_counter(); rx.process(host, "cu8"); t = time.perf_counter() - t
    npk = sum(int(rx.packets(c)["crc_ok"].sum()) for c in range(0, B, max(1, B // 16)))
    print(f"{tag}: {B} captures x {secs} s from {kind} host memory: {t * 1e3:.1f} ms = {B * nsamp / t / 1e9:.2f} Gsamples/s ({2 * B * nsamp / t / 1e9:.1f} GB/s of cu8), kernel {rx.last_kernel()}, "
          f"demod {rx.last_ms(0):.1f} ms, valid packets (sample) {npk}")
