#!/usr/bin/env python3
"""Phase cycle counts of the one-wavefront-per-capture batch kernel (WENET_RX_PROFILE=4).  usage: gpu_oct_prof.py [captures] [seconds] [caps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
caps = sys.argv[3] if len(sys.argv) > 3 else "7"
cfgname = sys.argv[4] if len(sys.argv) > 4 else "v2"
os.environ["WENET_RX_PROFILE"] = "4"
os.environ["WENET_RX_OCT"] = caps
if len(sys.argv) > 5:
    os.environ["WENET_RX_OCT_ND"] = sys.argv[5]
import numpy as np
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx

cfg = siggen.CONFIGS[cfgname]()
dev = torch.device("cuda", 0)
nsym = int(secs * cfg.Rs); nsamp = nsym * cfg.Ts
tx = Tx.from_config(cfg)
spp = tx.symbols_per_packet
nfr = nsym // spp + 1
_g = torch.Generator(device=dev); _g.manual_seed(2001)          # (seeded: numbers of different runs describe the same batch)
payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=_g)
symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
caps_t = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps_t], 8.0,
                   seeds=[7000 + i for i in range(B)])
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
for _ in range(2):
    rx.enqueue_device([int(c.data_ptr()) for c in caps_t], [nsamp] * B, "cu8"); rx.collect()
print("kernel", rx.last_kernel(), "captures", B, "demod ms", round(rx.last_ms(0), 3), "Gsamples/s", round(B * nsamp / rx.last_ms(0) / 1e6, 2))
L = rx._L
L.wenet_rx_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
buf = (C.c_longlong * 32)()
for ch in range(0, B, int(caps)):
    n = L.wenet_rx_debug_profile(rx._h, ch, buf)
    v = list(buf)
    fr = max(v[6], 1)
    print(f"group@{ch}: frames {v[6]}; cycles per frame -- capture wave 0: phase C end -> products written (mix / integrate) {v[0] / fr:.0f}, -> barrier (FFT ahead, tone search, wait) {v[1] / fr:.0f}, "
          f"phase C (decisions, bookkeeping) {v[2] / fr:.0f} [fine build: orders read {v[3] / fr:.0f}, +loads issued, requests {(v[4] - v[3]) / fr:.0f}, +decisions {(v[5] - v[4]) / fr:.0f}]"
          f" | duty / chain wave: wait requests {v[8] / fr:.0f}, chain (part 1) {v[9] / fr:.0f}, wait products (ND 2: barrier 1) {v[10] / fr:.0f}, chain part 2 {v[12] / fr:.0f}, sums + estimates {v[13] / fr:.0f}, barrier 2 {v[11] / fr:.0f}"
          f" | sum wave (ND 2): to barrier 1 {v[18] / fr:.0f}, sums + estimates {v[21] / fr:.0f}, barrier 2 {v[19] / fr:.0f} | duty total / frame {sum(v[8:14]) / fr:.0f}"
          f" | fine build, wave 0 sections: slot fetch/align {v[24] / fr:.0f}, tones (mix + window sums) {v[25] / fr:.0f}, timing products {v[26] / fr:.0f}, next slot fetch {v[27] / fr:.0f}, FFT {v[28] / fr:.0f}, tone search {v[29] / fr:.0f}")
    for w in range(1, int(caps)):                                  # the other capture waves of the group: the same three stamps
        if ch + w < B and L.wenet_rx_debug_profile(rx._h, ch + w, buf):
            u = list(buf); fw = max(u[6], 1)
            print(f"    capture wave {w}: mix / integrate {u[0] / fw:.0f}, FFT + tone search + wait {u[1] / fw:.0f}, phase C {u[2] / fw:.0f}")
    if ch >= 3 * int(caps):
        break
