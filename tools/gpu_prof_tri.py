#!/usr/bin/env python3
"""Development aid: busy ticks per wavefront role of the three-captures-per-workgroup demod kernel (WENET_RX_PROFILE=3)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WENET_RX_PROFILE"] = "3"
import torch
from wenet_amd import siggen, lib
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
cfg = siggen.config_v2(); dev = torch.device("cuda", 0); tx = Tx.from_config(cfg)
nsym = 2 * cfg.Rs; spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(1)
pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
caps = [torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([sym.data_ptr()] * B, [nsym] * B, [c.data_ptr() for c in caps], 8.0, seeds=[3 + i for i in range(B)])
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
for _ in range(2):
    rx.enqueue_device([c.data_ptr() for c in caps], [nsym * cfg.Ts] * B, "cu8"); rx.collect()
L = lib.load(); L.wenet_rx_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
print(f"B={B} demod {rx.last_ms(0):.2f} ms, {rx.last_ms(0) * 1e3 / rx.frames(0):.2f} us per frame step")
for ch in (0, 1, 2, 300):
    p = np.zeros(26, np.int64); L.wenet_rx_debug_profile(rx._h, ch, p.ctypes.data)
    fr = max(int(p[6]), 1)
    print(f"  capture {ch}: per frame: chain {p[0] / fr:8.0f}  estimator {p[1] / fr:8.0f}  T {p[2] / fr:8.0f}  D wave 0 {p[3] / fr:8.0f}  D wave 1 {p[7] / fr:8.0f}  iteration {p[4] / fr:8.0f}  (ticks; slips {p[5]}, frames {p[6]})")
    print(f"             T wave, cumulative: sum done {p[8] / fr:8.0f}  nin published {p[9] / fr:8.0f}  decisions done {p[10] / fr:8.0f}")
