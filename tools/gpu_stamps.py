#!/usr/bin/env python3
"""Development aid: cycle stamps of one frame of the streamed sequential demod kernel (WENET_RX_PROFILE=3)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WENET_RX_PROFILE"] = "3"
import torch
from wenet_amd import siggen, lib
from wenet_amd.rx import RxBatch
cfg = siggen.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "4fsk"]()
nsym = int(1.0 * cfg.Rs)
sym, _ = siggen.air_symbols(cfg, nsym, 1)
cap = siggen.make_capture_torch(cfg, sym, 10.0, 3)
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
for _ in range(2):
    rx.enqueue_device([int(cap.data_ptr())], [nsym * cfg.Ts], "cu8"); rx.collect()
L = lib.load(); L.wenet_rx_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
p = np.zeros(26, np.int64); L.wenet_rx_debug_profile(rx._h, 0, p.ctypes.data)
t0 = p[0]
print(f"frames {rx.frames(0)} demod {rx.last_ms(0):.2f} ms = {rx.last_ms(0) * 1e3 / rx.frames(0):.1f} us/frame")
print("chain end   +%d" % (p[1] - t0)); print("tsum end    +%d" % (p[2] - t0)); print("stream end  +%d" % (p[3] - t0))
for b in range(8):
    if p[4 + 2 * b]: print(f"block {b}: start +{p[4 + 2 * b] - t0} end +{p[5 + 2 * b] - t0}")
