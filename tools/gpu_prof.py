#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the demod kernel (WENET_RX_PROFILE instantiation)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WENET_RX_PROFILE"] = os.environ.get("WENET_RX_PROFILE", "1")
import torch
from wenet_amd import siggen, lib
from wenet_amd.rx import RxBatch
name = sys.argv[1] if len(sys.argv) > 1 else "v2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
cfg = siggen.CONFIGS[name]()
nsym = int(secs * cfg.Rs)
sym, _ = siggen.air_symbols(cfg, nsym, 1)
caps = [siggen.make_capture_torch(cfg, sym, 8.0, 10 + i) for i in range(B)]
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
ptrs = [int(c.data_ptr()) for c in caps]; ns = [nsym * cfg.Ts] * B
for _ in range(2):
    rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
L = lib.load()
L.wenet_rx_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
p = np.zeros(26, np.int64)
L.wenet_rx_debug_profile(rx._h, 0, p.ctypes.data)
fr = rx.frames(0)
if os.environ["WENET_RX_PROFILE"] == "1":
    n7 = ["chain busy", "estimator busy", "T busy", "D busy (wave 3)", "iteration total", "mispredictions", "frames"]
    print(f"{name} B={B} frames={fr} demod_ms={rx.last_ms(0):.2f} us/frame={rx.last_ms(0)*1e3/fr:.2f} (pipelined kernel)")
    for n, v in zip(n7, p):
        print(f"  {n:20s} {v/fr:10.1f} per frame")
    sub = ['load/stage', 'downconv+replay', 'barrier1', 'integrate', 'barrier2', 'tprod', 'barrier3', 'tsum', 'timing+decide', 'emit']
    for n, v in zip(sub, p[16:26]):
        print(f"     {n:18s} {v/fr:9.1f}")
    sys.exit(0)
names = ["load", "fft+iir", "peaks", "chain", "downconv", "integrate", "stash+tprod", "tsum", "timing+decide", "emit", "-", "loophead"]
print(f"{name} B={B} frames={fr} demod_ms={rx.last_ms(0):.2f} us/frame={rx.last_ms(0)*1e3/fr:.2f}")
tot = p[:12].sum()
for n, v in zip(names, p[:12]):
    print(f"  {n:14s} {v/fr:10.0f} cyc/frame  {100*v/max(tot,1):5.1f}%")
print(f"  total {tot/fr:.0f} cyc/frame -> counter freq ~ {tot/ (rx.last_ms(0)*1e-3)/1e6:.0f} MHz")
