#!/usr/bin/env python3
"""Development aid: misprediction rate of the nin speculation under symbol-clock error (PROF instantiation)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WENET_RX_PROFILE"] = "1"
import torch
from wenet_amd import siggen, lib
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx
cfg = siggen.config_v2()
dev = torch.device("cuda", 0)
tx = Tx.from_config(cfg)
nsym = 2 * cfg.Rs
spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(1)
pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
L = lib.load()
L.wenet_rx_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for ppm in (0.0, 100.0, 1000.0, 3000.0, 5000.0):
    for eb in (8.0, 12.0):
        out = torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev)
        tx.modulate_device([sym.data_ptr()], [nsym], [out.data_ptr()], eb, ppm=ppm, seeds=[3])
        torch.cuda.synchronize()
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        rx.enable_trace()
        for _ in range(2):
            rx.enqueue_device([out.data_ptr()], [nsym * cfg.Ts], "cu8"); rx.collect()
        p = np.zeros(26, np.int64)
        L.wenet_rx_debug_profile(rx._h, 0, p.ctypes.data)
        tr = rx.trace(0)
        slips = int((tr[:, 4] != cfg.Ts * 48).sum())
        print(f"ppm {ppm:6.0f} Eb/N0 {eb:4.1f}: frames {rx.frames(0)} slips {slips} mispredictions {p[5]} demod {rx.last_ms(0):.2f} ms")
        rx.close()
