"""Development aid: reproducibility of the decode step.  One batch (B captures x S seconds, generated on the GPU as bench.py does) is processed N times in one process; every
pass's packets (bytes, iteration counts, CRC flags) are compared with the first pass's, and differing packets are printed.  usage: gpu_repro.py [captures] [seconds] [passes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = siggen.config_v2()
dev = torch.device("cuda:0")
nsamp = int(secs * cfg.Fs); nsym = nsamp // (cfg.Fs // cfg.Rs)
tx = Tx.from_config(cfg)
spp = tx.symbols_per_packet
nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(2001)
payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], [8.0] * B, seeds=[7000 + i for i in range(B)])
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
ptrs = [int(c.data_ptr()) for c in caps]; ns = [nsamp] * B

def snapshot():
    out = []
    for ch in range(B):
        p = rx.packets(ch)
        out.append((p["bytes"].copy(), p["iter"].copy(), p["crc_ok"].copy()))
    return out

ref = None
ndiff_passes = 0
for it in range(passes):
    rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
    s = snapshot()
    if ref is None:
        ref = s
        print("pass 0:", sum(len(x[1]) for x in s), "packets,", int(sum(x[2].sum() for x in s)), "valid")
        continue
    d = 0
    for ch in range(B):
        a, b = ref[ch], s[ch]
        if len(a[1]) != len(b[1]):
            print(f"pass {it} capture {ch}: {len(b[1])} packets, first pass {len(a[1])}"); d += 1; continue
        bad = np.nonzero((a[1] != b[1]) | (a[2] != b[2]) | (a[0] != b[0]).any(axis=1))[0] if len(a[1]) else []
        for k in bad:
            nbytes = int((a[0][k] != b[0][k]).sum())
            print(f"pass {it} capture {ch} packet {k}: iter {a[1][k]} -> {b[1][k]}, crc {a[2][k]} -> {b[2][k]}, {nbytes} bytes differ")
            d += 1
    ndiff_passes += d > 0
print(f"{passes} passes, {ndiff_passes} with differing packets")
