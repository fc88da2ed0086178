#!/usr/bin/env python3
"""Development aid: a batch of captures of very different lengths through the batch demodulator, with and without dealing the captures to the
workgroups by length (WENET_RX_NO_SORT).  usage: gpu_ragged.py [captures]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from wenet_amd import siggen
    from wenet_amd.rx import RxBatch
    from wenet_amd.tx import Tx
    B = int(sys.argv[1]); cfg = siggen.CONFIGS["v2"](); dev = torch.device("cuda", 0)
    nsym = 10 * cfg.Rs; nsamp = nsym * cfg.Ts
    tx = Tx.from_config(cfg); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
    pay = torch.randint(0, 256, (64 * nfr, 256), dtype=torch.uint8, device=dev)
    sym = torch.empty(64 * nfr * spp, dtype=torch.uint8, device=dev)
    tx.frame_packets_device(pay.data_ptr(), 64 * nfr, sym.data_ptr())
    base = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(64)]
    tx.modulate_device([sym.data_ptr() + i * nfr * spp for i in range(64)], [nsym] * 64, [c.data_ptr() for c in base], 8.0, seeds=list(range(64)))
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    ns = [int(x) for x in rng.integers(nsamp // 5, nsamp, B)]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    ptrs = [int(base[i % 64].data_ptr()) for i in range(B)]
    for _ in range(3):
        rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
    tot = sum(ns)
    print(f"{rx.last_kernel()}: demod {rx.last_ms(0):.1f} ms, {tot / rx.last_ms(0) / 1e6:.1f} Gsamples/s demod-only; packets {sum(rx.npackets(i) for i in range(0, B, 97))}")
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "3584"
for tag, env in (("by length", {}), ("as given", {"WENET_RX_NO_SORT": "1"}), ("by length, one launch per round (no time slices)", {"WENET_RX_NO_DEV_SLICES": "1"})):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, __file__, B, "child"], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()
    print(tag + ":", out[-1] if out else "failed")
