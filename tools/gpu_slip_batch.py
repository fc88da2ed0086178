#!/usr/bin/env python3
"""Development aid: demod time of a 768-capture batch under symbol-clock error (timing slips re-run three pipeline steps; in the
three-captures-per-workgroup kernel a slip of one capture stalls its two neighbours too) -- both batch kernels."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch
    from wenet_amd import siggen
    from wenet_amd.rx import RxBatch
    from wenet_amd.tx import Tx
    cfg = siggen.config_v2(); dev = torch.device("cuda", 0); tx = Tx.from_config(cfg)
    B, nsym = 768, 2 * cfg.Rs
    spp = tx.symbols_per_packet; nfr = nsym // spp + 1
    g = torch.Generator(device=dev); g.manual_seed(1)
    pay = torch.randint(0, 256, (nfr, 256), dtype=torch.uint8, device=dev, generator=g)
    sym = torch.empty(nfr * spp, dtype=torch.uint8, device=dev)
    tx.frame_packets_device(pay.data_ptr(), nfr, sym.data_ptr())
    for ppm in (0.0, 100.0, 1000.0, 3000.0):
        caps = [torch.empty(2 * nsym * cfg.Ts, dtype=torch.uint8, device=dev) for _ in range(B)]
        tx.modulate_device([sym.data_ptr()] * B, [nsym] * B, [c.data_ptr() for c in caps], 8.0, ppm=ppm, seeds=[3 + i for i in range(B)])
        torch.cuda.synchronize()
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        for _ in range(2):
            rx.enqueue_device([c.data_ptr() for c in caps], [nsym * cfg.Ts] * B, "cu8"); rx.collect()
        print(f"  clock error {ppm:6.0f} ppm: demod {rx.last_ms(0):7.2f} ms, {sum(int(rx.packets(c)['crc_ok'].sum()) for c in range(B))} valid packets")
        rx.close()
else:
    for name, env in (("three captures per workgroup", {}), ("one capture per workgroup", {"WENET_RX_NO_TRI": "1"})):
        print(name)
        subprocess.run([sys.executable, __file__, "run"], env={**os.environ, **env})
