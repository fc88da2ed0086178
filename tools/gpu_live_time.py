"""Development aid: latency of wenet_rx_push ticks at N channels x 100 ms (one CPU-generated v2 capture on every channel), pinned / pageable host buffers,
gather kernel on / off (WENET_RX_NO_GATHER).  usage: gpu_live_time.py [channels] [seconds]"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
cfg = siggen.config_v2()
npk = int(secs * cfg.Rs / 2584) - 1
raw, _ = siggen.make_capture(cfg, npk, 8.0, seed=5)
raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
tick = cfg.Fs // 10
nsamp = raw.size // 2
for kind in ("pinned", "pageable"):
    for gather in ((1, 0) if kind == "pinned" else (1,)):
        if gather: os.environ.pop("WENET_RX_NO_GATHER", None)
        else: os.environ["WENET_RX_NO_GATHER"] = "1"
        keep = [torch.from_numpy(raw.copy()) for _ in range(nch)]
        if kind == "pinned": keep = [t.pin_memory() for t in keep]
        host = [t.numpy() for t in keep]
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        rx.push([h[:2 * tick] for h in host], "cu8"); rx.flush()
        lat, kms, pk = [], np.zeros(3), 0
        if os.environ.get('LIVE_NO_GC'): gc.disable()
        for k in range(0, nsamp - tick + 1, tick):
            chunks = [h[2 * k:2 * (k + tick)] for h in host]
            t0 = time.perf_counter()
            pk += rx.push(chunks, "cu8")
            lat.append(time.perf_counter() - t0)
            kms += [rx.last_ms(i) for i in range(3)]
        g = rx.live_gathered()
        rx.flush(); rx.close()
        lat = np.array(lat) * 1e3
        top = np.argsort(lat)[-3:][::-1]
        print("   slowest ticks:", ", ".join(f"#{int(i)} {lat[i]:.2f} ms" for i in top), "| median %.3f ms" % float(np.median(lat)))
        print(f"{nch} channels, {kind} buffers, gather {'on' if gather else 'off'} ({g} chunks gathered in the last tick): tick mean {lat.mean():.3f} ms, best {lat.min():.3f}, worst {lat.max():.3f}; "
              f"kernels per tick: demod {kms[0] / len(lat):.3f} deframe {kms[1] / len(lat):.3f} decode {kms[2] / len(lat):.3f}; packets {pk}; {0.1 / (lat.mean() / 1e3):.1f}x real time")
