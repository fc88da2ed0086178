"""Development aid (round 5): where a live tick's time goes.  N channels x 100 ms ticks through wenet_rx_push (RxBatch.push_ptrs, as bench.py's live_128 leg) from pinned or
pageable host buffers with WENET_RX_LIVE_TIMING=1: the library prints the host time per phase of the call when the streams end.
usage: gpu_live_phases.py [channels = 128] [seconds = 3] [pinned|pageable|both]"""
import gc, os, sys, time
os.environ["WENET_RX_LIVE_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
kinds = ("pinned", "pageable") if len(sys.argv) <= 3 or sys.argv[3] == "both" else (sys.argv[3],)
cfg = siggen.config_v2()
npk = int(secs * cfg.Rs / 2584) - 1
raws = []
for s in range(4):                                                    # four different captures dealt round the channels
    raw, _ = siggen.make_capture(cfg, npk, 8.0, seed=5 + s)
    raws.append(np.ascontiguousarray(raw).view(np.uint8).reshape(-1))
n = min(r.size for r in raws)
tick = cfg.Fs // 10
nsamp = n // 2
for kind in kinds:
    keep = [torch.from_numpy(raws[i % 4][:n].copy()) for i in range(nch)]
    if kind == "pinned": keep = [t.pin_memory() for t in keep]
    host = [t.numpy() for t in keep]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.push([h[:2 * tick] for h in host], "cu8"); rx.flush()
    rx.push([h[:2 * tick] for h in host], "cu8"); rx.flush()              # (a second one-tick session: what the FIRST tick of a session costs once the process is warm)
    base = np.array([h.ctypes.data for h in host], np.uint64)
    lat, kms, pk = [], np.zeros(3), 0
    gc.collect(); torch.cuda.synchronize(); gc.disable()
    for k in range(0, nsamp - tick + 1, tick):
        t0 = time.perf_counter()
        pk += rx.push_ptrs(base + np.uint64(2 * k), np.full(nch, tick, np.int64), "cu8")
        lat.append(time.perf_counter() - t0)
        kms += [rx.last_ms(i) for i in range(3)]
    gc.enable()
    g = rx.live_gathered()
    dig = rx.result_digest() if hasattr(rx, "result_digest") else 0
    sys.stderr.flush()
    rx.flush(); rx.close()
    lat = np.array(lat) * 1e3
    top = np.argsort(lat)[-4:][::-1]
    print("   slowest ticks:", ", ".join(f"#{int(i)} {lat[i]:.2f} ms" for i in top), flush=True)
    print(f"{nch} channels, {kind}: {len(lat)} ticks, mean {lat.mean():.3f} ms, median {np.median(lat):.3f}, best {lat.min():.3f}, worst {lat.max():.3f}; kernels per tick: demod {kms[0] / len(lat):.3f} "
          f"deframe {kms[1] / len(lat):.3f} decode {kms[2] / len(lat):.3f}; packets {pk}; chunks gathered {g}", flush=True)
