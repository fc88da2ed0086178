#!/usr/bin/env python3
"""Development check of the batch kernel on the 4-FSK / Ts 32 geometry (BASELINE config 4) against the oracle.  usage: gpu_oct4_dev.py [caps] [fast]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WENET_RX_OCT"] = sys.argv[1] if len(sys.argv) > 1 else "4"
fast = len(sys.argv) > 2 and sys.argv[2] == "fast"
import numpy as np
import torch  # noqa
import oracle_lib as ol
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
cfg = siggen.config_4fsk()
spec = [(4, 8.0, 0.0), (2, 12.0, 150.0), (3, 6.5, -300.0), (1, 20.0, 0.0), (2, 9.0, 2000.0)]
caps = [siggen.make_capture(cfg, n, eb, seed=40 + i, ppm=ppm)[0] for i, (n, eb, ppm) in enumerate(spec)]
caps.append(np.zeros(0, np.uint8)); caps.append(caps[0][:2 * 1536 * 5 + 7])
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode, max_iter=50)
rx.enable_trace()
if fast: rx.set_fast()
rx.process(caps, "cu8")
print("kernel:", rx.last_kernel(), "demod ms", rx.last_ms(0))
bad = 0
for i, c in enumerate(caps):
    if not c.size:
        print(i, "empty: frames", rx.frames(i)); continue
    sd, tr = ol.oracle_demod(c, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
    g, gt = rx.soft(i), rx.trace(i)
    nfr = sd.size // 96; n = min(nfr, rx.frames(i))
    msg = f"cap {i}: frames oracle {nfr} gpu {rx.frames(i)}"
    st = gt[:n, :7].view(np.uint32) == tr[:n, :7].view(np.uint32)
    if not st.all():
        fr, col = np.argwhere(~st)[0]; msg += f" | trace differs first at frame {fr} col {col}: gpu {gt[fr, :7]} ora {tr[fr, :7]}"
    ss = g[:n * 96].view(np.uint32) == sd[:n * 96].view(np.uint32)
    if not ss.all():
        k = int(np.argwhere(~ss)[0][0]); msg += f" | sd differs first at {k} (frame {k // 96}): gpu {g[k]} ora {sd[k]}; n_diff {int((~ss).sum())}; max abs {np.abs(g[:n*96]-sd[:n*96]).max():.3g}"
    ok = st.all() and ss.all() and nfr == rx.frames(i); bad += 0 if ok else 1
    ref = ol.oracle_deframe(sd, cfg.mode, max_iter=50); p = rx.packets(i)
    print(msg, "| OK" if ok else "| MISMATCH", "| packets", "same" if (p["n"] == ref["n"] and (p["bytes"] == ref["bytes"]).all()) else "DIFFER")
print("captures with mismatches:", bad)
