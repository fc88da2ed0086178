#!/usr/bin/env python3
"""Development check of the one-wavefront-per-capture batch kernel (demod_oct_impl.h) against the oracle, with diagnostics:
first differing frame / trace column / soft decision per capture.  usage: gpu_oct_dev.py [v2|v1] [caps] [fast]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
name = sys.argv[1] if len(sys.argv) > 1 else "v2"
caps = sys.argv[2] if len(sys.argv) > 2 else "8"
fast = len(sys.argv) > 3 and sys.argv[3] == "fast"
os.environ["WENET_RX_OCT"] = caps
import numpy as np
import torch  # noqa: F401
import oracle_lib as ol
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

cfg = siggen.CONFIGS[name]()
spec = [(3, 8.0, 0.0), (1, 20.0, 0.0), (5, 6.0, 900.0), (2, 9.0, -1400.0), (4, 7.0, 3000.0), (1, 12.0, 0.0), (6, 8.5, -250.0),
        (2, 8.0, 100.0), (3, 10.0, -100.0), (2, 5.0, 0.0), (1, 8.0, 0.0)]
capsl = [siggen.make_capture(cfg, n, eb, seed=900 + i, ppm=ppm)[0] for i, (n, eb, ppm) in enumerate(spec)]
capsl.insert(4, np.zeros(0, np.uint8))
capsl.append(capsl[0][:2 * cfg.Ts * 48 * 7 + 10])
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
rx.enable_trace()
if fast:
    rx.set_fast()
rx.process(capsl, "cu8")
print("kernel:", rx.last_kernel(), "demod ms", rx.last_ms(0))
bad = 0
for i, c in enumerate(capsl):
    if not c.size:
        print(i, "empty: frames", rx.frames(i)); continue
    sd, tr = ol.oracle_demod(c, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
    g, gt = rx.soft(i), rx.trace(i)
    nfr = sd.size // 48
    msg = f"cap {i}: frames oracle {nfr} gpu {rx.frames(i)}"
    n = min(nfr, rx.frames(i))
    same_tr = (gt[:n, :7].view(np.uint32) == tr[:n, :7].view(np.uint32))
    if not same_tr.all():
        fr, col = np.argwhere(~same_tr)[0]
        msg += f" | trace differs first at frame {fr} col {col}: gpu {gt[fr, :7]} ora {tr[fr, :7]}"
    same_sd = g[:n * 48].view(np.uint32) == sd[:n * 48].view(np.uint32)
    if not same_sd.all():
        k = int(np.argwhere(~same_sd)[0][0])
        rel = np.abs(g[:n * 48] - sd[:n * 48]) / np.maximum(np.abs(sd[:n * 48]), 1e-3)
        msg += f" | sd differs first at {k} (frame {k // 48} sym {k % 48}): gpu {g[k]} ora {sd[k]}; n_diff {int((~same_sd).sum())} max rel {rel.max():.3g}"
    ok = same_tr.all() and same_sd.all() and nfr == rx.frames(i)
    bad += 0 if ok else 1
    ref = ol.oracle_deframe(sd, cfg.mode)
    p = rx.packets(i)
    pk_ok = p["n"] == ref["n"] and (p["bytes"] == ref["bytes"]).all() and (p["iter"] == ref["iter"]).all()
    print(msg, "| OK" if ok else "| MISMATCH", "| packets", "same" if pk_ok else f"DIFFER ({p['n']} vs {ref['n']})")
print("captures with mismatches:", bad)
