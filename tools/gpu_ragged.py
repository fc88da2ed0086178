#!/usr/bin/env python3
"""Development aid / evidence (profiles/r05_ragged.txt): a SEEDED batch of captures of very different lengths through the batch demodulator under every schedule -- dealt to
the workgroups by length or as given, cut into time slices inside one launch or one launch per round -- with the demodulator's time, and the proof that the schedule changes
nothing: every schedule's result digest (every packet's bytes, CRC flag, iteration count, position; wenet_rx_result_digest), frame counts and soft decisions must be IDENTICAL,
and a spread of captures is compared with the oracle.  (Round 4's form drew its payloads from torch's unseeded CUDA generator in three child processes: its three packet counts
belonged to three different batches.)  usage: gpu_ragged.py [captures=5000] [shortest seconds=2] [longest seconds=10]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol
from wenet_amd import siggen
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
cfg = siggen.CONFIGS["v2"](); dev = torch.device("cuda", 0)
nsym = int(hi * cfg.Rs); nsamp = nsym * cfg.Ts
tx = Tx.from_config(cfg); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(20250929)
pay = torch.randint(0, 256, (64 * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(64 * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), 64 * nfr, sym.data_ptr())
base = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(64)]
tx.modulate_device([sym.data_ptr() + i * nfr * spp for i in range(64)], [nsym] * 64, [c.data_ptr() for c in base], 8.0, seeds=list(range(64)))
torch.cuda.synchronize()
rng = np.random.default_rng(1)
ns = [int(x) for x in rng.integers(int(lo * cfg.Fs), nsamp, B)]
ptrs = [int(base[i % 64].data_ptr()) for i in range(B)]
tot = sum(ns)
print(f"{B} captures of {lo:g}-{hi:g} s ({tot / 1e9:.2f} G samples), seeds fixed (payloads: torch generator 20250929; noise: 0..63; lengths: numpy default_rng(1))")
ref = None
soft_picks = list(range(0, B, max(1, B // 256)))
for tag, env in (("by length, time slices inside one launch", {}), ("as given", {"WENET_RX_NO_SORT": "1"}), ("by length, one launch per round (no time slices)", {"WENET_RX_NO_DEV_SLICES": "1"})):
    for k in ("WENET_RX_NO_SORT", "WENET_RX_NO_DEV_SLICES"): os.environ.pop(k, None)
    os.environ.update(env)
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    for _ in range(3):
        rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
    dg = rx.result_digest(); frames = [rx.frames(i) for i in range(B)]; soft = [rx.soft(i).copy() for i in soft_picks]
    print(f"{tag}: {rx.last_kernel()}: demod {rx.last_ms(0):.1f} ms, {tot / rx.last_ms(0) / 1e6:.1f} G samples/s demod-only; packets {dg[1]}, CRC-valid {dg[2]}, digest {dg[0]:016x}, decoder repeats {rx.decoder_repeats()}")
    if ref is None:
        ref = (dg, frames, soft)
        picks = list(range(0, B, max(1, B // 32)))[:36]
        t0 = time.time()
        for i in picks:
            raw = base[i % 64][: 2 * ns[i]].cpu().numpy()
            sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
            want = ol.oracle_deframe(sd, cfg.mode)
            got = rx.soft(i); p = rx.packets(i)
            assert got.shape == sd.shape and (got.view(np.uint32) == sd.view(np.uint32)).all(), f"capture {i}: soft decisions differ from the oracle's"
            assert p["n"] == want["n"] and (p["bytes"] == want["bytes"]).all() and (p["iter"] == want["iter"]).all(), f"capture {i}: packets differ from the oracle's"
        print(f"    oracle spot checks: {len(picks)} captures identical (soft decisions bit for bit, packets, iteration counts; {time.time() - t0:.0f} s of CPU)")
    else:
        assert dg == ref[0], f"{tag}: digest {dg} against {ref[0]}"
        assert frames == ref[1], tag
        assert all(a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(soft, ref[2])), tag
        print(f"    identical to the first schedule: digest over all {dg[1]} packets, frame counts of all captures, soft decisions of {len(soft_picks)} captures bit for bit")
    rx.close()
