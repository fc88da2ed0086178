#!/usr/bin/env python3
"""BASELINE config 3 as a measurement: independent captures with an Eb/N0 sweep through the GPU chain.
Prints the table the reference's benchmarking/README.md:63-79 prints (bytes decoded vs Eb/N0) plus PER,
mean iterations and pre-FEC payload BER, and the aggregate throughput of the batch."""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="v1"); ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--lo", type=float, default=4.0); ap.add_argument("--hi", type=float, default=12.0); ap.add_argument("--n", type=int, default=64)
ap.add_argument("--check-cpu", type=int, default=3, help="captures to cross-check against the reference CPU pipe")
ap.add_argument("--bins", type=int, default=0, help="print this many Eb/N0 bins (sums over the captures of a bin) instead of one row per capture")
a = ap.parse_args()
cfg = siggen.CONFIGS[a.config]()
nsym = int(a.seconds * cfg.Rs); nsamp = nsym * cfg.Ts
ebs = [a.lo + (a.hi - a.lo) * c / (a.n - 1) for c in range(a.n)]
from wenet_amd.tx import Tx
dev = torch.device("cuda", 0)
tx = Tx.from_config(cfg)
spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(3000)
pay = torch.randint(0, 256, (a.n * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
sym = torch.empty(a.n * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(pay.data_ptr(), a.n * nfr, sym.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(a.n)]
tx.modulate_device([sym.data_ptr() + c * nfr * spp for c in range(a.n)], [nsym] * a.n, [c.data_ptr() for c in caps], ebs,
                   seeds=[3000 + c for c in range(a.n)])
torch.cuda.synchronize()
pay_h = pay.cpu().numpy().reshape(a.n, nfr, 256)
pls = [[pay_h[c, k].tobytes() for k in range(nfr)] for c in range(a.n)]
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
ptrs = [int(c.data_ptr()) for c in caps]; ns = [nsamp] * a.n
rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
t0 = time.perf_counter(); rx.enqueue_device(ptrs, ns, "cu8"); rx.collect(); dt = time.perf_counter() - t0
sent = nsym // cfg.symbols_per_frame
print(f"# {a.config}: {a.n} captures x {a.seconds:g} s, Eb/N0 {a.lo}..{a.hi} dB, {sent} packets sent per capture")
print(f"# batch: {a.n * nsamp / dt / 1e6:.0f} Msamples/s ({dt * 1e3:.1f} ms; demod {rx.last_ms(0):.1f} ms, decode {rx.last_ms(2):.1f} ms; kernel {rx.last_kernel()})")
sym_h = sym.cpu().numpy().reshape(a.n, nfr * spp)[:, :nsym]
rows = []
spf = cfg.symbols_per_frame
body0 = (16 + 4) * (10 if cfg.mode == 1 else 8)                  # symbols of preamble + unique word ahead of the packet body
for c, eb in enumerate(ebs):
    p = rx.packets(c)
    ok = [bytes(p["bytes"][i][:256]) for i in range(p["n"]) if p["crc_ok"][i]]
    sset = set(pls[c])
    # per packet found (CRC-valid or not), against the frame that was sent at that position: channel bit errors in the
    # collected symbols (before FEC) and payload bit errors in the decoded bytes (after FEC)
    hard = (rx.soft(c) < 0).astype(np.uint8)
    nsymb = 323 * (10 if cfg.mode == 1 else 8)
    nbit = nerr = npre = epre = 0
    for i in range(p["n"]):
        st = int(p["start"][i])
        k = int(round((st - body0) / spf))
        if not (0 <= k < len(pls[c])) or (k * spf + body0 + nsymb) > nsym or st + nsymb > hard.size or cfg.M != 2:
            continue
        epre += int((hard[st:st + nsymb] != sym_h[c][k * spf + body0:k * spf + body0 + nsymb]).sum()); npre += nsymb
        nerr += int(np.unpackbits(np.frombuffer(pls[c][k], np.uint8) ^ p["bytes"][i][:256]).sum()); nbit += 2048
    rows.append(dict(eb=eb, found=p["n"], valid=len(ok), iters=float(p["iter"].sum()) if p["n"] else 0.0, epre=epre, npre=npre, nerr=nerr, nbit=nbit,
                     sent_ok=all(x in sset for x in ok)))
if a.bins > 0:
    print("| Eb/N0 dB (bin) | captures | packets found | CRC-valid | PER | mean iter | BER before FEC | BER after FEC (found packets) | all valid payloads were sent |")
    print("|---|---|---|---|---|---|---|---|---|")
    for b in range(a.bins):
        rr = rows[len(rows) * b // a.bins: len(rows) * (b + 1) // a.bins]
        if not rr: continue
        f = sum(r["found"] for r in rr); v = sum(r["valid"] for r in rr); npre = sum(r["npre"] for r in rr); nbit = sum(r["nbit"] for r in rr)
        print(f"| {rr[0]['eb']:5.2f}..{rr[-1]['eb']:5.2f} | {len(rr)} | {f} | {v} | {1 - v / max(sent * len(rr), 1):.4f} | {sum(r['iters'] for r in rr) / max(f, 1):.2f} | "
              f"{(sum(r['epre'] for r in rr) / npre) if npre else float('nan'):.2e} | {(sum(r['nerr'] for r in rr) / nbit) if nbit else float('nan'):.2e} | {all(r['sent_ok'] for r in rr)} |")
else:
    print("| Eb/N0 dB | packets found | CRC-valid | bytes decoded | PER | mean iter | BER before FEC | BER after FEC (found packets) | all valid payloads were sent |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['eb']:5.2f} | {r['found']} | {r['valid']} | {256 * r['valid']} | {1 - r['valid'] / max(sent, 1):.3f} | {r['iters'] / max(r['found'], 1):.2f} | "
              f"{(r['epre'] / r['npre']) if r['npre'] else float('nan'):.2e} | {(r['nerr'] / r['nbit']) if r['nbit'] else float('nan'):.2e} | {r['sent_ok']} |")
refdir = os.path.join(ROOT, "oracle", "_ref")
if a.check_cpu and os.path.exists(os.path.join(refdir, "fsk_demod")):
    l2 = os.path.join(refdir, "drs232_ldpc" if cfg.mode == 1 else "wenet_ldpc")
    for c in np.linspace(0, a.n - 1, a.check_cpu).astype(int):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "c.cu8"); caps[c].cpu().numpy().tofile(f)
            t1 = time.perf_counter()
            out = subprocess.run(f"{refdir}/fsk_demod --cu8 -s {cfg.M} {cfg.Fs} {cfg.Rs} {f} - 2>/dev/null | {l2} - - 2>/dev/null", shell=True, stdout=subprocess.PIPE).stdout
            print(f"# reference CPU pipe, capture {c} ({ebs[c]:.2f} dB): {len(out)} bytes in {time.perf_counter() - t1:.2f} s; identical to GPU: {out == rx.valid_payloads(int(c))}")
