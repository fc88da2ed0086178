#!/usr/bin/env python3
"""Development aid: quick parity probe of the HIP path against the oracle on the GPU box."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from wenet_amd import siggen  # noqa: E402
from wenet_amd.fsk import Fsk  # noqa: E402
from wenet_amd.ldpc import Deframer, ldpc_decode_batch, make_ldpc_struct, run_ldpc_decoder, sd_to_llr  # noqa: E402
from wenet_amd.rx import RxBatch  # noqa: E402


def beq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all()


def main():
    O = ol.oracle()
    kat = np.load(os.path.join(ROOT, "tests/golden/ldpc_kat.npz"))
    it, bits, pcc = run_ldpc_decoder(make_ldpc_struct(10), kat["llr"], -7)
    print("KAT: iter", it, "pcc", pcc, "bits ok", (bits == kat["bits"]).all())

    rng = np.random.default_rng(1)
    # random noisy codewords
    n = 64
    llrs = np.zeros((n, 2580), np.float32)
    for i in range(n):
        ib = rng.integers(0, 2, 2064, dtype=np.uint8)
        pb = np.zeros(516, np.uint8); O.ora_ldpc_encode(ib, pb)
        cw = np.concatenate([ib, pb]).astype(np.float64)
        snr = rng.uniform(1.5, 2.6)
        x = (1 - 2 * cw) + rng.standard_normal(2580) / snr
        llrs[i] = (2 * x * snr * snr).astype(np.float32)
    t = time.time(); gb, gi, gp = ldpc_decode_batch(llrs, 10); t_gpu = time.time() - t
    bad = 0
    for i in range(n):
        ob = np.zeros(2580, np.uint8); pc = C.c_int(-1)
        oi = O.ora_ldpc_decode(llrs[i], 10, ob, C.byref(pc))
        if oi != gi[i] or not (ob == gb[i]).all() or pc.value != gp[i]:
            bad += 1
            if bad < 4: print("  LDPC mismatch pkt", i, "iter", oi, gi[i], "pcc", pc.value, gp[i], "bitdiff", int((ob != gb[i]).sum()))
    print("LDPC batch: mismatches", bad, "of", n, "iters hist", np.bincount(gi), "t_gpu %.3f" % t_gpu)

    sd = rng.standard_normal(2580) * 0.3 + np.where(rng.integers(0, 2, 2580), 1.0, -1.0)
    sd = sd.astype(np.float32).astype(np.float64)
    ollr = np.zeros(2580, np.float32); O.ora_sd_to_llr(ollr, sd, 2580)
    gllr = sd_to_llr(sd)
    print("sd_to_llr bit-exact:", beq(ollr, gllr), "maxabs", float(np.abs(ollr - gllr).max()))

    for name, eb, npk in [("v2", 20, 6), ("v1", 8, 6), ("v2", 8, 8), ("4fsk", 12, 3)]:
        cfg = siggen.CONFIGS[name]()
        raw, pl = siggen.make_capture(cfg, npk, eb, seed=500 + eb)
        sd_o, tr_o = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
        f = Fsk(cfg.Fs, cfg.Rs, cfg.Fs // cfg.Rs, cfg.M)
        t = time.time(); sd_g, used, tr_g = f.demod_stream(raw, "cu8", soft=True, want_trace=True); tg = time.time() - t
        ok = beq(sd_o, sd_g)
        print(f"{name} {eb}dB: frames oracle {sd_o.size // f.Nbits} gpu {sd_g.size // f.Nbits} sd bit-exact {ok} t={tg:.3f}s")
        if not ok:
            nb = f.Nbits
            nfr = min(sd_o.size, sd_g.size) // nb
            d = (sd_o[:nfr * nb].view(np.uint32) != sd_g[:nfr * nb].view(np.uint32)).reshape(nfr, nb).any(axis=1)
            ff = int(np.argmax(d)) if d.any() else -1
            print("   first bad frame", ff, "bad frames", int(d.sum()), "max abs diff", float(np.abs(sd_o[:nfr*nb] - sd_g[:nfr*nb]).max()))
            for k in range(max(ff - 1, 0), min(ff + 2, nfr)):
                print("   frame", k, "oracle tr", tr_o[k], "\n            gpu tr", tr_g[k], "\n     sd o", sd_o[k*nb:k*nb+4], "g", sd_g[k*nb:k*nb+4])
        d_o = ol.oracle_deframe(sd_o, cfg.mode, want_llr=True)
        df = Deframer(cfg.mode)
        d_g = df.push(sd_o)
        same = d_o["n"] == d_g["n"] and beq(d_o["bytes"], d_g["bytes"]) and (d_o["iter"] == d_g["iter"]).all() and (d_o["crc_ok"] == d_g["crc_ok"]).all()
        print(f"   deframe+decode: oracle n={d_o['n']} gpu n={d_g['n']} identical {same} iters {d_g['iter']} crc {d_g['crc_ok'].astype(int)} starts_eq {np.array_equal(d_o['start'], d_g['start'])}")
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        rx.enable_trace(); rx.enable_llr_dump()
        rx.process([raw, raw[: raw.size // 2]], "cu8")
        p = rx.packets(0)
        same2 = p["n"] == d_o["n"] and beq(p["bytes"], d_o["bytes"]) and (p["iter"] == d_o["iter"]).all()
        l = rx.llrs(0)
        print(f"   batch chain: n={p['n']} identical-to-oracle {same2} soft bit-exact {beq(rx.soft(0), sd_o)} llr bit-exact {beq(l, d_o['llr'])} ms demod/deframe/decode {rx.last_ms(0):.2f}/{rx.last_ms(1):.3f}/{rx.last_ms(2):.3f}; ch1 frames {rx.frames(1)} pk {rx.npackets(1)}")


if __name__ == "__main__":
    main()
