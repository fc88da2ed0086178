#!/usr/bin/env python3
"""Generate the committed golden fixtures FROM THE REFERENCE ITSELF.

Run in the development container (needs /root/reference, compiled unmodified into oracle/_ref by
`make -C oracle ref`).  Every expected value below comes out of the reference's own code:

  * soft-decision stream        oracle/_ref/fsk_demod (the reference CLI binary), file -> stdout
  * per-frame modem state       reference fsk_create_hbr/fsk_demod_sd through libwenet_ref.so
                                (f_est[], nin, norm_rx_timing, ppm, EbNodB, first 8 fft_est bins)
  * per-packet LLRs/iter/pcc/bits   reference sd_to_llr + run_ldpc_decoder through libwenet_ref.so on the
                                symbols the reference deframer collects
  * packet bytes                oracle/_ref/drs232_ldpc | wenet_ldpc (the reference CLI binaries)
  * stats JSON                  first and last line the reference prints with --stats=100

Inputs are this repository's own synthetic captures (wenet_amd/siggen.py, fixed seeds) stored as the
raw cu8/cs16 bytes, so the fixtures are self-contained data: nothing here is reference source text.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from wenet_amd import siggen  # noqa: E402

CASES = [
    # name, config, fmt, ebno, npackets, seed, ppm
    ("v1_20dB", "v1", "cu8", 20.0, 5, 1120, 0.0),
    ("v1_8dB", "v1", "cu8", 8.0, 5, 1108, 0.0),
    ("v1_6dB", "v1", "cu8", 6.0, 5, 1106, 0.0),
    ("v2_20dB", "v2", "cu8", 20.0, 5, 2220, 0.0),
    ("v2_8dB", "v2", "cu8", 8.0, 5, 2208, 0.0),
    ("v2_6dB", "v2", "cu8", 6.0, 5, 2206, 0.0),
    ("v2_cs16_10dB", "v2", "cs16", 10.0, 5, 2310, 0.0),
    ("v2_ppm150_12dB", "v2", "cu8", 12.0, 6, 2412, 150.0),      # exercises nin != N (timing slips)
    ("4fsk_12dB", "4fsk", "cu8", 12.0, 4, 4412, 0.0),
]


def ref_frame_trace(raw, fmt, cfg):
    """Drive the reference library frame by frame (the fsk_demod main loop) and record its state."""
    R = ol.ref()
    O = ol.oracle()
    P = cfg.Fs // cfg.Rs
    f = R.fsk_create_hbr(cfg.Fs, cfg.Rs, P, cfg.M, 1200, 400)
    rb = ol.raw_bytes(raw)
    bps = ol.BYTES_PER_SAMPLE[fmt]
    nsamp = rb.size // bps
    nbits = R.ref_fsk_Nbits(f)
    ndft = R.ref_fsk_Ndft(f)
    off = 0
    rows, sds = [], []
    fe = np.zeros(4, np.float32)
    fft_est = np.zeros(ndft // 2, np.float32)
    while True:
        nin = int(R.fsk_nin(f))
        if off + nin > nsamp:
            break
        comp = np.zeros(2 * nin, np.float32)
        O.ora_convert_samples(ol.FMT[fmt], rb[off * bps:].ctypes.data, nin, comp)   # exact (validated vs the CLI stream below)
        sd = np.zeros(nbits, np.float32)
        R.fsk_demod_sd(f, sd.ctypes.data, comp.ctypes.data)
        R.ref_fsk_f_est(f, fe)
        fe[cfg.M:] = 0          # struct FSK.f_est[M..3] is uninitialised heap memory in the reference
        R.ref_fsk_fft_est(f, fft_est)
        rows.append(np.concatenate([fe, [R.ref_fsk_nin_field(f), R.ref_fsk_norm_rx_timing(f), R.ref_fsk_ppm(f),
                                         R.ref_fsk_EbNodB(f)], fft_est[:8]]).astype(np.float32))
        sds.append(sd)
        off += nin
    R.fsk_destroy(f)
    return np.array(rows, np.float32), np.concatenate(sds)


def main():
    ol.build_ref()
    ol.build_oracle()
    R = ol.ref()
    for name, cname, fmt, ebno, npk, seed, ppm in CASES:
        cfg = siggen.CONFIGS[cname]()
        raw, payloads = siggen.make_capture(cfg, npk, ebno, seed, fmt=fmt, ppm=ppm)
        sd_cli, _ = ol.ref_cli_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, soft=True)
        hard_cli, _ = ol.ref_cli_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, soft=False)
        _, stats_err = ol.ref_cli_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, soft=True, extra=("--stats=100",))
        stats_lines = [l for l in stats_err.decode().splitlines() if l.startswith("{")]
        trace, sd_lib = ref_frame_trace(raw, fmt, cfg)
        assert sd_lib.size == sd_cli.size and (sd_lib.view(np.uint32) == sd_cli.view(np.uint32)).all(), name
        pk_cli, _ = ol.ref_cli_ldpc(sd_cli, cfg.mode)
        # packet positions: the oracle's deframer (its packet bytes are checked against the reference CLI here)
        d = ol.oracle_deframe(sd_cli, cfg.mode)
        ora_valid = b"".join(bytes(d["bytes"][i][:256]) for i in range(d["n"]) if d["crc_ok"][i])
        assert ora_valid == pk_cli, name
        spp = 323 * (10 if cfg.mode == 1 else 8)
        llrs, iters, pccs, bits = [], [], [], []
        scr = np.ctypeslib.as_array(R.ref_scramble_code(), shape=(1000,)).copy()
        for st in d["start"]:
            sym = sd_cli[st:st + spp].astype(np.float64)
            if cfg.mode == 1:
                sym = sym.reshape(-1, 10)[:, 8:0:-1].reshape(-1)      # out[8b+j] = in[10b+8-j] (drs232_ldpc.c:220-225)
            else:
                sym = sym * scr[np.arange(spp) % 1000]                # wenet_ldpc.c:207
            sym = np.ascontiguousarray(sym[:2580])
            llr = np.zeros(2580, np.float32)
            R.sd_to_llr(llr, sym, 2580)
            out = np.zeros(2580, np.uint8)
            pcc = C.c_int(-1)
            it = R.ref_ldpc_decode(llr, 10, out, C.byref(pcc))
            llrs.append(llr); iters.append(it); pccs.append(pcc.value); bits.append(np.packbits(out))
        n_ok = sum(pk_cli[i * 256:(i + 1) * 256] in payloads for i in range(len(pk_cli) // 256))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            raw=raw, fmt=fmt, config=cname, ebno=np.float32(ebno), seed=np.int32(seed), ppm=np.float32(ppm),
            sd=sd_cli, hard=np.packbits(hard_cli), trace=trace,
            pkt_start=d["start"].astype(np.int64), llr=np.array(llrs, np.float32).reshape(-1, 2580),
            iters=np.array(iters, np.int32), pcc=np.array(pccs, np.int32),
            bits=np.array(bits, np.uint8).reshape(-1, 323), packets=np.frombuffer(pk_cli, np.uint8),
            stats_first=stats_lines[0] if stats_lines else "", stats_last=stats_lines[-1] if stats_lines else "",
            n_sent=np.int32(npk))
        print(f"{name}: frames {trace.shape[0]} nin!=N {(trace[:, 4] != cfg.Ts * 48).sum()} packets found {d['n']} "
              f"valid {len(pk_cli) // 256} payload-correct {n_ok} iters {iters}")


if __name__ == "__main__":
    main()
