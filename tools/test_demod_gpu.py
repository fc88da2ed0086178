#!/usr/bin/env python3
"""The reference's benchmarking harness on the GPU chain: benchmarking/test_demod.py (argv shape, modes, one result line per file) without csdr.

    python tools/test_demod_gpu.py -m wenet_i2s_demod [-f './generated/wenet_sample_i2s_fs960000*.bin'] [--quick] [--generate DIR]

The reference reads complex-float files (benchmarking/generate_lowsnr.py:100-125), converts them with `csdr convert_f_u8` / `convert_f_s16`
and pipes them through `fsk_demod --cu8|--cs16 -s --stats=100 ... | {drs232,wenet}_ldpc - - | wc -c`, one file after the other; it prints
`file, bytes out, seconds`.  Here all files of the run form ONE batch: the floats are quantised on the GPU (wenet_rx_set_cf32_quantise, the
restatement of the two csdr converters, parity unpinned) and go through the batch chain; the printed lines have the reference's format, the
seconds being the batch's wall time divided over the files.  --generate writes synthetic complex-float files of the same naming scheme with the
repo's own generator (the reference's generator needs an off-air recording)."""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = {   # benchmarking/test_demod.py:20-45: converter, modem parameters, framing, default file mask
    "wenet_rs232_demod": ("cu8", 2, 921416, 115177, 1, "./generated/wenet_sample_fs921416*.bin"),
    "wenet_rs232_demod_c16": ("cs16", 2, 921416, 115177, 1, "./generated/wenet_sample_fs921416*.bin"),
    "wenet_i2s_demod": ("cu8", 2, 960000, 96000, 2, "./generated/wenet_sample_i2s_fs960000*.bin"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--mode", default="wenet_i2s_demod", choices=sorted(MODES))
    ap.add_argument("-f", "--files", default=None, help="glob of complex-float sample files")
    ap.add_argument("-q", "--quick", action="store_true", help="only the last file of the list (as the reference's --quick)")
    ap.add_argument("--generate", default=None, metavar="DIR", help="write synthetic sample files (Eb/N0 5..12 dB) into DIR first")
    ap.add_argument("--packets", type=int, default=100)
    a = ap.parse_args()
    to_fmt, M, Fs, Rs, framing, mask = MODES[a.mode]
    from wenet_amd import siggen
    from wenet_amd.rx import RxBatch
    if a.generate:
        os.makedirs(a.generate, exist_ok=True)
        cfg = siggen.config_v2() if framing == 2 else siggen.config_v1()
        stem = "wenet_sample_i2s_fs960000" if framing == 2 else "wenet_sample_fs921416"
        for k, eb in enumerate(np.arange(5.0, 12.5, 0.5)):
            x, _ = siggen.make_capture(cfg, a.packets, float(eb), seed=4100 + k, fmt="cf32")
            x.tofile(os.path.join(a.generate, f"{stem}_{eb:04.1f}dB.bin"))
        mask = os.path.join(a.generate, stem + "*.bin")
    files = sorted(glob.glob(a.files or mask))
    if not files:
        print("No files found matching supplied path.")
        return
    if a.quick:
        files = files[-1:]
    caps = [np.fromfile(f, dtype=np.complex64) for f in files]
    rx = RxBatch(Fs, Rs, M, framing=framing)
    rx.set_cf32_quantise(to_fmt)
    print(f"Command: <{len(files)} files> | (GPU) convert_f_{'u8' if to_fmt == 'cu8' else 's16'} | fsk_demod --{to_fmt} -s {M} {Fs} {Rs} | "
          f"{'wenet_ldpc' if framing == 2 else 'drs232_ldpc'} | wc -c")
    t0 = time.time()
    rx.process(caps, "cf32")
    dt = time.time() - t0
    for i, f in enumerate(files):
        print("%s, %d, %.3f" % (os.path.basename(f), len(rx.valid_payloads(i)), dt / len(files)))
    rx.close()


if __name__ == "__main__":
    main()
