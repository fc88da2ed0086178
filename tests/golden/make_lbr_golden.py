#!/usr/bin/env python3
"""Golden fixtures for the low-rate constructor (`fsk_demod -l`, fsk_create, src/fsk.c:278-398), generated FROM THE
REFERENCE ITSELF: every expected array is the output of oracle/_ref/fsk_demod (the reference CLI built from the unmodified
sources by `make -C oracle ref`).  Inputs are this repository's own synthetic captures (wenet_amd/siggen.py, fixed
seeds), stored as raw sample bytes.  Run in the development container; writes tests/golden/lbr_golden.npz."""
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
    # name, M, Fs, Rs, fmt, seconds, Eb/N0, seed, ppm
    ("lbr4_s16", 4, 8000, 100, "s16", 8, 10.0, 7104, 0.0),
    ("lbr2_s16_ppm", 2, 8000, 100, "s16", 8, 9.0, 7102, -350.0),      # timing slips: nin = N -+ Ts/2
    ("lbr4_cs16_300", 4, 9600, 300, "cs16", 6, 14.0, 7304, 0.0),
]

out = {"names": np.array([c[0] for c in CASES])}
for name, M, Fs, Rs, fmt, secs, eb, seed, ppm in CASES:
    cfg = siggen.config_lbr(M, Fs, Rs)
    raw, bits = siggen.make_lbr_capture(cfg, secs, eb, seed, fmt=fmt, ppm=ppm)
    sd, err = ol.ref_cli_demod(raw, fmt, Fs, Rs, M, soft=True, extra=("-l", "--stats=1"))
    hard, _ = ol.ref_cli_demod(raw, fmt, Fs, Rs, M, soft=False, extra=("-l",))
    lines = [l for l in err.decode().splitlines() if l.startswith("{")]
    out[name + "_params"] = np.array([M, Fs, Rs, {"s16": 0, "cs16": 1, "cu8": 2}[fmt]], np.int64)
    out[name + "_raw"] = ol.raw_bytes(raw)
    out[name + "_sd"] = sd
    out[name + "_bits"] = hard
    out[name + "_tx_bits"] = bits
    out[name + "_stats_first"] = np.array(lines[0] if lines else "")
    out[name + "_stats_last"] = np.array(lines[-1] if lines else "")
    nb = Rs * (M // 2)                                     # bits per one-second frame
    # for the record: how many of the reference's hard decisions in seconds 3..5 are right, at the modem's own delay
    # (4-FSK tone keying is 3-sym, see siggen.modulate, so its hard bits come out inverted)
    best = min(int((hard[3 * nb:5 * nb] != (bits[3 * nb + d:5 * nb + d] ^ inv)).sum()) for d in range(-8, 9) for inv in (0, 1))
    print(f"{name}: {sd.size} soft decisions, {len(lines)} stats lines, bit errors in seconds 3..5: {best} of {2 * nb}")
np.savez_compressed(os.path.join(HERE, "lbr_golden.npz"), **out)
print("wrote", os.path.join(HERE, "lbr_golden.npz"), os.path.getsize(os.path.join(HERE, "lbr_golden.npz")), "bytes")
