#!/usr/bin/env python3
"""Timing of the low-rate path (`fsk_demod -l`, fsk_create): one Horus-style capture through the GPU kernel and through the
reference CPU binary, for DESIGN.md.  The path exists for interface completeness; one audio-rate stream cannot load a GPU."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import siggen
from wenet_amd.fsk import Fsk
M, Fs, Rs, secs = 4, 48000, 100, 120
cfg = siggen.config_lbr(M, Fs, Rs)
raw, _ = siggen.make_lbr_capture(cfg, secs, 10.0, seed=1, fmt="s16")
f = Fsk(Fs, Rs, 0, M, lbr=True)
f.demod_stream(raw[:3 * Fs], "s16")
f.close()
f = Fsk(Fs, Rs, 0, M, lbr=True)
t0 = time.perf_counter(); sd, used, _ = f.demod_stream(raw, "s16"); dt = time.perf_counter() - t0
print(f"GPU: {secs} s of {Fs} Hz real s16, {M}-FSK {Rs} baud: {dt * 1e3:.1f} ms = {secs / dt:.0f}x real time ({sd.size // f.Nbits} frames, {dt * 1e3 / (sd.size // f.Nbits):.2f} ms per one-second frame)")
ref = os.path.join(ROOT, "oracle", "_ref", "fsk_demod")
if os.path.exists(ref):
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "in.s16"); raw.tofile(p)
        t0 = time.perf_counter(); out = subprocess.run([ref, "-l", "-s", str(M), str(Fs), str(Rs), p, "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout; dt = time.perf_counter() - t0
        print(f"reference CPU binary: {dt * 1e3:.1f} ms = {secs / dt:.0f}x real time; identical output: {out == sd.tobytes()}")
