#!/usr/bin/env python3
"""Soak parity run of the LIVE channels (wenet_rx_push): rounds of N random channels (config, Eb/N0, clock error, length, format) cut into random ticks -- pageable,
pinned (every source alignment), registered and mixed host buffers, the tick as arrays or as addresses -- against ONE run of the CPU oracle per channel: soft
decisions, packets, iteration counts, CRC flags and packet positions, bit for bit.  usage: soak_live.py [rounds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import oracle_lib as ol
import test_gpu_live as tl
from wenet_amd import siggen
from wenet_amd.fsk import BYTES_PER_SAMPLE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
t0 = time.time()
bad = chans = ticks = packets = 0
for r in range(rounds):
    name = str(rng.choice(["v2", "v2", "v1", "4fsk"]))
    fmt = str(rng.choice(["cu8", "cu8", "cs16", "cf32"]))
    pinned = [False, True, "mixed", "registered"][int(rng.integers(0, 4))]
    cfg = siggen.CONFIGS[name]()
    n = int(rng.integers(1, 40))
    caps = [siggen.make_capture(cfg, int(rng.integers(1, 6)), float(rng.uniform(4, 14)), seed=seed0 * 100000 + r * 100 + ch, fmt=fmt,
                                ppm=float(rng.choice([0.0, rng.uniform(-800, 800)])))[0] for ch in range(n)]
    bps = BYTES_PER_SAMPLE[fmt]
    mean = int(rng.choice([300, 3000, 30000, 200000]))
    cuts = [tl._ragged_cuts(rng, ol.raw_bytes(c).size // bps, mean) for c in caps]
    try:
        out, frames, reported = tl._run_live(cfg, caps, fmt, cuts, pinned=pinned)
        total = tl._check(cfg, caps, fmt, out, frames)
        if total != reported:
            raise AssertionError(f"{reported} packets reported, {total} checked")
        packets += total
    except AssertionError as e:
        bad += 1
        print("MISMATCH", r, name, fmt, pinned, n, mean, e)
    chans += n; ticks += max(len(c) for c in cuts)
print(f"live soak: {rounds} rounds, {chans} channels, {ticks} ticks, {packets} packets, mismatches {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
