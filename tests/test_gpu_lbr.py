"""GPU: the low-rate constructor fsk_create (src/fsk.c:278-398, `fsk_demod -l`: one-second frames, P = 8, 1024-point
estimator over 800..2500 Hz) and, with it, every frame geometry that does not fit LDS -- the sequential demod kernel with
its frame buffers in global memory.  Bit-exact against the reference CLI's goldens, the oracle and the reference binary."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal
from wenet_amd import siggen
from wenet_amd.fsk import Fsk

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(os.path.dirname(HERE), "wenet_amd", "bin")
FMTS = {0: "s16", 1: "cs16", 2: "cu8"}


def _golden():
    return np.load(os.path.join(HERE, "golden", "lbr_golden.npz"))


@pytest.mark.parametrize("case", ["lbr4_s16", "lbr2_s16_ppm", "lbr4_cs16_300"])
def test_lbr_goldens_from_reference_cli(case):
    g = _golden()
    M, Fs, Rs, fmt = (int(v) for v in g[case + "_params"])
    raw = g[case + "_raw"]
    f = Fsk(Fs, Rs, 0, M, lbr=True)
    assert (f.N, f.Nsym, f.P, f.Ndft, f.Nbits) == (Fs, Rs, 8, 1024, Rs * (M // 2))
    sd, used, _ = f.demod_stream(raw, FMTS[fmt])
    assert bits_equal(sd, g[case + "_sd"])
    f.close()
    f = Fsk(Fs, Rs, 0, M, lbr=True)
    bits, _, _ = f.demod_stream(raw, FMTS[fmt], soft=False)
    assert (bits == g[case + "_bits"]).all()
    f.close()


@pytest.mark.parametrize("M,Fs,Rs,fmt,eb,ppm", [(4, 48000, 100, "s16", 11, 0.0),        # Horus binary: 100 baud 4-FSK from a 48 kHz sound card
                                               (2, 48000, 1200, "s16", 12, 250.0),     # 1200 symbols per frame
                                               (4, 8000, 50, "cu8", 12, -300.0),
                                               (2, 9600, 300, "cs16", 9, 0.0)])
def test_lbr_vs_oracle(M, Fs, Rs, fmt, eb, ppm):
    cfg = siggen.config_lbr(M, Fs, Rs)
    raw, _ = siggen.make_lbr_capture(cfg, 5, eb, seed=70 + M + Rs, fmt=fmt, ppm=ppm)
    ref, tr_ref = ol.oracle_demod(raw, fmt, Fs, Rs, M, lbr=True, want_trace=True)
    f = Fsk(Fs, Rs, 0, M, lbr=True)
    sd, used, tr = f.demod_stream(raw, fmt, want_trace=True)
    assert sd.size == ref.size >= 3 * Rs * (M // 2) and bits_equal(sd, ref)
    assert bits_equal(tr[:, :4], tr_ref[:, :4]) and (tr[:, 4] == tr_ref[:, 4]).all()     # tone estimates, nin
    assert bits_equal(tr[:, 5], tr_ref[:, 5]) and bits_equal(tr[:, 6], tr_ref[:, 6])        # norm_rx_timing, ppm
    f.close()


def test_lbr_per_frame_api_and_streaming():
    """fsk_create / fsk_nin / fsk_demod_sd the way src/fsk_demod.c:270-413 uses them, then the same capture in uneven chunks."""
    O = ol.oracle()
    cfg = siggen.config_lbr(4, 8000, 100)
    raw, _ = siggen.make_lbr_capture(cfg, 9, 10.0, seed=5, fmt="s16", ppm=2500.0)
    f = Fsk(cfg.Fs, cfg.Rs, 0, cfg.M, lbr=True)
    fo = O.ora_fsk_create(cfg.Fs, cfg.Rs, cfg.M)
    off, nins = 0, []
    for _ in range(8):
        nin = f.nin()
        assert nin == O.ora_fsk_nin(fo)
        nins.append(nin)
        comp = np.zeros(2 * nin, np.float32)
        O.ora_convert_samples(0, raw[off:].ctypes.data, nin, comp)
        b = np.zeros(f.Nbits, np.float32)
        O.ora_fsk_demod_frame(fo, None, b.ctypes.data, comp.ctypes.data)
        assert bits_equal(f.demod_sd(comp.view(np.complex64)), b)
        off += nin
    O.ora_fsk_destroy(fo); f.close()
    assert len(set(nins)) > 1                                              # the clock error made the frame length slip
    ref, _ = ol.oracle_demod(raw, "s16", cfg.Fs, cfg.Rs, cfg.M, lbr=True)
    f = Fsk(cfg.Fs, cfg.Rs, 0, cfg.M, lbr=True)
    rng = np.random.default_rng(3)
    buf, pos, out = np.zeros(0, np.int16), 0, []
    while True:
        n = int(rng.integers(1000, 14000))
        buf = np.concatenate([buf, raw[pos:pos + n]]); pos += n
        sd, used, _ = f.demod_stream(buf, "s16")
        out.append(sd); buf = buf[used:]
        if pos >= raw.size and used == 0:
            break
    assert bits_equal(np.concatenate(out), ref)
    f.close()


def test_lbr_illegal_parameters():
    for args in ((8000, 300, 2), (8000, 2000, 2), (8000, 100, 3), (0, 100, 2)):     # Fs % Rs, Ts % 8, M, Fs (fsk.c:286-295)
        with pytest.raises(RuntimeError):
            Fsk(args[0], args[1], 0, args[2], lbr=True)


@pytest.mark.parametrize("flags,M,Fs,Rs,fmt", [(["-s", "--stats=1"], 4, 8000, 100, "s16"), ([], 2, 8000, 100, "s16"),
                                               (["-s", "-t", "-c"], 4, 9600, 300, "cs16")])
def test_lbr_cli_matches_reference_binary(flags, M, Fs, Rs, fmt, tmp_path):
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    cfg = siggen.config_lbr(M, Fs, Rs)
    raw, _ = siggen.make_lbr_capture(cfg, 7, 11.0, seed=90 + M, fmt=fmt)
    path = tmp_path / "in.raw"
    raw.tofile(path)
    outs = []
    for exe in (os.path.join(ol.REF_DIR, "fsk_demod"), os.path.join(BIN, "fsk_demod")):
        p = subprocess.run([exe, "-l"] + flags + [str(M), str(Fs), str(Rs), str(path), str(tmp_path / "o.bin")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        outs.append((p.stderr.decode(), (tmp_path / "o.bin").read_bytes()))
    (ref_err, ref_out), (my_err, my_out) = outs
    assert len(ref_out) > 0 and my_out == ref_out
    strip = lambda t: re.sub(r'"secs": \d+', '"secs": 0', t)
    assert strip(my_err) == strip(ref_err)
    if any(f.startswith("--stats") or f == "-t" for f in flags):
        assert ref_err.count('"EbNodB"') >= 2


@pytest.mark.parametrize("M,Fs,Rs,P", [(2, 960000, 9600, 100), (4, 960000, 8000, 8)])
def test_hbr_geometry_beyond_lds_vs_oracle(M, Fs, Rs, P):
    """fsk_create_hbr with Ts = 100 / 120 (Ndft 4096): the 48-symbol frame no longer fits LDS and runs from global scratch
    too.  (Fs * Ndft stays below 2^32: beyond that the reference's own `est_max*Ndft` overflows int, fsk.c:569.)"""
    cfg = siggen.ModemConfig("wide", 1, M, Fs, Rs, Fs * 0.2, float(Rs))
    rng = np.random.default_rng(17)
    bits = rng.integers(0, 2, 48 * 30 * (M // 2), dtype=np.uint8)
    raw = siggen.to_cu8(siggen.add_noise(siggen.modulate(bits, cfg), cfg, 10.0, rng))
    ref, _ = ol.oracle_demod(raw, "cu8", Fs, Rs, M, P=P)
    f = Fsk(Fs, Rs, P, M)
    sd, _, _ = f.demod_stream(raw, "cu8")
    assert sd.size == ref.size > 20 * 48 and bits_equal(sd, ref)
    f.close()


def test_batch_path_with_geometry_beyond_lds():
    """The many-captures entry (wenet_rx_process) on a geometry whose frame lives in global scratch: every capture gets its
    own scratch block; soft decisions and packets equal the oracle's, captures of different length side by side."""
    from wenet_amd.rx import RxBatch
    Fs, Rs, M = 960000, 9600, 2
    cfg = siggen.ModemConfig("wide", 1, M, Fs, Rs, Fs * 0.2, 2.0 * Rs)
    caps = [siggen.make_capture(cfg, n, eb, seed=300 + n)[0] for n, eb in ((3, 12.0), (2, 12.0), (4, 10.0))]
    rx = RxBatch(Fs, Rs, M, framing=cfg.mode)
    rx.process(caps, "cu8")
    for c, raw in enumerate(caps):
        ref, _ = ol.oracle_demod(raw, "cu8", Fs, Rs, M)
        assert bits_equal(rx.soft(c), ref)
        d = ol.oracle_deframe(ref, cfg.mode)
        p = rx.packets(c)
        assert p["n"] == d["n"] and (p["crc_ok"] == d["crc_ok"]).all() and (p["bytes"] == d["bytes"]).all()
    assert sum(int(rx.packets(c)["crc_ok"].sum()) for c in range(3)) >= 5          # (the first packet of a capture is lost to acquisition)
    rx.close()
