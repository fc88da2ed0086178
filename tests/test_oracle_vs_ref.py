"""CPU: the oracle against the REFERENCE ITSELF (oracle/_ref/libwenet_ref.so + CLI binaries, compiled from
the unmodified sources by oracle/Makefile).  Skipped where oracle/_ref has not been built."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal
from wenet_amd import siggen

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def test_phi0_everywhere():
    O, R = ol.oracle(), ol.ref()
    xs = np.concatenate([np.linspace(0, 12, 60001, dtype=np.float32),
                         (np.arange(0, 70000, 7) / 65536.0).astype(np.float32),
                         np.float32([32767, 32768, 1e10, np.inf, np.nan, -1, -0.0, 1e-7, 9.08e-5, 10.0, 5.0, 1.0])])
    for x in xs:
        a, b = O.ora_phi0(float(x)), R.ref_phi0(float(x))
        assert a == b, (x, a, b)


def test_code_tables_match_reference():
    R = ol.ref()
    rows = np.ctypeslib.as_array(R.ref_ldpc_H_rows(), shape=(12, 516)).T - 1
    assert (rows == siggen.h_rows()).all()
    sc = np.ctypeslib.as_array(R.ref_scramble_code(), shape=(1000,))
    bits = np.unpackbits(siggen.scramble_bytes())[:1000]
    assert ((sc < 0).astype(np.uint8) == bits).all()


@pytest.mark.parametrize("name,eb,fmt,est", [("v1", 9, "cu8", (0, 0)), ("v2", 7, "cu8", (0, 0)), ("v2", 15, "cs16", (0, 0)),
                                              ("v2", 12, "cu8", (100000, 330000)), ("4fsk", 10, "cu8", (0, 0)), ("v1", 12, "s16", (0, 0))])
def test_demod_vs_reference_cli(name, eb, fmt, est):
    cfg = siggen.CONFIGS[name]()
    raw, _ = siggen.make_capture(cfg, 4, eb, seed=31 + eb, fmt=fmt)
    extra = ("-b", str(est[0]), "-u", str(est[1])) if est[0] else ()
    for soft in (True, False):
        ref_out, _ = ol.ref_cli_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, soft=soft, extra=extra)
        ora_out, _ = ol.oracle_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, est=est, hard=not soft)
        assert bits_equal(ora_out, ref_out)


def test_frame_state_vs_reference_library():
    O, R = ol.oracle(), ol.ref()
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 2, 8.0, seed=77, ppm=120.0)
    rb = ol.raw_bytes(raw)
    fr = R.fsk_create_hbr(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M, 1200, 400)
    fo = O.ora_fsk_create_hbr(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    ndft = R.ref_fsk_Ndft(fr)
    h1, h2 = np.zeros(ndft, np.float32), np.zeros(ndft, np.float32)
    R.ref_fsk_hann(fr, h1); O.ora_fsk_get_hann(fo, h2)
    assert bits_equal(h1, h2)
    off = 0
    for _ in range(120):
        nin = int(R.fsk_nin(fr))
        assert nin == O.ora_fsk_nin(fo)
        comp = np.zeros(2 * nin, np.float32)
        O.ora_convert_samples(2, rb[off * 2:].ctypes.data, nin, comp)
        a, b = np.zeros(48, np.float32), np.zeros(48, np.float32)
        R.fsk_demod_sd(fr, a.ctypes.data, comp.ctypes.data)
        O.ora_fsk_demod_frame(fo, None, b.ctypes.data, comp.ctypes.data)
        assert bits_equal(a, b)
        e1, e2 = np.zeros(ndft // 2, np.float32), np.zeros(ndft // 2, np.float32)
        R.ref_fsk_fft_est(fr, e1); O.ora_fsk_get_fft_est(fo, e2)
        assert bits_equal(e1, e2)
        p1, p2 = np.zeros(8, np.float32), np.zeros(8, np.float32)
        R.ref_fsk_phi_c(fr, p1); O.ora_fsk_get_phi_c(fo, p2)
        assert bits_equal(p1[:4], p2[:4])
        for k, fn in enumerate(["norm_rx_timing", "ppm", "EbNodB", "snr_est", "stats_rx_timing", "foff"]):
            assert getattr(R, "ref_fsk_" + fn)(fr) == O.ora_fsk_get_scalar(fo, k), fn
        off += nin
    R.fsk_destroy(fr); O.ora_fsk_destroy(fo)


def test_sd_to_llr_and_decoder_random():
    O, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(5)
    for t in range(12):
        sd = (np.where(rng.integers(0, 2, 2580), 1.0, -1.0) * rng.uniform(0.2, 3) + rng.standard_normal(2580) * rng.uniform(0.1, 1.5))
        sd = sd.astype(np.float32).astype(np.float64)
        if t == 0:
            sd[:] = np.abs(sd[0])              # zero variance corner
        a, b = np.zeros(2580, np.float32), np.zeros(2580, np.float32)
        O.ora_sd_to_llr(a, sd, 2580); R.sd_to_llr(b, sd, 2580)
        assert bits_equal(a, b)
        o1, o2 = np.zeros(2580, np.uint8), np.zeros(2580, np.uint8)
        p1, p2 = C.c_int(-3), C.c_int(-3)
        for mi in (10, 50):
            i1 = O.ora_ldpc_decode(a, mi, o1, C.byref(p1)); i2 = R.ref_ldpc_decode(b, mi, o2, C.byref(p2))
            assert i1 == i2 and p1.value == p2.value and (o1 == o2).all()


def test_pipeline_vs_reference_cli():
    for name, eb in (("v1", 8), ("v2", 8), ("v2", 5)):
        cfg = siggen.CONFIGS[name]()
        raw, _ = siggen.make_capture(cfg, 8, eb, seed=900 + eb)
        sd, _ = ol.ref_cli_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
        pk, err = ol.ref_cli_ldpc(sd, cfg.mode, "-v")
        d = ol.oracle_deframe(sd, cfg.mode)
        assert b"".join(bytes(d["bytes"][i][:256]) for i in range(d["n"]) if d["crc_ok"][i]) == pk
        its = [int(l.split("iter:")[1]) for l in err.decode().splitlines() if "iter:" in l]
        assert its == list(d["iter"])


# ---- the low-rate constructor fsk_create (fsk.c:278-398, `fsk_demod -l`) ---------------------------------------
@pytest.mark.parametrize("M,Fs,Rs,fmt,eb,ppm", [(4, 8000, 100, "s16", 12, 0.0), (2, 8000, 100, "s16", 9, 0.0), (4, 9600, 300, "cs16", 15, 300.0),
                                               (2, 48000, 1200, "s16", 12, 0.0)])
def test_lbr_demod_vs_reference_cli(M, Fs, Rs, fmt, eb, ppm):
    cfg = siggen.config_lbr(M, Fs, Rs)
    raw, _ = siggen.make_lbr_capture(cfg, 6, eb, seed=40 + M, fmt=fmt, ppm=ppm)
    for soft in (True, False):
        ref_out, _ = ol.ref_cli_demod(raw, fmt, Fs, Rs, M, soft=soft, extra=("-l",))
        ora_out, _ = ol.oracle_demod(raw, fmt, Fs, Rs, M, hard=not soft, lbr=True)
        assert ref_out.size >= 4 * Rs * (M // 2) and bits_equal(ora_out, ref_out)


def test_lbr_geometry_and_state_vs_reference_library():
    O, R = ol.oracle(), ol.ref()
    cfg = siggen.config_lbr(4, 8000, 100)
    raw, _ = siggen.make_lbr_capture(cfg, 5, 10.0, seed=8, fmt="s16", ppm=-400.0)
    fr = R.fsk_create(cfg.Fs, cfg.Rs, cfg.M, 1200, 400)
    fo = O.ora_fsk_create(cfg.Fs, cfg.Rs, cfg.M)
    for k, fn in enumerate(["Ndft", "N", "Ts", "Nmem", "P", "Nsym", "Nbits", "nstash", "mode", "est_min", "est_max", "est_space"]):
        assert getattr(R, "ref_fsk_" + fn)(fr) == O.ora_fsk_geom(fo, k), fn
    nb = O.ora_fsk_geom(fo, 6)
    off = 0
    for _ in range(4):
        nin = int(R.fsk_nin(fr))
        assert nin == O.ora_fsk_nin(fo)
        comp = np.zeros(2 * nin, np.float32)
        O.ora_convert_samples(0, raw[off:].ctypes.data, nin, comp)
        a, b = np.zeros(nb, np.float32), np.zeros(nb, np.float32)
        R.fsk_demod_sd(fr, a.ctypes.data, comp.ctypes.data)
        O.ora_fsk_demod_frame(fo, None, b.ctypes.data, comp.ctypes.data)
        assert bits_equal(a, b)
        e1, e2 = np.zeros(512, np.float32), np.zeros(512, np.float32)
        R.ref_fsk_fft_est(fr, e1); O.ora_fsk_get_fft_est(fo, e2)
        assert bits_equal(e1, e2)
        for k, fn in enumerate(["norm_rx_timing", "ppm", "EbNodB", "snr_est", "stats_rx_timing", "foff"]):
            assert getattr(R, "ref_fsk_" + fn)(fr) == O.ora_fsk_get_scalar(fo, k), fn
        off += nin
    R.fsk_destroy(fr); O.ora_fsk_destroy(fo)
    assert not O.ora_fsk_create(8000, 300, 2)          # Fs % Rs (fsk.c:292)
    assert not O.ora_fsk_create(8000, 2000, 2)         # Ts % 8  (fsk.c:294)
