"""GPU: the HIP path against the oracle on seeded inputs -- API mirrors, streaming, edge cases,
and size-independent properties at BASELINE's full sizes."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal
from wenet_amd import siggen
from wenet_amd.fsk import Fsk
from wenet_amd.ldpc import Deframer, ldpc_decode_batch, make_ldpc_struct, run_ldpc_decoder, sd_to_llr
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


def test_per_frame_api_equals_reference_loop():
    """fsk_create_hbr / fsk_nin / fsk_demod_sd used the way src/fsk_demod.c:270-413 uses them."""
    O = ol.oracle()
    cfg = siggen.config_v1()
    raw, _ = siggen.make_capture(cfg, 1, 9.0, seed=4, fmt="cf32", ppm=200.0)
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    fo = O.ora_fsk_create_hbr(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    off = 0
    for _ in range(40):
        nin = f.nin()
        assert nin == O.ora_fsk_nin(fo)
        x = np.ascontiguousarray(raw[off:off + nin])
        b = np.zeros(48, np.float32)
        O.ora_fsk_demod_frame(fo, None, b.ctypes.data, x.ctypes.data)
        assert bits_equal(f.demod_sd(x), b)
        off += nin
    O.ora_fsk_destroy(fo)
    f.close()


@pytest.mark.parametrize("name,fmt,eb,est", [("v2", "s16", 14, (0, 0)), ("v2", "cf32", 9, (0, 0)), ("v1", "cs16", 8, (0, 0)),
                                              ("v2", "cu8", 12, (100000, 330000)), ("4fsk", "cs16", 11, (0, 0))])
def test_formats_and_estimator_limits(name, fmt, eb, est):
    cfg = siggen.CONFIGS[name]()
    raw, _ = siggen.make_capture(cfg, 3, eb, seed=60 + eb, fmt=fmt)
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    if est[0]:
        f.set_est_limits(*est)
    sd, _, _ = f.demod_stream(raw, fmt)
    ref, _ = ol.oracle_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M, est=est)
    assert bits_equal(sd, ref)
    f.close()


def test_streaming_chunks_equal_one_shot():
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 4, 8.0, seed=12, ppm=-180.0)
    ref, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    rng = np.random.default_rng(2)
    buf = np.zeros(0, np.uint8)
    pos, out = 0, []
    while pos < raw.size or buf.size >= 2 * f.nin():
        n = 2 * int(rng.integers(1, 3000))
        buf = np.concatenate([buf, raw[pos:pos + n]]); pos += n
        sd, used, _ = f.demod_stream(buf, "cu8")
        out.append(sd)
        buf = buf[2 * used:]
        if pos >= raw.size and used == 0:
            break
    assert bits_equal(np.concatenate(out), ref)
    f.close()


def test_streaming_with_changing_input_format(monkeypatch):
    """One stream handle fed the same signal alternately as cu8 and as the exactly equivalent cf32 / cs16-free
    floats: the cu8 launches use the raw-byte sample ring (three captures per CU), the others the float ring, and
    the carried samp_old[] crosses from one to the other.  The result must equal the oracle's on the cu8 stream."""
    monkeypatch.setenv("WENET_RX_FORCE_RAW", "1")
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 4, 9.0, seed=41, ppm=220.0)
    ref, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    as_f32 = ((raw.astype(np.float32) - 127.0) / 128.0)                  # exact (src/fsk_demod.c:283-284)
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    rng = np.random.default_rng(8)
    pos, out, k = 0, [], 0
    nsamp = raw.size // 2
    while True:
        n = int(rng.integers(200, 4000))
        hi = min(pos + n, nsamp)
        if k % 3 == 1:
            sd, used, _ = f.demod_stream(np.ascontiguousarray(as_f32[2 * pos:2 * hi]), "cf32")
        else:
            sd, used, _ = f.demod_stream(np.ascontiguousarray(raw[2 * pos:2 * hi]), "cu8")
        out.append(sd)
        pos += used
        k += 1
        if hi == nsamp and used == 0:
            break
    assert bits_equal(np.concatenate(out), ref)
    f.close()


def test_edge_cases_demod():
    cfg = siggen.config_v2()
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    sd, used, _ = f.demod_stream(np.zeros(0, np.uint8), "cu8")
    assert sd.size == 0 and used == 0
    sd, used, _ = f.demod_stream(np.zeros(2 * (cfg.Ts * 48 - 1), np.uint8), "cu8")      # one sample short of a frame
    assert sd.size == 0 and used == 0
    f.close()
    for raw in (np.full(2 * 480 * 30, 127, np.uint8),          # silence (all-zero COMP): estimator sees nothing
                np.tile(np.array([0, 255], np.uint8), 480 * 30),  # rail-to-rail
                np.full(2 * 480 * 30, 255, np.uint8)):
        f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
        sd, _, _ = f.demod_stream(raw, "cu8")
        ref, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
        assert bits_equal(sd, ref)
        f.close()
    # NaN / Inf input (COMP API): the reference returns early and re-emits the previous frame (fsk.c:876-880)
    x, _ = siggen.make_capture(cfg, 1, 20.0, seed=3, fmt="cf32")
    x = x.copy(); x[480 * 5 + 17] = np.nan; x[480 * 9 + 3] = np.inf
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    sd, _, _ = f.demod_stream(x, "cf32")
    ref, _ = ol.oracle_demod(x, "cf32", cfg.Fs, cfg.Rs, cfg.M)
    assert np.array_equal(sd.view(np.uint32), ref.view(np.uint32))
    f.close()


def test_illegal_parameters_rejected():
    with pytest.raises(RuntimeError):
        Fsk(921600, 96000, 9, 2)            # BASELINE config 2 as written: Fs % Rs != 0 (src/fsk.c:143)
    with pytest.raises(RuntimeError):
        Fsk(960000, 96000, 3, 2)            # (Fs/Rs) % P != 0 (src/fsk.c:145)
    with pytest.raises(RuntimeError):
        Fsk(960000, 96000, 10, 3)


STATS_KERNELS = [pytest.param(None, id="stats-workgroup-per-packet"), pytest.param("0", id="stats-lane-per-packet")]


def _pick_stats_kernel(monkeypatch, small_slots):
    """the LLR statistics run in one of two kernels by batch size (ldpc_kernel.hip: a workgroup per packet up to 32 768 slots, a lane per packet beyond): small tests
    would only ever see the first, so the ones that pin the statistics run with both"""
    if small_slots is not None:
        monkeypatch.setenv("WENET_RX_SMALL_STATS_SLOTS", small_slots)


@pytest.mark.parametrize("small_slots", STATS_KERNELS)
def test_sd_to_llr_corners(small_slots, monkeypatch):
    _pick_stats_kernel(monkeypatch, small_slots)
    O = ol.oracle()
    rng = np.random.default_rng(8)
    cases = [rng.standard_normal(2580), np.full(2580, 0.25), np.zeros(2580),
             np.where(rng.integers(0, 2, 2580), 1e-30, -1e-30), rng.standard_normal(2580) * 1e6,
             np.concatenate([[1e3], np.zeros(2579)]), rng.standard_normal(2580).astype(np.float32).astype(np.float64)]
    for sd in cases:
        sd = np.ascontiguousarray(sd, np.float64)
        a = np.zeros(2580, np.float32)
        O.ora_sd_to_llr(a, sd, 2580)
        b = sd_to_llr(sd)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(a)
        assert (a[m].view(np.uint32) == b[m].view(np.uint32)).all()


def test_ldpc_corners_and_max_iter_50():
    O = ol.oracle()
    rng = np.random.default_rng(9)
    llrs = [np.zeros(2580), np.full(2580, 5.0), np.full(2580, -5.0), np.full(2580, np.nan), np.full(2580, 1e9),
            rng.standard_normal(2580) * 40000, rng.standard_normal(2580) * 1e-4, rng.standard_normal(2580)]
    for mi in (10, 50, 1):
        x = np.ascontiguousarray(np.array(llrs), np.float32)
        bits, iters, pcc = ldpc_decode_batch(x, mi)
        for i in range(x.shape[0]):
            ob = np.zeros(2580, np.uint8); pc = C.c_int(-1)
            it = O.ora_ldpc_decode(x[i], mi, ob, C.byref(pc))
            assert it == iters[i] and pc.value == pcc[i] and (ob == bits[i]).all(), (mi, i)


def test_ldpc_many_noisy_codewords():
    O = ol.oracle()
    rng = np.random.default_rng(10)
    n = 96
    x = np.zeros((n, 2580), np.float32)
    for i in range(n):
        ib = rng.integers(0, 2, 2064, dtype=np.uint8)
        pb = np.zeros(516, np.uint8); O.ora_ldpc_encode(ib, pb)
        snr = rng.uniform(1.5, 2.6)
        y = (1 - 2.0 * np.concatenate([ib, pb])) + rng.standard_normal(2580) / snr
        x[i] = (2 * y * snr * snr).astype(np.float32)
    bits, iters, pcc = ldpc_decode_batch(x, 10)
    for i in range(n):
        ob = np.zeros(2580, np.uint8); pc = C.c_int(-1)
        it = O.ora_ldpc_decode(x[i], 10, ob, C.byref(pc))
        assert it == iters[i] and pc.value == pcc[i] and (ob == bits[i]).all()
    assert len(set(iters.tolist())) > 3          # the sample really spans easy and hard packets


@pytest.mark.parametrize("small_slots", STATS_KERNELS)
def test_deframer_false_uw_and_back_to_back(small_slots, monkeypatch):
    """UW hits inside noise, a detection right after a packet (stale window), EOF inside a packet."""
    _pick_stats_kernel(monkeypatch, small_slots)
    rng = np.random.default_rng(11)
    for mode in (1, 2):
        cfg = siggen.config_v1() if mode == 1 else siggen.config_v2()
        raw, _ = siggen.make_capture(cfg, 5, 7.0, seed=70 + mode, lead_symbols=777)
        sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
        sd = np.concatenate([rng.standard_normal(5000).astype(np.float32), sd, sd[:4000]])
        ref = ol.oracle_deframe(sd, mode)
        d = Deframer(mode)
        got = d.push(sd)
        assert got["n"] == ref["n"] and (got["start"] == ref["start"]).all()
        assert (got["bytes"] == ref["bytes"]).all() and (got["iter"] == ref["iter"]).all()
        assert (got["crc_ok"] == ref["crc_ok"]).all()
        d.close()


@pytest.mark.parametrize("small_slots", STATS_KERNELS)
def test_batch_ragged_and_slot_invariance(small_slots, monkeypatch):
    _pick_stats_kernel(monkeypatch, small_slots)
    cfg = siggen.config_v2()
    caps = [siggen.make_capture(cfg, n, eb, seed=200 + n)[0] for n, eb in ((3, 8.0), (1, 20.0), (5, 6.0), (2, 9.0))]
    caps.append(np.zeros(0, np.uint8))                      # an empty capture in the batch
    caps.append(caps[0][:2 * 480 * 7 + 10])                 # a ragged tail
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(caps, "cu8")
    res = []
    for i, c in enumerate(caps):
        sd, _ = ol.oracle_demod(c, "cu8", cfg.Fs, cfg.Rs, cfg.M) if c.size else (np.zeros(0, np.float32), None)
        assert bits_equal(rx.soft(i), sd)
        ref = ol.oracle_deframe(sd, cfg.mode) if sd.size else dict(n=0)
        assert rx.npackets(i) == ref["n"]
        if ref["n"]:
            assert (rx.packets(i)["bytes"] == ref["bytes"]).all()
        res.append(rx.valid_payloads(i))
    rx.process(caps[::-1], "cu8")                           # same captures, other slots: identical results
    assert [rx.valid_payloads(i) for i in range(len(caps))] == res[::-1]
    rx.close()


@pytest.mark.parametrize("name", ["v2", "v1"])
def test_three_captures_per_workgroup_kernel(name, monkeypatch):
    """The batch kernel that carries three captures per workgroup with one shared NCO-chain wave (demod_tri_impl.h; picked by
    itself from three captures per CU on, forced here): captures of different length and SNR, heavy clock error (slips of one
    capture re-run its two neighbours' barriers), an empty one, a count that is not a multiple of three -- every capture equals
    the oracle, in any slot."""
    monkeypatch.setenv("WENET_RX_TRI", "1")
    cfg = siggen.CONFIGS[name]()
    spec = ((3, 8.0, 0.0), (1, 20.0, 0.0), (5, 6.0, 900.0), (2, 9.0, -1400.0), (4, 7.0, 3000.0), (1, 12.0, 0.0), (6, 8.5, -250.0))
    caps = [siggen.make_capture(cfg, n, eb, seed=500 + i, ppm=ppm)[0] for i, (n, eb, ppm) in enumerate(spec)]
    caps.insert(4, np.zeros(0, np.uint8))                   # an empty capture inside a group
    caps.append(caps[0][:2 * cfg.Ts * 48 * 7 + 10])          # a ragged tail; 9 captures... and one more for a lone last group
    caps.append(caps[2][:2 * cfg.Ts * 48 * 40])
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.enable_trace()
    rx.process(caps, "cu8")
    res, slips = [], 0
    for i, c in enumerate(caps):
        sd, _ = ol.oracle_demod(c, "cu8", cfg.Fs, cfg.Rs, cfg.M) if c.size else (np.zeros(0, np.float32), None)
        assert bits_equal(rx.soft(i), sd), i
        ref = ol.oracle_deframe(sd, cfg.mode) if sd.size else dict(n=0)
        assert rx.npackets(i) == ref["n"]
        if ref["n"]:
            assert (rx.packets(i)["bytes"] == ref["bytes"]).all() and (rx.packets(i)["iter"] == ref["iter"]).all()
        if c.size:
            slips += int((rx.trace(i)[:, 4] != cfg.Ts * 48).sum())
        res.append(rx.valid_payloads(i))
    assert slips > 20                                       # the re-run path was exercised
    rx.process(caps[::-1], "cu8")                           # same captures, other slots and other neighbours: identical results
    assert [rx.valid_payloads(i) for i in range(len(caps))] == res[::-1]
    rx.close()


def test_full_size_properties_10s_8dB():
    """BASELINE config 2 size (10 s, 9.6 M samples, Eb/N0 8 dB): encode -> channel -> decode round trip.
    Every CRC-valid packet is one of the transmitted payloads, in order; most packets are recovered;
    the frame count is what the sample count dictates; re-running is idempotent."""
    import torch
    cfg = siggen.config_v2()
    nsym = 10 * cfg.Rs
    sym, payloads = siggen.air_symbols(cfg, nsym, 2001)
    cap = siggen.make_capture_torch(cfg, sym, 8.0, 7000)
    torch.cuda.synchronize()
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.enqueue_device([int(cap.data_ptr())], [nsym * cfg.Ts], "cu8"); rx.collect()
    out1 = rx.valid_payloads(0)
    frames = rx.frames(0)
    assert abs(frames - nsym // 48) <= 2
    got = [out1[i:i + 256] for i in range(0, len(out1), 256)]
    idx = [payloads.index(g) for g in got]                  # raises if a "valid" packet was never sent
    assert idx == sorted(idx) and len(set(idx)) == len(idx)
    assert len(got) >= 0.9 * (nsym // cfg.symbols_per_frame)
    rx.enqueue_device([int(cap.data_ptr())], [nsym * cfg.Ts], "cu8"); rx.collect()
    assert rx.valid_payloads(0) == out1
    # cross-check the whole 10 s against the oracle
    host = cap.cpu().numpy()
    sd, _ = ol.oracle_demod(host, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    assert bits_equal(rx.soft(0), sd)
    ref = ol.oracle_deframe(sd, cfg.mode)
    assert b"".join(bytes(ref["bytes"][i][:256]) for i in range(ref["n"]) if ref["crc_ok"][i]) == out1
    rx.close()


@pytest.mark.parametrize("name,ppm", [("v2", 600.0), ("v2", -450.0), ("v1", 300.0), ("v1", -900.0)])
def test_heavy_clock_error_many_slips(name, ppm):
    """Large symbol-clock errors: nin != N every few frames, i.e. the pipelined kernel's speculation
    (nin = N) fails constantly and its rollback path carries the whole capture."""
    cfg = siggen.CONFIGS[name]()
    raw, _ = siggen.make_capture(cfg, 3, 12.0, seed=int(abs(ppm)), ppm=ppm)
    ref, tr = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
    assert (tr[:, 4] != cfg.Ts * 48).sum() > 5                      # really many slips
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    sd, _, gtr = f.demod_stream(raw, "cu8", want_trace=True)
    assert bits_equal(sd, ref)
    assert bits_equal(np.ascontiguousarray(gtr[:, 4:7]), np.ascontiguousarray(tr[:, 4:7]))
    f.close()


@pytest.mark.parametrize("name,ppm,shift", [("v1", 3000.0, 0.0), ("v1", 5000.0, 0.0), ("v1", -6000.0, 0.0), ("v2", 4000.0, 0.0),
                                            ("v2", 0.0, 30000.0), ("v2", 0.0, -45000.0), ("v1", 3000.0, 12000.0), ("v2", 0.0, 260000.0)])
def test_baud_rate_error_and_frequency_shift(name, ppm, shift):
    """The reference benchmark's robustness knobs (benchmarking/test_demod.py:71-73, README "Baud Rate Error":
    resampling by 1.003 .. 1.006, and a frequency shift before the demodulator).  Whatever the reference does
    with such a capture -- degrade, lose lock, chase a tone that left the estimator band -- the GPU does the same."""
    import dataclasses
    cfg0 = siggen.CONFIGS[name]()
    cfg = dataclasses.replace(cfg0, f_low=cfg0.f_low + shift)
    raw, _ = siggen.make_capture(cfg, 4, 12.0, seed=int(abs(ppm) + abs(shift) / 100), ppm=ppm)
    ref, tr = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.enable_trace()
    rx.process([raw], "cu8")
    assert bits_equal(rx.soft(0), ref)
    assert bits_equal(np.ascontiguousarray(rx.trace(0)[:, :7]), np.ascontiguousarray(tr[:, :7]))
    d = ol.oracle_deframe(ref, cfg.mode)
    assert rx.valid_payloads(0) == b"".join(bytes(d["bytes"][i][:256]) for i in range(d["n"]) if d["crc_ok"][i])
    rx.close()


@pytest.mark.parametrize("M,Ts,P,fmt,nopipe", [(4, 8, 8, "cu8", False), (4, 10, 10, "cs16", False), (2, 10, 5, "cu8", False),
                                               (2, 10, 2, "cs16", False), (2, 12, 12, "cu8", False), (2, 12, 6, "cf32", False),
                                               (2, 16, 16, "cu8", False), (2, 10, 10, "cu8", True), (4, 8, 4, "cu8", True)])
def test_configuration_matrix(M, Ts, P, fmt, nopipe, monkeypatch):
    """Every kernel variant against the oracle: pipelined kernel with 2 and 4 tones, raw-byte and float sample ring,
    the fast (Ts 8/10, P = Ts) and the generic integrator path (other Ts, fsk_demod's -p option), and the sequential
    kernel (configurations too large for the pipeline, or forced)."""
    import dataclasses
    if nopipe:
        monkeypatch.setenv("WENET_RX_NO_PIPE", "1")
    if fmt == "cu8" and (Ts + P) % 3 != 0:                                   # some of the cu8 cases through the raw-byte ring
        monkeypatch.setenv("WENET_RX_FORCE_RAW", "1")
    Rs = 48000
    cfg = dataclasses.replace(siggen.config_v1(), name="m", M=M, Fs=Rs * Ts, Rs=Rs, f_low=Rs * 1.0, f_space=float(Rs))
    raw, _ = siggen.make_capture(cfg, 2 if M == 2 else 3, 11.0, seed=100 * M + Ts + P, fmt=fmt, ppm=120.0)
    ref, tr = ol.oracle_demod(raw, fmt, cfg.Fs, cfg.Rs, M, P=P, want_trace=True)
    f = Fsk(cfg.Fs, cfg.Rs, P, M)
    sd, _, gtr = f.demod_stream(raw, fmt, want_trace=True)
    assert sd.size > 0 and bits_equal(sd, ref)
    assert bits_equal(np.ascontiguousarray(gtr[:, :7]), np.ascontiguousarray(tr[:, :7]))
    f.close()
    rx = RxBatch(cfg.Fs, cfg.Rs, M, P=P, framing=cfg.mode)
    rx.process([raw, raw[:raw.size // 2]], fmt)
    assert bits_equal(rx.soft(0), ref)
    d = ol.oracle_deframe(ref, cfg.mode)
    assert rx.valid_payloads(0) == b"".join(bytes(d["bytes"][i][:256]) for i in range(d["n"]) if d["crc_ok"][i])
    rx.close()


@pytest.mark.parametrize("split", ["0", "1"])
def test_both_timing_sum_variants(split, monkeypatch):
    """The batch chain picks the lane-split timing sum when there are more captures than CUs and the packed one
    otherwise; both are forced here on a small batch (slips, low SNR and a NaN-free silent capture included)."""
    monkeypatch.setenv("WENET_RX_TSUM_SPLIT", split)
    monkeypatch.setenv("WENET_RX_FORCE_RAW", "1")                             # what a large batch runs: raw ring + split sum
    caps, cfgs = [], []
    for name, eb, ppm, seed in (("v2", 8.0, 0.0, 1), ("v2", 5.0, 300.0, 2), ("v2", 14.0, -2500.0, 3)):
        cfg = siggen.CONFIGS[name]()
        raw, _ = siggen.make_capture(cfg, 3, eb, seed=seed, ppm=ppm)
        caps.append(raw)
    caps.append(np.full(2 * 480 * 20, 127, np.uint8))
    cfg = siggen.config_v2()
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.enable_trace()
    rx.process(caps, "cu8")
    for i, raw in enumerate(caps):
        ref, tr = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
        assert bits_equal(rx.soft(i), ref)
        assert bits_equal(np.ascontiguousarray(rx.trace(i)[:, :7]), np.ascontiguousarray(tr[:, :7]))
    rx.close()


def test_every_capture_length_around_frame_boundaries():
    """0..4 frames +- one sample, one launch each and all together as a ragged batch."""
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 1, 10.0, seed=77)
    N = cfg.Ts * 48
    lens = [0, 1, N - 1, N, N + 1, 2 * N - 1, 2 * N, 2 * N + 1, 3 * N, 4 * N + 5]
    caps = [raw[:2 * n] for n in lens]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(caps, "cu8")
    for i, c in enumerate(caps):
        ref = ol.oracle_demod(c, "cu8", cfg.Fs, cfg.Rs, cfg.M)[0] if c.size else np.zeros(0, np.float32)
        assert bits_equal(rx.soft(i), ref), lens[i]
        f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
        sd, used, _ = f.demod_stream(c, "cu8")
        assert bits_equal(sd, ref) and used <= lens[i]
        f.close()
    rx.close()


def test_frame_cap_resume():
    """cap_frames smaller than the data: the launch stops mid-stream with speculative stages in flight; the
    next call must resume from exactly the committed state."""
    import ctypes as C
    from wenet_amd import lib as _lib
    cfg = siggen.config_v1()
    raw, _ = siggen.make_capture(cfg, 2, 9.0, seed=5, ppm=250.0)
    ref, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    L = _lib.load()
    h = L.wenet_fsk_create_hbr(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M, 1200, 400)
    buf = np.ascontiguousarray(raw)
    out = []
    pos = 0
    caps = [1, 2, 3, 5, 7, 11, 13]
    k = 0
    while True:
        chunk = buf[2 * pos:]
        o = np.zeros(48 * 16, np.float32)
        used = C.c_long(0)
        n = L.wenet_fsk_demod_stream(h, 2, chunk.ctypes.data, chunk.size // 2, 1, o.ctypes.data, caps[k % len(caps)], C.byref(used), None)
        assert n >= 0
        if n == 0:
            break
        out.append(o[:n * 48].copy())
        pos += used.value
        k += 1
    L.wenet_fsk_destroy(h)
    assert bits_equal(np.concatenate(out), ref)


@pytest.mark.parametrize("small_slots", STATS_KERNELS)
def test_degenerate_packets_on_the_float_stream(small_slots, monkeypatch):
    """Round 5: the lane-per-packet statistics kernel forms s / mean through the packet's reciprocal and one fused correction step, and falls back to the division when the
    packet's mean is zero / tiny / huge or a symbol is not finite.  Packets that force every branch -- scaled by 1e-36 and 1e+36 (the mean leaves the safe range), all zero, one
    infinite symbol, one NaN -- ride in one stream beside ordinary ones; LLR-derived results (iterations, bytes, CRC flags) must equal the oracle's for every packet."""
    _pick_stats_kernel(monkeypatch, small_slots)
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 9, 9.0, seed=901, lead_symbols=300)
    sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    ref0 = ol.oracle_deframe(sd, cfg.mode)
    assert ref0["n"] >= 8
    sd = sd.copy()
    spp = 2584
    st = [int(x) for x in ref0["start"]]
    with np.errstate(over="ignore", invalid="ignore"):
        sd[st[1]:st[1] + spp] *= np.float32(1e-36)
        sd[st[2]:st[2] + spp] *= np.float32(1e36)
        sd[st[3]:st[3] + spp] = 0.0
        sd[st[4] + 700] = np.inf
        sd[st[5] + 1300] = np.nan
        sd[st[6] + 5] = -np.inf
    ref = ol.oracle_deframe(sd, cfg.mode)
    d = Deframer(cfg.mode)
    got = d.push(sd)
    d.close()
    assert got["n"] == ref["n"] and (got["start"] == ref["start"]).all()
    assert (got["iter"] == ref["iter"]).all() and (got["crc_ok"] == ref["crc_ok"]).all() and (got["bytes"] == ref["bytes"]).all()
