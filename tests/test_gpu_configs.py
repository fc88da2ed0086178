"""GPU: the HIP path against the oracle at the SHAPE of BASELINE.json's configurations 3, 4 and 5
(SURVEY.md 8d makes them concrete).  Capture lengths are cut to what the CPU oracle finishes in seconds;
the batch shape (how many captures share one launch, the Eb/N0 ladder, 4-FSK geometry, max_iter 50) is the
configuration's own.  Everything is compared bit for bit: soft decisions, LLRs, iteration counts, packet bytes."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


def _oracle_chain(raw, cfg, max_iter=10, want_llr=False):
    sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    return sd, ol.oracle_deframe(sd, cfg.mode, max_iter=max_iter, want_llr=want_llr)


def _valid(ref):
    return b"".join(bytes(ref["bytes"][i][:256]) for i in range(ref["n"]) if ref["crc_ok"][i])


def test_config4_4fsk_fs1843200_max_iter_50_end_to_end():
    """BASELINE config 4: 4-FSK, Rs 57 600 sym/s (115.2 kbit/s), Fs 1 843 200 (Ts 32, N 1536, 1024-point estimator),
    v1 framing on the 4-FSK soft stream, LDPC with max_iter = 50 (struct LDPC.max_iter, src/mpdecode_core.h:18-33;
    the CLI's MAX_ITER is 10).  8 dB and a 6.5 dB capture so that some packets really use more than ten iterations."""
    cfg = siggen.config_4fsk()
    assert (cfg.Fs, cfg.Rs, cfg.M, cfg.Ts) == (1843200, 57600, 4, 32)
    caps = [siggen.make_capture(cfg, 24, 8.0, seed=4001)[0], siggen.make_capture(cfg, 12, 6.5, seed=4002)[0],
            siggen.make_capture(cfg, 6, 8.0, seed=4003, ppm=150.0)[0]]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode, max_iter=50)
    rx.enable_llr_dump()
    rx.process(caps, "cu8")
    iters = []
    for i, raw in enumerate(caps):
        sd, ref = _oracle_chain(raw, cfg, max_iter=50, want_llr=True)
        assert bits_equal(rx.soft(i), sd), i
        p = rx.packets(i)
        assert p["n"] == ref["n"] and ref["n"] > 0
        assert (p["start"] == ref["start"]).all()
        assert bits_equal(rx.llrs(i), ref["llr"]), i
        assert (p["iter"] == ref["iter"]).all() and (p["crc_ok"] == ref["crc_ok"]).all()
        assert (p["bytes"] == ref["bytes"]).all()
        iters += ref["iter"].tolist()
    assert max(iters) > 10, "no packet needed more than the CLI's ten iterations: the max_iter=50 path was not exercised"
    # the same batch with the CLI's limit differs exactly where the oracle with max_iter=10 differs
    rx10 = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode, max_iter=10)
    rx10.process(caps, "cu8")
    for i, raw in enumerate(caps):
        _, ref10 = _oracle_chain(raw, cfg, max_iter=10)
        p = rx10.packets(i)
        assert (p["iter"] == ref10["iter"]).all() and (p["bytes"] == ref10["bytes"]).all()
    rx10.close()
    rx.close()


def test_config3_64_captures_ebno_ladder_4_to_12_dB():
    """BASELINE config 3: 64 independent v2 captures, capture c at Eb/N0 = 4 + 8 c / 63 dB, ONE launch.
    Every capture equals the oracle; every CRC-valid payload was transmitted; the packet error rate falls with Eb/N0."""
    cfg = siggen.config_v2()
    npk = 36                                                                   # a little over 1 s per capture (35 packets/s)
    ebno = [4.0 + 8.0 * c / 63.0 for c in range(64)]
    made = [siggen.make_capture(cfg, npk, ebno[c], seed=3000 + c) for c in range(64)]
    caps = [m[0] for m in made]
    assert min(c.size // 2 for c in caps) >= cfg.Fs                            # >= 1 s each
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(caps, "cu8")
    good = np.zeros(64, int)
    for c in range(64):
        sd, ref = _oracle_chain(caps[c], cfg)
        assert bits_equal(rx.soft(c), sd), c
        p = rx.packets(c)
        assert p["n"] == ref["n"], c
        if ref["n"]:
            assert (p["bytes"] == ref["bytes"]).all() and (p["iter"] == ref["iter"]).all() and (p["crc_ok"] == ref["crc_ok"]).all(), c
        out = rx.valid_payloads(c)
        assert out == _valid(ref)
        sent = made[c][1]
        got = [out[i:i + 256] for i in range(0, len(out), 256)]
        idx = [sent.index(g) for g in got]                                     # raises if a "valid" packet was never sent
        assert idx == sorted(idx) and len(set(idx)) == len(idx)
        good[c] = len(got)
    # PER against Eb/N0: nothing decodes at the bottom of the ladder, (nearly) everything at the top, and the
    # octave averages rise monotonically (single captures scatter: "monotone within noise")
    octave = good.reshape(8, 8).mean(axis=1)
    assert octave[0] <= 1 and octave[-1] >= npk - 2
    assert all(octave[k + 1] >= octave[k] - 1.0 for k in range(7)), octave


@pytest.mark.parametrize("nchan", [128, 16])
def test_config5_v2_channels_in_one_launch(nchan):
    """BASELINE config 5: 128 concurrent Wenet-v2 (I2S framing) channels at 96 kbit/s, 8 dB -- all 128 in one launch
    (what one GPU does when it serves the whole set) and the 16-channel share of one GPU of eight.  Every channel
    equals the oracle; lengths are ragged so that channels finish at different frames."""
    cfg = siggen.config_v2()
    rng = np.random.default_rng(5000 + nchan)
    made = [siggen.make_capture(cfg, int(rng.integers(10, 19)), 8.0, seed=5000 + ch) for ch in range(nchan)]
    caps = [m[0] for m in made]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(caps, "cu8")
    npk = 0
    for ch in range(nchan):
        sd, ref = _oracle_chain(caps[ch], cfg)
        assert bits_equal(rx.soft(ch), sd), ch
        p = rx.packets(ch)
        assert p["n"] == ref["n"], ch
        assert (p["bytes"] == ref["bytes"]).all() and (p["iter"] == ref["iter"]).all(), ch
        out = rx.valid_payloads(ch)
        assert out == _valid(ref)
        assert all(out[i:i + 256] in made[ch][1] for i in range(0, len(out), 256))
        npk += len(out) // 256
    assert npk >= 0.8 * sum(len(m[1]) for m in made) - nchan                   # 8 dB: most packets come through (first one may be lost to acquisition)
    rx.close()
