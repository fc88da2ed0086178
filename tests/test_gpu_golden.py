"""GPU: the HIP path, called through the C ABI, against the committed golden vectors that the reference
itself produced (tests/golden/make_golden.py).  Bit-exact: soft decisions, hard bits, tone estimates,
nin, timing, ppm, LLRs, iteration counts, parity-check counts, decoded bits, packet bytes."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal, load_golden
from wenet_amd import siggen
from wenet_amd.fsk import Fsk
from wenet_amd.ldpc import Deframer, ldpc_decode_batch, make_ldpc_struct, run_ldpc_decoder, sd_to_llr
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["float-ring", "raw-ring"])
def ring_variant(request, monkeypatch):
    """Every golden case runs through both sample-ring variants of the pipelined demodulator: the float ring (what a
    small batch gets) and the raw cu8 ring (what a batch of more than two captures per CU gets; forced here)."""
    if request.param == "raw-ring":
        monkeypatch.setenv("WENET_RX_FORCE_RAW", "1")
    else:
        monkeypatch.delenv("WENET_RX_FORCE_RAW", raising=False)
    return request.param


def test_reference_kat_through_run_ldpc_decoder():
    kat = load_golden("ldpc_kat")
    it, bits, pcc = run_ldpc_decoder(make_ldpc_struct(10), kat["llr"], -7)
    assert it == 8 and pcc == 516 and (bits == kat["bits"]).all()


def test_demod_stream(golden):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    f.enable_stats(1, 1)
    sd, used, tr = f.demod_stream(golden["raw"], str(golden["fmt"]), soft=True, want_trace=True)
    assert bits_equal(sd, golden["sd"])
    g = golden["trace"]
    assert tr.shape[0] == g.shape[0]
    M = cfg.M
    assert bits_equal(np.ascontiguousarray(tr[:, :M]), np.ascontiguousarray(g[:, :M]))          # f_est
    assert bits_equal(np.ascontiguousarray(tr[:, 4:7]), np.ascontiguousarray(g[:, 4:7]))        # nin, norm_rx_timing, ppm
    # EbNodB is finished on the host from the kernel's mean/std (glibc log10f, like the reference)
    mean, std = tr[:, 7].astype(np.float64), tr[:, 8].astype(np.float64)
    import math
    lib = C.CDLL("libm.so.6"); lib.log10f.restype = C.c_float; lib.log10f.argtypes = [C.c_float]
    eb = np.array([np.float32(-6) + np.float32(20) * np.float32(lib.log10f(np.float32((1e-6 + m) / (1e-6 + s)))) for m, s in zip(mean, std)], np.float32)
    assert bits_equal(eb, np.ascontiguousarray(g[:, 7]))
    f.close()


def test_demod_hard(golden):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    f = Fsk(cfg.Fs, cfg.Rs, cfg.Ts, cfg.M)
    bits, _, _ = f.demod_stream(golden["raw"], str(golden["fmt"]), soft=False)
    assert (np.packbits(bits) == golden["hard"]).all()
    f.close()


def test_full_chain(golden):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.enable_llr_dump()
    rx.process([golden["raw"]], str(golden["fmt"]))
    assert bits_equal(rx.soft(0), golden["sd"])
    p = rx.packets(0)
    assert p["n"] == golden["pkt_start"].size and (p["start"] == golden["pkt_start"]).all()
    assert bits_equal(rx.llrs(0), golden["llr"].reshape(-1, 2580))
    assert (p["iter"] == golden["iters"]).all()
    assert (p["bytes"] == golden["bits"][:, :258]).all()
    assert rx.valid_payloads(0) == golden["packets"].tobytes()
    rx.close()


def test_ldpc_api_on_golden_llrs(golden):
    llr = golden["llr"].reshape(-1, 2580)
    if llr.shape[0] == 0:
        pytest.skip("no packets in this fixture")
    bits, iters, pcc = ldpc_decode_batch(llr, 10)
    assert (iters == golden["iters"]).all()
    assert (np.packbits(bits, axis=1) == golden["bits"]).all()
    wrote = pcc >= 0
    assert (pcc[wrote] == golden["pcc"][wrote]).all()


def test_deframer_stream_chunked(golden):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    d = Deframer(cfg.mode)
    sd = golden["sd"]
    out, its = b"", []
    rng = np.random.default_rng(1)
    pos = 0
    while pos < sd.size:                                   # arbitrary chunk boundaries, like a pipe
        n = int(rng.integers(1, 4000))
        r = d.push(sd[pos:pos + n])
        pos += n
        for i in range(r["n"]):
            its.append(int(r["iter"][i]))
            if r["crc_ok"][i]:
                out += bytes(r["bytes"][i][:256])
    assert out == golden["packets"].tobytes()
    assert its == list(golden["iters"])
    d.close()
