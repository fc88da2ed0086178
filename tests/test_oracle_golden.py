"""CPU: the plain-C oracle (oracle/wenet_oracle.c) against the committed golden vectors, which were
produced by the reference itself (tests/golden/make_golden.py), and against the reference's own
embedded known-answer vector (src/H2064_516_sparse.h:27-33)."""
import ctypes as C

import numpy as np

from conftest import bits_equal, load_golden
from wenet_amd import siggen


def test_reference_kat(oracle):
    kat = load_golden("ldpc_kat")
    out = np.zeros(2580, np.uint8)
    pcc = C.c_int(-1)
    it = oracle.ora_ldpc_decode(kat["llr"], 10, out, C.byref(pcc))
    assert it == int(kat["expect_iter"]) == 8
    assert pcc.value == int(kat["expect_pcc"]) == 516
    assert (out == kat["bits"]).all()
    # the KAT really exercises the decoder: raw hard decisions differ from the answer
    assert ((kat["llr"] < 0).astype(np.uint8) != kat["bits"]).sum() == 65


def test_demod_soft_stream_bit_exact(golden, ol):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    sd, tr = ol.oracle_demod(golden["raw"], str(golden["fmt"]), cfg.Fs, cfg.Rs, cfg.M, want_trace=True)
    assert bits_equal(sd, golden["sd"])
    g = golden["trace"]
    assert tr.shape[0] == g.shape[0]
    assert bits_equal(tr[:, :8], np.ascontiguousarray(g[:, :8]))      # f_est[4], nin, norm_rx_timing, ppm, EbNodB


def test_demod_hard_bits(golden, ol):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    bits, _ = ol.oracle_demod(golden["raw"], str(golden["fmt"]), cfg.Fs, cfg.Rs, cfg.M, hard=True)
    assert (np.packbits(bits) == golden["hard"]).all()


def test_deframe_llr_decode(golden, ol):
    cfg = siggen.CONFIGS[str(golden["config"])]()
    d = ol.oracle_deframe(golden["sd"], cfg.mode, want_llr=True)
    assert d["n"] == golden["pkt_start"].size
    assert (d["start"] == golden["pkt_start"]).all()
    assert bits_equal(d["llr"], golden["llr"].reshape(-1, 2580))
    assert (d["iter"] == golden["iters"]).all()
    # decoded codeword bits: fixture holds all 2580 bits packed; the oracle reports the 258 packed bytes
    assert (d["bytes"] == golden["bits"][:, :258]).all()
    valid = b"".join(bytes(d["bytes"][i][:256]) for i in range(d["n"]) if d["crc_ok"][i])
    assert valid == golden["packets"].tobytes()


def test_pcc_matches_reference(golden, oracle):
    llr = golden["llr"].reshape(-1, 2580)
    for i in range(llr.shape[0]):
        out = np.zeros(2580, np.uint8)
        pcc = C.c_int(-1)
        it = oracle.ora_ldpc_decode(np.ascontiguousarray(llr[i]), 10, out, C.byref(pcc))
        assert it == golden["iters"][i] and pcc.value == golden["pcc"][i]
        assert (np.packbits(out) == golden["bits"][i]).all()
