"""CPU: the synthetic transmitter (wenet_amd/siggen.py) produces frames the Wenet format defines."""
import numpy as np

from wenet_amd import siggen


def test_crc16_known_value():
    assert siggen.crc16_ccitt_false(b"123456789") == 0x29B1


def test_ldpc_parity_satisfies_every_check():
    rng = np.random.default_rng(3)
    ib = rng.integers(0, 2, 2064, dtype=np.uint8)
    pb = siggen.ldpc_parity_bits(ib)
    rows = siggen.h_rows()
    # staircase: check k = sum(data bits of row k) + p[k-1] + p[k] == 0 (mod 2)
    s = ib[rows].sum(axis=1) + pb + np.concatenate([[0], pb[:-1]])
    assert (s % 2 == 0).all()


def test_frame_layout():
    payload = bytes(range(256))
    f1 = siggen.frame_packet(payload, 1)
    f2 = siggen.frame_packet(payload, 2)
    assert len(f1) == len(f2) == 16 + 4 + 256 + 2 + 65
    assert f1[:16] == b"\x55" * 16 and f1[16:20] == bytes([0xAB, 0xCD, 0xEF, 0x01])
    assert f1[20:276] == payload and f2[20:276] != payload
    crc = siggen.crc16_ccitt_false(payload)
    assert f1[276] == (crc & 0xFF) and f1[277] == (crc >> 8)
    code = siggen.scramble_bytes()
    body = np.frombuffer(f2[20:], np.uint8) ^ code[np.arange(323) % 125]
    assert body.tobytes() == f1[20:]


def test_rs232_bits():
    bits = siggen.bytes_to_air_bits(bytes([0xAB]), 1)
    assert list(bits) == [0, 1, 1, 0, 1, 0, 1, 0, 1, 1]          # start, LSB first, stop (drs232_ldpc.c:77-79)
    assert list(siggen.bytes_to_air_bits(bytes([0xAB]), 2)) == [1, 0, 1, 0, 1, 0, 1, 1]


def test_capture_is_deterministic():
    cfg = siggen.config_v2()
    a, _ = siggen.make_capture(cfg, 1, 8.0, seed=5)
    b, _ = siggen.make_capture(cfg, 1, 8.0, seed=5)
    assert (a == b).all() and a.dtype == np.uint8 and a.size == 2 * cfg.Ts * cfg.symbols_per_frame


# ---- transmit-side goldens produced by the reference's own code (tests/golden/make_tx_golden.py) -------------
import ctypes as C
import os

import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tx_golden.npz")


def test_parity_matches_reference_encoder_golden():
    g = np.load(GOLD)
    O = ol.oracle()
    for blk, par in zip(g["blocks"], g["parity"]):
        ib = np.unpackbits(blk)
        assert (siggen.ldpc_parity_bits(ib) == par).all()                  # tx/ldpc_enc.c:33-48
        pb = np.zeros(516, np.uint8)
        O.ora_ldpc_encode(np.ascontiguousarray(ib), pb)                     # mpdecode_core.c:72-91 restated
        assert (pb == par).all()


def test_encoder_table_equals_decoder_table():
    """tx/Hrow2064.txt (encoder, row-major) == src/H2064_516_sparse.h H_rows (decoder, column-major) == the table the
    kernels and the transmitter of this repository use."""
    g = np.load(GOLD)
    assert (g["hrow_txt"].astype(np.int64) - 1 == siggen.h_rows()).all()


def test_parity_matches_reference_encoder_live():
    path = os.path.join(ol.REF_DIR, "ldpc_enc.so")
    if not os.path.exists(path):
        import pytest
        pytest.skip("oracle/_ref/ldpc_enc.so not built")
    enc = C.CDLL(path)
    enc.encode.restype = None
    enc.encode.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(12)
    for _ in range(64):
        ib = rng.integers(0, 2, 2064, dtype=np.uint8)
        pb = np.zeros(516, np.uint8)
        enc.encode(ib.ctypes.data, pb.ctypes.data)
        assert (siggen.ldpc_parity_bits(ib) == pb).all()


def test_noise_model_matches_reference_script_golden():
    g = np.load(GOLD)
    cfg = siggen.config_v2()
    x = siggen.modulate(g["noise_bits"], cfg)
    assert (x == g["noise_in"]).all()
    np.random.seed(int(g["noise_seed"]))
    y = siggen.add_noise(x, cfg, float(g["noise_ebno"]), None, normal=np.random.randn)   # generate_lowsnr.py:70-89
    assert y.dtype == g["noise_out"].dtype and (y == g["noise_out"]).all()


def test_packet_vocabulary():
    """wenet_amd/packets.py (rx/WenetPackets.py:28-47,80-123 restated; reference module not importable here)."""
    from wenet_amd import packets as P
    assert [P.census_class(bytes([t]) + b"\0" * 255) for t in (0, 1, 2, 3, 0x54, 0x55, 0x56, 0x10, 0xFF)] == [0, 1, 2, 3, 4, 5, 6, 7, 7]
    assert P.decode_packet_type(b"\x55abc") == 0x55
    for cs in ("VK5QI", "N0CALL", "A", "W1AW-9"):
        assert P.ssdv_decode_callsign(P.ssdv_encode_callsign(cs)) == cs
    pkt = bytearray(256)
    pkt[0], pkt[1] = 0x55, 0x66
    pkt[2:6] = P.ssdv_encode_callsign("VK5QI")
    pkt[6], pkt[7], pkt[8], pkt[9], pkt[10] = 7, 1, 44, 40, 30
    info = P.ssdv_packet_info(bytes(pkt))
    assert info == {"callsign": "VK5QI", "packet_type": "FEC", "image_id": 7, "packet_id": 300, "width": 640, "height": 480, "error": "None"}
    assert P.ssdv_packet_info(b"\x56" * 256)["error"] != "None" and P.ssdv_packet_info(b"\x55" * 10)["error"] != "None"
