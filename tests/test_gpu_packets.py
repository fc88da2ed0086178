"""GPU: the packet-consumer side of the batch API -- per-type packet streams and SSDV image runs of a capture
equal what rx/rx_ssdv.py's dispatch (restated in wenet_amd/packets.py, pinned by tests/test_packets.py) makes of the packets the
ORACLE pipe decodes from the same capture."""
import numpy as np
import pytest

import oracle_lib as ol
from wenet_amd import packets as P, siggen
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


def _payloads(rng):
    out = []
    def ssdv(call, img, pid):
        b = bytes([0x55, 0x66]) + P.ssdv_encode_callsign(call) + bytes([img, pid >> 8, pid & 255, 20, 15])
        return b + rng.integers(0, 256, 256 - len(b), dtype=np.uint8).tobytes()
    pid = 0
    for img, call, n in ((3, "VK5QI", 5), (4, "VK5QI", 3), (4, "N0CALL", 2), (4, "VK5QI", 4), (5, "VK5QI", 1)):
        for k in range(n):
            out.append(ssdv(call, img, pid)); pid += 1
            if k % 2 == 1:                                      # telemetry and idle packets interleaved, as on air
                t = [0x00, 0x01, 0x02, 0x03, 0x54, 0x56, 0x77][len(out) % 7]
                out.append(bytes([t]) + rng.integers(0, 256, 255, dtype=np.uint8).tobytes())
    return out


@pytest.mark.parametrize("name", ["v2", "v1"])
def test_per_type_streams_and_ssdv_runs(name):
    cfg = siggen.CONFIGS[name]()
    rng = np.random.default_rng(86)
    caps, sent = [], []
    for eb, seed in ((12.0, 1), (7.6, 2), (5.0, 3)):            # clean, marginal (some packets lost -> runs cut differently), dead
        pl = _payloads(rng)
        raw, _ = siggen.make_capture(cfg, len(pl), eb, seed=860 + seed, payloads=pl)
        caps.append(raw); sent.append(pl)
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.process(caps, "cu8")
    for ch, raw in enumerate(caps):
        sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
        ref = ol.oracle_deframe(sd, cfg.mode)
        pipe = [bytes(ref["bytes"][i][:256]) for i in range(ref["n"]) if ref["crc_ok"][i]]        # what the reference pipe hands to rx_ssdv.py
        assert all(p in sent[ch] for p in pipe)
        for cls in range(8):
            assert rx.packets_of_class(ch, cls) == [p for p in pipe if P.census_class(p) == cls], (ch, cls)
        assert rx.census(ch) == [sum(1 for p in pipe if P.census_class(p) == cls) for cls in range(8)]
        runs = P.ssdv_image_runs(pipe)
        got = rx.ssdv_images(ch)
        assert len(got) == len(runs)
        stream = rx.packets_of_class(ch, 5)
        for g, (info, pk) in zip(got, runs):
            assert (g["callsign"], g["image_id"], g["packet_id"], g["width"], g["height"], g["fec"]) == \
                   (info["callsign"], info["image_id"], info["packet_id"], info["width"], info["height"], info["packet_type"] == "FEC")
            assert g["npackets"] == len(pk) and stream[g["first_index"]:g["first_index"] + g["npackets"]] == pk
    assert len(rx.ssdv_images(0)) >= 4                          # the clean capture really exercises several image changes
    rx.close()
