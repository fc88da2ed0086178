"""CPU: the packet-consumer vocabulary against fixtures produced by the reference's own functions
(tests/golden/packets_golden.json <- rx/WenetPackets.py via tests/golden/make_packets_golden.py) -- the Python twin
(wenet_amd/packets.py) and, through the C ABI (no GPU needed: pure host functions), the library."""
import ctypes as C
import json
import os

from conftest import GOLDEN_DIR
from wenet_amd import lib, packets as P

G = json.load(open(os.path.join(GOLDEN_DIR, "packets_golden.json")))


def test_type_constants_and_dispatch():
    ours = {k: v for k, v in vars(P.WENET_PACKET_TYPES).items() if not k.startswith("_")}
    assert ours == G["types"]
    L = lib.load()
    for c in G["cases"]:
        p = bytes.fromhex(c["packet"])
        assert P.decode_packet_type(p) == c["type"]
        assert L.wenet_packet_type_class(p) == P.census_class(p)


def test_ssdv_header_fields():
    L = lib.load()
    for c in G["cases"] + G["odd_lengths"]:
        p = bytes.fromhex(c["packet"])
        assert P.ssdv_packet_info(p) == c["ssdv_info"]
        if len(p) == 256:
            inf = lib.SsdvInfo()
            rc = L.wenet_ssdv_packet_info(p, C.byref(inf))
            want = c["ssdv_info"]
            if want["error"] != "None":
                assert rc == -1
            else:
                assert rc == 0
                assert (inf.callsign.decode(), "FEC" if inf.fec else "No-FEC", inf.image_id, inf.packet_id, inf.width, inf.height) == \
                       (want["callsign"], want["packet_type"], want["image_id"], want["packet_id"], want["width"], want["height"])


def test_callsign_codes():
    for c in G["callsigns"]:
        assert P.ssdv_decode_callsign(bytes.fromhex(c["code"])) == c["callsign"]
        if "-" not in c["callsign"]:                      # '-' stands for several codes: the encoder picks one of them
            assert P.ssdv_decode_callsign(P.ssdv_encode_callsign(c["callsign"])) == c["callsign"]
