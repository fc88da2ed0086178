"""Wenet packet vocabulary on the consumer side of the receive path.

Restates the constants and the SSDV header parsing of rx/WenetPackets.py:28-47,60-120 and the image-run state machine of
rx/rx_ssdv.py:224-268 so that the batch API's per-type streams (include/wenet_rx.h `wenet_rx_packet_census`,
`wenet_rx_get_packets_of_class`, `wenet_rx_ssdv_images`) and the tests speak the reference's names.  Pinned against the reference's
own functions through tests/golden/packets_golden.json (made by tests/golden/make_packets_golden.py, which executes
decode_packet_type / ssdv_decode_callsign / ssdv_packet_info from the reference file; the module as a whole needs `crcmod`,
absent here, for its Habitat upload helper only).  The C ABI (`wenet_packet_type_class`, `wenet_ssdv_packet_info`) holds the
product's implementation; this module is the readable twin the tests compare both against.
"""
from __future__ import annotations

import struct


class WENET_PACKET_TYPES:                       # rx/WenetPackets.py:28-35
    TEXT_MESSAGE = 0x00
    GPS_TELEMETRY = 0x01
    ORIENTATION_TELEMETRY = 0x02
    SEC_PAYLOAD_TELEMETRY = 0x03
    IMAGE_TELEMETRY = 0x54
    SSDV = 0x55
    IDLE = 0x56


# order of wenet_rx_packet_census(): counts[k] belongs to CENSUS_CLASSES[k]; the last class is "anything else"
CENSUS_CLASSES = ["TEXT_MESSAGE", "GPS_TELEMETRY", "ORIENTATION_TELEMETRY", "SEC_PAYLOAD_TELEMETRY",
                  "IMAGE_TELEMETRY", "SSDV", "IDLE", "OTHER"]


def decode_packet_type(packet) -> int:          # rx/WenetPackets.py:44-47
    return bytearray(packet)[0]


def census_class(packet) -> int:
    t = decode_packet_type(packet)
    if t <= 3:
        return t
    if 0x54 <= t <= 0x56:
        return t - 0x54 + 4
    return 7


_SSDV_ALPHABET = "-0123456789---ABCDEFGHIJKLMNOPQRSTUVWXYZ"


def ssdv_decode_callsign(code) -> str:          # rx/WenetPackets.py:80-100 (base-40, least significant character first)
    value = struct.unpack(">I", bytes(bytearray(code)))[0]
    out = ""
    while value:
        out += _SSDV_ALPHABET[value % 40]
        value //= 40
    return out


def ssdv_encode_callsign(callsign: str) -> bytes:
    value = 0
    for ch in reversed(callsign):
        value = value * 40 + _SSDV_ALPHABET.index(ch)
    return struct.pack(">I", value)


def ssdv_packet_info(packet) -> dict:           # rx/WenetPackets.py:103-123
    p = bytearray(packet)
    if len(p) != 256:
        return {"error": "ERROR: Invalid Packet Length"}
    if p[0] != WENET_PACKET_TYPES.SSDV:
        return {"error": "ERROR: Not a SSDV Packet."}
    return {"callsign": ssdv_decode_callsign(p[2:6]), "packet_type": "FEC" if p[1] == 0x66 else "No-FEC",
            "image_id": p[6], "packet_id": (p[7] << 8) + p[8], "width": p[9] * 16, "height": p[10] * 16, "error": "None"}


def ssdv_image_runs(packets):
    """rx/rx_ssdv.py:182-268 restated as data: walk the 256-byte packets of a pipe, keep the SSDV ones, start a new image where
    image_id or callsign differs from the previous SSDV packet's.  Returns [(info_of_first_packet, [packets...]), ...]."""
    runs, cur = [], None
    for p in packets:
        if decode_packet_type(p) != WENET_PACKET_TYPES.SSDV:
            continue
        info = ssdv_packet_info(p)
        if info["error"] != "None":
            continue
        if cur is None or info["image_id"] != cur["image_id"] or info["callsign"] != cur["callsign"]:
            runs.append((info, []))
            cur = info
        runs[-1][1].append(bytes(p))
    return runs
