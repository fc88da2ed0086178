"""Wenet packet vocabulary on the consumer side of the receive path.

Restates the constants and the SSDV header parsing of rx/WenetPackets.py:28-47,60-120 so that the batch API's
per-type census (include/wenet_rx.h `wenet_rx_packet_census`) and tests speak the reference's names.  The
reference module itself cannot be imported in this image (it needs `crcmod`), so this restatement is
PARITY UNPINNED against it; the census is checked against a host-side count over the decoded packets.
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
