#!/usr/bin/env python3
"""Generate tests/golden/packets_golden.json from the reference's packet-consumer code.

rx/WenetPackets.py cannot be imported whole in this image: its first lines import `crcmod`, which is not installed and
not obtainable (no network).  `crcmod` is used by exactly one function (crc16_ccitt, rx/WenetPackets.py:635-642, reached
only from the Habitat upload helpers).  Nothing is stubbed: this script reads the reference file where it lies, parses it
(ast) and executes ONLY these top-level definitions, none of which reaches crcmod --
    WENET_PACKET_TYPES, decode_packet_type, _ssdv_callsign_alphabet, ssdv_decode_callsign, ssdv_packet_info
-- with the standard-library modules they name (struct, traceback).  Parity status of the consumer row is therefore:
pinned for type dispatch and SSDV header parsing; the Habitat sentence / crc16_ccitt path is outside the receive hot path
and stays uncovered.  The image runs are the state machine of rx/rx_ssdv.py:110-145,224-268 (new image when image_id or
callsign changes), restated in emulate_rx_ssdv() below because that file is a script with side effects (sockets, os.system).

Run in the build container only (needs /root/reference):  python tests/golden/make_packets_golden.py
"""
import ast
import json
import os
import struct
import traceback

import numpy as np

REF = os.environ.get("WENET_REF", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = {"WENET_PACKET_TYPES", "decode_packet_type", "_ssdv_callsign_alphabet", "ssdv_decode_callsign", "ssdv_packet_info"}


def load_reference_functions():
    path = os.path.join(REF, "rx", "WenetPackets.py")
    tree = ast.parse(open(path).read(), path)
    keep = []
    for node in tree.body:
        name = getattr(node, "name", None)
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
        if name in WANTED:
            keep.append(node)
    assert {getattr(n, "name", None) or n.targets[0].id for n in keep} == WANTED
    ns = {"struct": struct, "traceback": traceback}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def ssdv_encode_callsign(callsign):
    value = 0
    for ch in reversed(callsign):
        value = value * 40 + "-0123456789---ABCDEFGHIJKLMNOPQRSTUVWXYZ".index(ch)
    return struct.pack(">I", value)


def main():
    R = load_reference_functions()
    rng = np.random.default_rng(8602)
    packets = []
    for t in (0x00, 0x01, 0x02, 0x03, 0x54, 0x56, 0x10, 0xFF, 0x57, 0x53):
        packets.append(bytes([t]) + rng.integers(0, 256, 255, dtype=np.uint8).tobytes())
    for call, fec, img, pid, w, h in (("VK5QI", 0x66, 3, 0, 20, 15), ("N0CALL", 0x67, 255, 65535, 255, 255), ("A", 0x66, 0, 1, 1, 1),
                                      ("", 0x00, 7, 300, 40, 30), ("ZZ9ZZZ", 0x66, 9, 17, 64, 48), ("4X-1", 0x66, 200, 4096, 2, 3)):
        body = bytes([0x55, fec]) + ssdv_encode_callsign(call) + bytes([img, pid >> 8, pid & 255, w, h])
        packets.append(body + rng.integers(0, 256, 256 - len(body), dtype=np.uint8).tobytes())
    for _ in range(12):                                                     # random headers: whatever the reference makes of them
        packets.append(b"\x55" + rng.integers(0, 256, 255, dtype=np.uint8).tobytes())
    cases = []
    for p in packets:
        info = R["ssdv_packet_info"](p)
        cases.append({"packet": p.hex(), "type": int(R["decode_packet_type"](p)), "ssdv_info": info})
    odd = [{"packet": b.hex(), "ssdv_info": R["ssdv_packet_info"](b)} for b in (b"\x55" * 255, b"\x55" * 257, b"")]
    callsigns = [{"code": c.hex(), "callsign": R["ssdv_decode_callsign"](list(c))} for c in
                 [struct.pack(">I", int(v)) for v in list(rng.integers(0, 2 ** 32, 24, dtype=np.uint64)) + [0, 1, 39, 40, 2 ** 32 - 1]]]
    types = {k: v for k, v in vars(R["WENET_PACKET_TYPES"]).items() if not k.startswith("_")}
    json.dump({"source": "rx/WenetPackets.py: decode_packet_type, ssdv_decode_callsign, ssdv_packet_info executed from the reference file "
                         "(see the generator's header for why the module is not imported whole)",
               "types": types, "cases": cases, "odd_lengths": odd, "callsigns": callsigns},
              open(os.path.join(HERE, "packets_golden.json"), "w"), indent=0)
    print(f"{len(cases)} packets, {len(callsigns)} callsigns")


if __name__ == "__main__":
    main()
