#!/usr/bin/env python3
"""Development aid (round 5, last): one kernel's ISA cut at its s_barrier instructions -- per segment the instructions, VALU, LDS, scratch accesses, s_waitcnt and lane moves
(v_readlane / v_writelane: scalar spills) -- from a device-only compile of one translation unit with extra flags.  Six seconds per candidate, no GPU: "does this form make the
register allocator reload inside the iterations?" is answered before the variant goes to the box (tools/experiments/README.md "Round 5, last").

    python tools/isa_segments.py [--unit ldpc_kernel] [--kernel wenet_decode_kernel] [-- extra hipcc flags, e.g. -DWR_DEC_NO_LIGHT_LAST]

For wenet_decode_kernel the segments are: 0 set-up + a packet's prologue up to the first barrier | 1 prologue | 2 initial messages | 3 CHECK PASS | 4 VARIABLE PASS |
5 stop rules, loop exit, epilogue up to its first barrier | 6 byte staging | 7 pack + store + the next packet's top."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-gpu-flush-denormals-to-zero --offload-device-only -S").split()


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="ldpc_kernel")
    ap.add_argument("--kernel", default="wenet_decode_kernel")
    a = ap.parse_args(argv)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [a.unit + ".hip", "-o", out], cwd=os.path.join(ROOT, "wenet_amd", "csrc"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.exit(r.stdout[-3000:])
        txt = open(out).read()
    m = re.search(r"^(_Z\d+%s\w*):[^\n]*\n(.*?)s_endpgm" % re.escape(a.kernel), txt, re.S | re.M)
    if not m:
        sys.exit("kernel %s not found in %s" % (a.kernel, a.unit))
    name = m.group(1)
    meta = re.search(r"\.name:\s+%s\n(.*?)\n  - " % re.escape(name), txt + "\n  - ", re.S)
    note = re.search(r"\.name:\s+%s\b.*?(?=\n  - |\Z)" % re.escape(name), txt, re.S)
    for src in (txt,):
        for key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            mm = re.findall(r"\.%s:\s+(\d+)" % key, note.group(0) if note else "")
            if mm:
                print("%s %s" % (key, mm[0]), end="   ")
    print()
    keys = ("n", "valu", "lds", "scratch", "waitcnt", "lane")
    seg, cur = [], dict.fromkeys(keys, 0)
    for ln in m.group(2).splitlines():
        ln = ln.strip()
        if not ln or ln.startswith(";") or ln.startswith(".") or ln.endswith(":"):
            continue
        op = ln.split()[0]
        cur["n"] += 1
        cur["valu"] += op.startswith("v_")
        cur["lds"] += op.startswith("ds_")
        cur["scratch"] += op.startswith("scratch_")
        cur["waitcnt"] += op == "s_waitcnt"
        cur["lane"] += op in ("v_readlane_b32", "v_writelane_b32")
        if op == "s_barrier":
            seg.append(cur)
            cur = dict.fromkeys(keys, 0)
    seg.append(cur)
    print("segment  " + "  ".join("%8s" % k for k in keys))
    for i, s in enumerate(seg):
        print("%7d  " % i + "  ".join("%8d" % s[k] for k in keys))


if __name__ == "__main__":
    main()
