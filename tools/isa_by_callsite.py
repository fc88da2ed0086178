#!/usr/bin/env python3
"""Static instruction counts of a kernel by OUTERMOST call site in the kernel body (development aid).  Usage:
    hipcc ... --cuda-device-only -S -gline-tables-only -o k.s demod_oct.hip ; python tools/isa_by_callsite.py k.s ILi2ELi10ELi256E [impl.h]
Every instruction is attributed to the line of the kernel body from which the code it belongs to was inlined (the last frame of
the .loc inline chain), so the stages called from the frame loop show up with the size of THAT copy."""
import collections
import re
import sys

asm, kern = sys.argv[1], sys.argv[2]
impl = sys.argv[3] if len(sys.argv) > 3 else "/root/repo/wenet_amd/csrc/demod_oct_impl.h"
base = impl.split("/")[-1].replace(".", r"\.")
src = open(impl).read().splitlines()
per = collections.defaultdict(collections.Counter)
cur, inside = -1, False
for ln in open(asm):
    s = ln.strip()
    if re.match(r"^_Z\w+:", s):
        inside = kern in s
        continue
    if s.startswith(".Lfunc_end"):
        inside = False
    if not inside:
        continue
    if s.startswith(".loc"):
        chain = re.findall(base + r":(\d+):\d+", s)
        cur = int(chain[-1]) if chain else -1
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    c = per[cur]
    c["all"] += 1
    if op.startswith("v_"):
        c["valu"] += 1
        if op.startswith("v_pk_"):
            c["pk"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        c["vmem"] += 1
    elif op.startswith("s_nop"):
        c["nop"] += 1
    elif op.startswith("s_waitcnt"):
        c["wait"] += 1
    else:
        c["salu"] += 1
tot = collections.Counter()
for line, c in sorted(per.items()):
    tot += c
    if c["all"] >= 6:
        print(f"{line:5d} all{c['all']:5d} v{c['valu']:5d} pk{c['pk']:4d} lds{c['lds']:4d} vm{c['vmem']:4d} s{c['salu']:4d} nop{c['nop']:4d} w{c['wait']:3d} | "
              f"{src[line - 1].strip()[:110] if 0 < line <= len(src) else ''}")
print(f"total all{tot['all']:5d} v{tot['valu']:5d} pk{tot['pk']:4d} lds{tot['lds']:4d} vm{tot['vmem']:4d} s{tot['salu']:4d} nop{tot['nop']:4d} w{tot['wait']:3d}")
