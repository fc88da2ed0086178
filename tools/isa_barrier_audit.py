#!/usr/bin/env python3
"""ISA audit (round 5): every s_barrier of every shipped kernel must be preceded, on every path, by a completed wait for the wavefront's LDS stores.

Why: hipcc 7.2 dropped the `s_waitcnt lgkmcnt(0)` that __syncthreads()'s release fence asks for at the barrier that heads wenet_decode_kernel's packet loop -- the stores
in flight arrive over the loop's back edges (thread 0's claim of the next packet slot: ds_write_b32, s_branch, s_barrier), and the wait-count pass had removed the fence's
"soft" wait when it first visited the header, before it knew them.  On gfx950 another wavefront's read behind the barrier then overtook the store about once in 10^6
packets (tools/experiments/README.md, round 5).  The product's barriers now carry a written-out wait (WR_LDS_BARRIER, lds_barrier); this tool checks the code objects.

For each s_barrier: walk backwards through the instruction stream -- through fall-through predecessors and through every branch that targets a label on the way -- until an
`s_waitcnt` whose lgkmcnt field is 0 (or the kernel's entry).  A DS store / atomic met before that is a finding.  (Conservative: a path the program cannot take still counts.)

    python tools/isa_barrier_audit.py [objects or libraries ...]          default: wenet_amd/csrc/*.o      exit status 1 if anything is found
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from isa_summary import LLVM, ROOT, code_objects, demangle

STORE = re.compile(r"^(ds_write|ds_add|ds_sub|ds_or|ds_and|ds_xor|ds_max|ds_min|ds_inc|ds_dec|ds_cmpst|ds_wrxchg|ds_append|ds_consume|ds_pk_add|ds_rsub|ds_mskor)")


def waits_lgkm0(line):
    """s_waitcnt with lgkmcnt(0) (the assembler prints only the fields that are waited for)"""
    if not line.startswith("s_waitcnt"):
        return False
    m = re.search(r"lgkmcnt\((\d+)\)", line)
    if m:
        return int(m.group(1)) == 0
    m = re.search(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", line)          # a raw immediate: lgkmcnt = bits 11:8
    return bool(m) and ((int(m.group(1), 0) >> 8) & 0xf) == 0


def kernels(co):
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--symbolize-operands", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=False).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            name = m.group(1)
            if re.match(r"^L\d+$", name) and cur is not None:
                cur.append(("label", name))
            else:
                cur = out.setdefault(name, [])
            continue
        m = re.match(r"^<(L\d+)>:", line.strip())
        if m and cur is not None:
            cur.append(("label", m.group(1)))
            continue
        if cur is None:
            continue
        t = line.split("//")[0].strip()
        if t:
            cur.append(("inst", t))
    return out


def audit(insts):
    labels = {v: i for i, (k, v) in enumerate(insts) if k == "label"}
    branches = {}                                   # label -> indices of branches to it
    for i, (k, v) in enumerate(insts):
        if k == "inst" and (v.startswith("s_cbranch") or v.startswith("s_branch")):
            m = re.search(r"(L\d+)", v)
            if m:
                branches.setdefault(m.group(1), []).append(i)
    findings = []
    for b, (k, v) in enumerate(insts):
        if k != "inst" or not v.startswith("s_barrier"):
            continue
        seen, stack, bad = set(), [b - 1], None
        while stack and bad is None:
            i = stack.pop()
            while i >= 0:
                if i in seen:
                    break
                seen.add(i)
                k2, v2 = insts[i]
                if k2 == "label":
                    for j in branches.get(v2, []):
                        stack.append(j - 1 if not insts[j][1].startswith("s_branch") else j - 1)
                    # fall-through predecessor: the instruction before the label, unless it is an unconditional branch / end
                    if i > 0 and insts[i - 1][0] == "inst" and (insts[i - 1][1].startswith("s_branch") or insts[i - 1][1].startswith("s_endpgm") or insts[i - 1][1].startswith("s_setpc")):
                        break
                    i -= 1
                    continue
                if waits_lgkm0(v2):
                    break
                if v2.startswith("s_barrier") and i != b:           # (an earlier barrier had its own audit; stores between the two are still searched for: go on)
                    pass
                if STORE.match(v2):
                    bad = (i, v2)
                    break
                i -= 1
        if bad is not None:
            findings.append((b, bad))
    return findings


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "wenet_amd", "csrc", "*.o")))
    total = nbar = 0
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            for co in code_objects(obj, td):
                ks = kernels(co)
                names = demangle([k for k in ks])
                for k, insts in ks.items():
                    nb = sum(1 for kk, v in insts if kk == "inst" and v.startswith("s_barrier"))
                    if not nb:
                        continue
                    nbar += nb
                    f = audit(insts)
                    short = re.sub(r"\(.*$", "", names.get(k, k))[:80]
                    print(f"{os.path.basename(obj):22s} {short:80s} {nb:3d} barriers, {len(f)} reached by an LDS store without a completed wait")
                    for b, (i, v) in f:
                        print(f"        s_barrier (instruction {b}) <- {v} (instruction {i})")
                    total += len(f)
    print(f"{nbar} barriers audited, {total} findings")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
