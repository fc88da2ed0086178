#!/usr/bin/env python3
"""Static instruction counts of one kernel instantiation attributed to source regions (development aid).

    python tools/isa_by_line.py [--kernel ILi2ELi10ELi256E] [--src wenet_amd/csrc/demod_oct.hip] [--lines]

Compiles the device side with -gline-tables-only -S, walks the assembly of the chosen kernel, attributes every instruction to
the source line of its last .loc, and sums over the regions named by `// @region name` ... markers or, by default, over the
lambdas / blocks of demod_oct_impl.h found by their opening lines.  Static counts: a loop body counts once."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-gpu-flush-denormals-to-zero -Wno-everything --cuda-device-only -S -gline-tables-only").split()

REGION_STARTS = [  # (regex on a source line of demod_oct_impl.h, region name); a region runs to the next start
    (r"^template <int M, int TS, int NDFT", "prologue"),
    (r"auto prefetch_est = ", "prefetch_est"),
    (r"auto prefetch_slot = ", "prefetch_slot"),
    (r"auto slot_align = ", "slot_align/sample"),
    (r"auto bfly4 = ", "estimate_fft"),
    (r"auto estimate_pick_to = ", "estimate_pick"),
    (r"oct_g_f32x2 \*Fscr = ", "dstage"),
    (r"float t_rxt = 0.f", "tstage2"),
    (r"auto chain_split = ", "chain"),
    (r"auto tsum = ", "tsum"),
    (r"auto alive_mask = ", "frame_loop_setup"),
    (r"if \(is_chain\) \{$", "duty_loop"),
    (r"^        \} else \{$", "capture_loop"),
    (r"// =+ save carried state", "epilogue"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ILi2ELi10ELi256E")
    ap.add_argument("--src", default="wenet_amd/csrc/demod_oct.hip")
    ap.add_argument("--impl", default="wenet_amd/csrc/demod_oct_impl.h")
    ap.add_argument("--lines", action="store_true", help="per-line counts of the impl file")
    ap.add_argument("--extra", default="", help="extra compiler flags")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + a.extra.split() + ["-I" + os.path.join(ROOT, "wenet_amd", "csrc"), "-o", out, os.path.join(ROOT, a.src)])
        asm = open(out).read().splitlines()
    files, cur_file, cur_line, inside = {}, None, 0, False
    per = collections.defaultdict(collections.Counter)          # (file, line) -> Counter
    for ln in asm:
        s = ln.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", s)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        if re.match(r"^_Z\w+:", s):
            inside = a.kernel in s
            continue
        if s.startswith(".Lfunc_end"):
            inside = False
        if not inside:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur_file, cur_line = files.get(int(m.group(1)), "?"), int(m.group(2))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = per[(os.path.basename(cur_file or "?"), cur_line)]
        c["all"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if op.startswith("v_pk_"):
                c["pk"] += 1
            if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                c["lane"] += 1
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
    impl = os.path.basename(a.impl)
    src = open(os.path.join(ROOT, a.impl)).read().splitlines()
    starts = []
    for i, l in enumerate(src, 1):
        for rx, name in REGION_STARTS:
            if re.search(rx, l) and name not in [n for _, n in starts]:
                starts.append((i, name))
    starts.sort()
    tot = collections.defaultdict(collections.Counter)
    for (f, line), c in per.items():
        if f != impl:
            tot["(other files: " + f + ")"] += c
            continue
        name = "?"
        for i, n in starts:
            if line >= i:
                name = n
        tot[name] += c
    print(f"{'region':34s} {'all':>6s} {'valu':>6s} {'pk':>5s} {'lane':>5s} {'lds':>5s} {'vmem':>5s} {'salu':>5s} {'nop':>5s} {'wait':>5s}")
    for name, c in sorted(tot.items(), key=lambda x: -x[1]["all"]):
        print(f"{name:34s} {c['all']:6d} {c['valu']:6d} {c['pk']:5d} {c['lane']:5d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:5d} {c['nop']:5d} {c['wait']:5d}")
    s = sum(tot.values(), collections.Counter())
    print(f"{'total':34s} {s['all']:6d} {s['valu']:6d} {s['pk']:5d} {s['lane']:5d} {s['lds']:5d} {s['vmem']:5d} {s['salu']:5d} {s['nop']:5d} {s['wait']:5d}")
    if a.lines:
        for (f, line), c in sorted(per.items()):
            if f == impl and c["all"] >= 3:
                print(f"{line:5d} {c['all']:5d} v{c['valu']:4d} pk{c['pk']:4d} lds{c['lds']:4d} nop{c['nop']:3d} w{c['wait']:3d} | {src[line - 1].strip()[:120]}")


if __name__ == "__main__":
    main()
