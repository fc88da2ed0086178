#!/usr/bin/env python3
"""ISA evidence from the shipped build: registers, spills, LDS and instruction mix of every kernel in wenet_amd/csrc/*.o
(the objects libwenet_rx.so is linked from), read with llvm-readelf / llvm-objdump from the gfx950 code objects.

    python tools/isa_summary.py [-o profiles/r03_isa_summary.txt]

Also prints `source_sha16`: the identity of the demodulator sources that tools/gpu_profile_round.sh stamps into the PMC profile
and bench.py checks before it quotes that profile (wenet_amd/codeid.py)."""
import argparse
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, ROOT)


def code_objects(obj, td):
    """gfx950 code objects bundled in a host object / shared library"""
    local = os.path.join(td, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return sorted(glob.glob(local + ".*gfx950*"))


def kernel_meta(co):
    """per kernel: the fields of its `amdhsa.kernels` entry in the code object's metadata note.  (The note is YAML and a kernel's keys are sorted: `.group_segment_fixed_size`
    comes BEFORE `.name` -- the line-by-line reader of rounds 3-4 attached it to the previous kernel, so the LDS column of those summaries is shifted by one row.)"""
    import yaml
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True, check=False).stdout
    out = {}
    for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", txt, flags=re.S | re.M):
        try:
            y = yaml.safe_load(doc)
        except yaml.YAMLError:
            continue
        for k in (y or {}).get("amdhsa.kernels", []):
            name = k.get(".name", "")
            out[name] = {f.lstrip("."): v for f, v in k.items() if isinstance(v, int)}
    return {k: v for k, v in out.items() if "vgpr_count" in v}


def kernel_mix(co):
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], stdout=subprocess.PIPE, text=True, check=False).stdout
    mix, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = mix.setdefault(m.group(1), collections.Counter())
            continue
        if cur is None:
            continue
        t = line.split("//")[0].split()
        if not t:
            continue
        op = t[0]
        cur["insts"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith("v_pk_"):
                cur["valu_packed"] += 1
            if "dpp" in line or "quad_perm" in line or "row_" in line or "wave_sh" in line:
                cur["valu_dpp"] += 1
            if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                cur["v_lane_moves"] += 1
            if op.startswith("v_mfma"):
                cur["mfma"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            cur["vmem"] += 1
            if op.startswith("scratch_"):
                cur["scratch"] += 1
        elif op.startswith("s_waitcnt"):
            cur["s_waitcnt"] += 1
        elif op.startswith("s_nop"):
            cur["s_nop"] += 1
        elif op.startswith("s_barrier"):
            cur["s_barrier"] += 1
        elif op.startswith("s_"):
            cur["salu_other"] += 1
    return mix


def demangle(names):
    if not names:
        return {}
    try:
        out = subprocess.run(["c++filt"] + list(names), stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("objects", nargs="*", default=sorted(glob.glob(os.path.join(ROOT, "wenet_amd", "csrc", "*.o"))))
    a = ap.parse_args()
    from wenet_amd import codeid
    lines = [f"# ISA summary of the shipped kernels (tools/isa_summary.py; hipcc {codeid.hipcc_version()})",
             f"# source_sha16 {codeid.source_sha16()}  (wenet_amd/codeid.py: the demodulator / decoder sources + build flags)",
             "# columns: VGPR SGPR | spills VGPR SGPR | LDS bytes (static) scratch bytes | instructions: all VALU (packed, DPP, lane moves) LDS VMEM s_waitcnt s_nop"]
    with tempfile.TemporaryDirectory() as td:
        for obj in a.objects:
            for co in code_objects(obj, td):
                meta, mix = kernel_meta(co), kernel_mix(co)
                names = demangle([k for k in meta])
                for k in sorted(meta, key=lambda x: names[x]):
                    m, c = meta[k], mix.get(k, collections.Counter())
                    short = re.sub(r"\(anonymous namespace\)::", "", names[k])
                    short = re.sub(r"\(.*$", "", short)[:70]
                    lines.append(f"{os.path.basename(obj):22s} {short:70s} {m.get('vgpr_count', 0):4d} {m.get('sgpr_count', 0):4d} | "
                                 f"{m.get('vgpr_spill_count', 0):4d} {m.get('sgpr_spill_count', 0):4d} | {m.get('group_segment_fixed_size', 0):6d} "
                                 f"{m.get('private_segment_fixed_size', 0):5d} | {c['insts']:6d} {c['valu']:6d} ({c['valu_packed']:5d}, {c['valu_dpp']:4d}, {c['v_lane_moves']:4d}) "
                                 f"{c['lds']:5d} {c['vmem']:5d} {c['s_waitcnt']:5d} {c['s_nop']:5d}")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
