#!/usr/bin/env python3
"""Development aid (round 4): demod time of a device-resident batch against the length of the time slices inside one launch of the batch demodulator
(WENET_RX_DEV_SLICE_SAMPLES; 0 = the library's own choice, "off" = WENET_RX_NO_DEV_SLICES).
usage: gpu_dev_slices.py [seconds] B[,B...] slice[,slice...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = sys.argv[1] if len(sys.argv) > 1 else "10"
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3584,4000").split(",")]
slices = (sys.argv[3] if len(sys.argv) > 3 else "off,0,150000,300000,600000,1200000,2400000").split(",")
for B in Bs:
    for sl in slices:
        e = dict(os.environ)
        if sl == "off":
            e["WENET_RX_NO_DEV_SLICES"] = "1"
        elif sl != "0":
            e["WENET_RX_DEV_SLICE_SAMPLES"] = sl
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--captures", str(B), "--seconds", secs, "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline", "--no-extras"], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"B={B} slice={sl}: demod {d['kernel_ms']['demod']:.1f} ms, step {d['ms_per_step']:.1f} ms, {d['value'] / 1e3:.1f} G samples/s, {d['packets_valid_per_step_rank0']} packets", flush=True)
