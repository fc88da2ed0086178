#!/usr/bin/env python3
"""Development aid: demod kernel time over batch sizes, default kernel choice vs the one-wavefront-per-capture kernel forced.
usage: gpu_batch_sweep.py [seconds] [B ...]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = sys.argv[1] if len(sys.argv) > 1 else "2"
Bs = [int(x) for x in sys.argv[2:]] or [128, 256, 512, 768, 1024, 1536, 2048, 3584]
for B in Bs:
    row = []
    for tag, env in (("lib", {"WENET_RX_NO_OCT": "1"}), ("oct7", {"WENET_RX_OCT": "7"}), ("oct4", {"WENET_RX_OCT": "4"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--captures", str(B), "--seconds", secs, "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline", "--no-extras"], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        row.append(f"{tag}: {d['roofline']['kernel'].replace('wenet_demod_', '')} demod {d['kernel_ms']['demod']:.1f} ms")
    print(f"B={B}: " + " | ".join(row), flush=True)
