"""Development aid (round 5): what did the eight wavefronts of a packet READ when the decoder's agreement guard lists it?  Needs a library built with -DWR_GUARD_DEBUG
(tools/variant_build.sh gdbg "-DWR_GUARD_DEBUG" ldpc_kernel wenet_rx; WENET_RX_LIB=...): every wavefront then also leaves the SUM of the counts it read, how many "a data bit
is set" flags it saw, where it ran and when.  Runs with WENET_RX_NO_SETTLE=1 so that listed packets stay visible (done == 2).
usage: gpu_guard_events.py [captures] [seconds] [passes]"""
import ctypes as C, os, sys, collections
import numpy as np
os.environ["WENET_RX_NO_SETTLE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from wenet_amd import siggen, lib as _lib
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = siggen.config_v2(); dev = torch.device("cuda:0"); L = _lib.load()
nsamp = int(secs * cfg.Fs); nsym = nsamp // (cfg.Fs // cfg.Rs)
tx = Tx.from_config(cfg); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(2001)
payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], [8.0] * B, seeds=[7000 + i for i in range(B)])
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
ptrs = [int(c.data_ptr()) for c in caps]; ns = [nsamp] * B
for f in (L.wenet_rx_debug_guard, L.wenet_rx_debug_slots): f.restype = C.c_longlong; f.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
def raw(fn, rec):
    n = -int(fn(rx._h, None, 0)); buf = np.empty(n, np.uint8); assert fn(rx._h, buf.ctypes.data, n) == n; return buf.reshape(-1, rec)
kinds = collections.Counter(); oddw = collections.Counter(); nev = 0; npk = 0
for it in range(passes):
    rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
    slots = raw(L.wenet_rx_debug_slots, 276)
    gd = raw(L.wenet_rx_debug_guard, 128).view(np.uint32).reshape(-1, 8, 4)
    npk += int((gd[:, 0, 0] != 0).sum())
    for sidx in np.nonzero(slots[:, 259] == 2)[0]:
        r = gd[sidx]
        sums = [int(x) & 0xfffff for x in r[:, 0]]; flags = [(int(x) >> 20) & 0x7ff for x in r[:, 0]]; res = [int(x) & 0xff for x in r[:, 1]]; pcc = [int(x) >> 8 for x in r[:, 1]]
        simd = [(int(x) >> 4) & 3 for x in r[:, 2]]; present = [bool(x) for x in r[:, 0]]
        key = list(zip(sums, flags, res))
        maj = collections.Counter(key).most_common(1)[0][0]
        odd = [w for w in range(8) if key[w] != maj]
        nev += 1
        for w in odd:
            oddw[w] += 1
            kinds["absent" if not present[w] else "left at another iteration" if res[w] != maj[2] else "sum of counts lower by %d" % (maj[0] - sums[w]) if sums[w] < maj[0] else "sum of counts higher by %d" % (sums[w] - maj[0]) if sums[w] > maj[0] else "flags differ"] += 1
        if nev <= 40:
            print(f"pass {it} slot {sidx}: majority (sum of counts, flags seen, left at) {maj}; last counts {pcc}; simd {simd}; odd: " + "; ".join(f"wave {w}: {key[w] if present[w] else 'absent'} t-t0 {int(r[w,3]) - int(r[0,3])}" for w in odd))
print(f"{passes} passes, {npk} packets, {nev} listed; odd wavefront index: {sorted(oddw.items())}")
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])[:30]: print(f"    {v:5d}  {k}")
