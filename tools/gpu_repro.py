"""Development aid: reproducibility of the decode step.  One batch (B captures x S seconds, generated on the GPU as bench.py does, seeded) is processed N times in one
process; every pass's packets (bytes, iteration counts, CRC flags) are compared with the first pass's, and differing packets are printed.
With a canary build of the library (-DWR_DEC_CANARY: tools/variant_build.sh <name> "-DWR_DEC_CANARY ..." ldpc_kernel wenet_rx; WENET_RX_LIB=...) every wavefront of the
decoder also leaves a record per packet (compute unit, slot it believed it decoded, iteration at which it left the loop, barriers passed); the tool then checks that the eight
wavefronts of a packet agree and prints the records of every packet that deviates -- and a histogram of the compute units the deviations happened on.
usage: gpu_repro.py [captures] [seconds] [passes]"""
import ctypes as C, os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from wenet_amd import siggen, lib as _lib
from wenet_amd.rx import RxBatch
from wenet_amd.tx import Tx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = siggen.config_v2()
dev = torch.device("cuda:0")
L = _lib.load()
canary = hasattr(L, "wenet_rx_debug_canary")
try:
    uid = [l.split()[-1] for l in subprocess.run(["rocm-smi", "--showuniqueid"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=60).stdout.splitlines() if "Unique ID" in l and "GPU[" in l][0]
except Exception:
    uid = "?"
print(f"library {_lib.LIB_PATH} source {L.wenet_rx_source_id().decode()} canary {canary}; host {os.uname().nodename} gpu unique_id {uid} {torch.cuda.get_device_name(0)}", flush=True)
nsamp = int(secs * cfg.Fs); nsym = nsamp // (cfg.Fs // cfg.Rs)
tx = Tx.from_config(cfg)
spp = tx.symbols_per_packet
nfr = nsym // spp + 1
g = torch.Generator(device=dev); g.manual_seed(2001)
payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], [8.0] * B, seeds=[7000 + i for i in range(B)])
torch.cuda.synchronize()
rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
ptrs = [int(c.data_ptr()) for c in caps]; ns = [nsamp] * B

if canary:
    L.wenet_rx_debug_canary.restype = C.c_longlong; L.wenet_rx_debug_canary.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    L.wenet_rx_debug_slots.restype = C.c_longlong; L.wenet_rx_debug_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    L.wenet_rx_debug_canary_iters.restype = C.c_longlong; L.wenet_rx_debug_canary_iters.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]

def raw(fn, rec):
    n = -int(fn(rx._h, None, 0))
    buf = np.empty(n, np.uint8)
    assert fn(rx._h, buf.ctypes.data, n) == n
    return buf.reshape(-1, rec)

def snapshot():
    out = []
    for ch in range(B):
        p = rx.packets(ch)
        out.append((p["bytes"].copy(), p["iter"].copy(), p["crc_ok"].copy()))
    return out

def cu_of(hw, xcc):        # (xcc, se, sh, cu)
    return (int(xcc) & 0xf, (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)

ref = None
ndiff_passes = 0
bad_cus = {}
for it in range(passes):
    rx.enqueue_device(ptrs, ns, "cu8"); rx.collect()
    if canary:
        slots = raw(L.wenet_rx_debug_slots, 276)
        cn = raw(L.wenet_rx_debug_canary, 256).view(np.uint32).reshape(-1, 8, 8)          # [slot][wave][word]
        used = cn[:, 0, 0] != 0
        agree = (cn[:, :, 2] == cn[:, :1, 2]).all(1) & (cn[:, :, 3] == cn[:, :1, 3]).all(1) & (cn[:, :, 4] == cn[:, :1, 4]).all(1)
        odd = np.nonzero(used & ~agree)[0]
        if ref is None:
            ref = slots.copy()
            print(f"pass 0: {int(used.sum())} packets decoded, {int(slots[used, 258].sum())} valid", flush=True)
        diff = np.nonzero((slots[:, :272] != ref[:, :272]).any(1))[0]
        rep = sorted(set(odd.tolist()) | set(diff.tolist()))
        for sidx in rep:
            nb = int((slots[sidx, :258] != ref[sidx, :258]).sum())
            wb = sorted(set((np.nonzero(slots[sidx, :258] != ref[sidx, :258])[0] // 64).tolist()))
            it_now, it_ref = int(slots[sidx, 260:264].view(np.int32)[0]), int(ref[sidx, 260:264].view(np.int32)[0])
            r = cn[sidx]
            cu = cu_of(r[0, 0], r[0, 1])
            bad_cus[cu] = bad_cus.get(cu, 0) + 1
            print(f"pass {it} slot {sidx}: {nb} bytes differ (64-byte runs {wb}), iter {it_ref} -> {it_now}; CU xcc/se/sh/cu {cu} wg {int(r[0,5])>>12} seq {int(r[0,5])&0xfff}")
            for w in range(8):
                print(f"    wave {w}: simd {(int(r[w,0])>>4)&3} hwslot {int(r[w,0])&15} pipe {(int(r[w,0])>>6)&3} tg {(int(r[w,0])>>16)&15} queue {(int(r[w,0])>>24)&7} me {(int(r[w,0])>>30)&3} slot-seen {int(r[w,2])} left-at {int(r[w,3])} barriers {int(r[w,4])} pcc {int(r[w,6])} t {int(r[w,7])}" + (f" counts seen, last six iterations: {[(int(r[w,7]) >> (10 * k)) & 0x3ff for k in range(2, -1, -1)] + [(int(r[w,6]) >> (10 * k)) & 0x3ff for k in range(2, -1, -1)]}" if os.environ.get("CANARY_HIST") else ""))
            its = np.zeros((8, 10, 8), np.uint32)
            got = L.wenet_rx_debug_canary_iters(rx._h, int(sidx), its.ctypes.data)
            if got == 2560 and os.environ.get("CANARY_LEVEL") == "10":
                for w in range(8):
                    x = its[w].reshape(-1)
                    if x[67]:
                        print(f"    wave {w}: at the top of iteration {int(x[66])}: carried count {int(x[68])}, the cell read again {int(x[69])} (lanes that differ {(int(x[65]) << 32) | int(x[64]):016x}), its flag {int(x[70])}, the other parity's count cell {int(x[71])}")
            elif got == 2560 and os.environ.get("CANARY_LEVEL") in ("8", "9"):
                for w in range(8):
                    x = its[w].reshape(-1)
                    print(f"    wave {w}: count carried into iteration 1.. : {[int(v) & 0xffff for v in x[1:12] if v]}")
            elif got == 2560 and os.environ.get("CANARY_LEVEL") in ("6", "7"):
                for w in range(8):
                    x = its[w].reshape(-1)
                    if x[64] or x[65] or x[67]:
                        m = (int(x[65]) << 32) | int(x[64])
                        print(f"    wave {w}: iteration {int(x[66])}: lanes whose count != 516: {m:016x}; tag {int(x[67]):x}; counts by lane: {[int(v) for v in x[:64]]}")
            elif got == 2560 and os.environ.get("CANARY_LEVEL") == "4":
                for w in range(8):
                    x = its[w, 0]
                    print(f"    wave {w}: lanes with ssum != 516 when only SOME had: {(int(x[3]) << 32) | int(x[2]):016x} (iteration {int(x[4])}); lanes with any != 0 when only some had: {(int(x[6]) << 32) | int(x[5]):016x} (iteration {int(x[0])})")
            elif got == 2560 and its[:, :, 0].any() and not (its[:, 0, 0] & 0x80000000).any():            # level 3: histories kept in registers (six last iterations' sums, oldest first)
                for w in range(8):
                    x = its[w, 0]; h = (int(x[1]) << 32) | int(x[0])
                    print(f"    wave {w}: sums seen (last six iterations) {[(h >> (10 * k)) & 0x3ff for k in range(5, -1, -1)]} lanes that ever differed from lane 0: {(int(x[3]) << 32) | int(x[2]):x}; any bits {int(x[4]):b} lanes {(int(x[6]) << 32) | int(x[5]):x}")
            elif got == 2560 and its[:, :, 0].any():
                for w in range(8):
                    print(f"    wave {w} per iteration (ssum[lanes that differ] any[lanes] exec):", " | ".join(
                        f"{int(x[0]) & 0x7fffffff}[{(int(x[2]) << 32) | int(x[1]):x}] {int(x[3])}[{(int(x[5]) << 32) | int(x[4]):x}] {'full' if (int(x[6]), int(x[7])) == (0xffffffff, 0xffffffff) else hex((int(x[7]) << 32) | int(x[6]))}"
                        for x in its[w] if x[0]))
            # the packets the same workgroup decoded just before / after
            wg, seq = int(r[0, 5]) >> 12, int(r[0, 5]) & 0xfff
            near = np.nonzero(used & ((cn[:, 0, 5] >> 12) == wg) & (np.abs((cn[:, 0, 5] & 0xfff).astype(np.int64) - seq) <= 2))[0]
            print("    same workgroup, neighbouring packets:", [(int(x), int(cn[x, 0, 5]) & 0xfff, "differs" if (slots[x, :272] != ref[x, :272]).any() else "same") for x in near])
        ndiff_passes += len(rep) > 0
        continue
    dg = rx.result_digest()
    if ref is not None and dg == ref_dg:                 # (the digest covers every packet's bytes, flag, iteration count and position: equal = nothing to list)
        continue
    s = snapshot()
    if ref is None:
        ref_dg = dg
        ref = s
        print("pass 0:", sum(len(x[1]) for x in s), "packets,", int(sum(x[2].sum() for x in s)), "valid", flush=True)
        continue
    d = 0
    for ch in range(B):
        a, b = ref[ch], s[ch]
        if len(a[1]) != len(b[1]):
            print(f"pass {it} capture {ch}: {len(b[1])} packets, first pass {len(a[1])}"); d += 1; continue
        bad = np.nonzero((a[1] != b[1]) | (a[2] != b[2]) | (a[0] != b[0]).any(axis=1))[0] if len(a[1]) else []
        for k in bad:
            nbytes = int((a[0][k] != b[0][k]).sum())
            print(f"pass {it} capture {ch} packet {k}: iter {a[1][k]} -> {b[1][k]}, crc {a[2][k]} -> {b[2][k]}, {nbytes} bytes differ")
            d += 1
    ndiff_passes += d > 0
print(f"packets the agreement guard decoded again: {rx.decoder_repeats()}")
print(f"{passes} passes, {ndiff_passes} with differing packets" + (f"; deviations by compute unit (xcc, se, sh, cu): {bad_cus}" if canary else ""))
