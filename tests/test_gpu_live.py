"""GPU: live channels (wenet_rx_push / wenet_rx_flush) -- N streams fed in ragged ticks with everything carried on the GPU must equal, per channel
and bit for bit, ONE run of the oracle over the concatenated samples: soft decisions, packet bytes, iteration counts, CRC flags, LLRs and the
packets' positions in the symbol stream.  (BASELINE config 5 taken as written: 128 CONCURRENT channels; per channel the loops of
src/fsk_demod.c:270-413 and src/wenet_ldpc.c:171-258 / src/drs232_ldpc.c:176-274.)"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal
from wenet_amd import siggen
from wenet_amd.fsk import BYTES_PER_SAMPLE
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


def _pinned_copy(b, shift):
    """the bytes of b in PINNED host memory, `shift` bytes behind the start of the allocation (any alignment of the chunks)"""
    import torch
    t = torch.empty(b.size + shift + 16, dtype=torch.uint8).pin_memory()
    a = t.numpy()[shift:shift + b.size]
    a[:] = b
    return a, t


def _run_live(cfg, caps, fmt, cuts, want_llr=False, max_iter=10, pinned=False):
    """push every channel's capture in the ticks `cuts[ch]` gives (sample counts per tick, 0 allowed); returns per channel the concatenated results"""
    n = len(caps)
    bps = BYTES_PER_SAMPLE[fmt]
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode, max_iter=max_iter)
    if want_llr:
        rx.enable_llr_dump()
    raw = [ol.raw_bytes(c) for c in caps]
    keep = []
    is_pinned = [bool(pinned) and (pinned != "mixed" or ch % 2 == 0) for ch in range(n)]      # "mixed": every other channel stays in pageable memory
    if pinned:                                                            # channel ch's buffer starts ch bytes into its allocation: every source alignment
        for ch in range(n):
            if is_pinned[ch] and pinned == "registered":                  # the caller's own (pageable) array, pinned through the library (wenet_rx_pin_host)
                raw[ch] = raw[ch].copy()
                rx.pin(raw[ch])
            elif is_pinned[ch]:
                raw[ch], t = _pinned_copy(raw[ch], ch % 16)
                keep.append(t)
    pos = [0] * n
    out = [dict(sd=[], bytes=[], iter=[], ok=[], start=[], llr=[]) for _ in range(n)]
    nt = max(len(c) for c in cuts)
    reported = 0
    for t in range(nt):
        chunks = []
        for ch in range(n):
            k = cuts[ch][t] if t < len(cuts[ch]) else 0
            chunks.append(raw[ch][pos[ch] * bps:(pos[ch] + k) * bps])
            pos[ch] += k
        if pinned and t % 2:                                              # the address form of the same call (RxBatch.push_ptrs)
            got = rx.push_ptrs(np.array([c.ctypes.data if c.size else 0 for c in chunks], np.uint64), np.array([c.size // bps for c in chunks], np.int64), fmt)
        else:
            got = rx.push(chunks, fmt)
        assert rx.live_gathered() == sum(1 for ch, c in enumerate(chunks) if c.size and is_pinned[ch])
        tick_pk = 0
        for ch in range(n):
            out[ch]["sd"].append(rx.soft(ch).copy())
            p = rx.packets(ch)
            tick_pk += p["n"]
            if p["n"]:
                out[ch]["bytes"].append(p["bytes"].copy()); out[ch]["iter"].append(p["iter"]); out[ch]["ok"].append(p["crc_ok"]); out[ch]["start"].append(p["start"])
                if want_llr:
                    out[ch]["llr"].append(rx.llrs(ch).copy())
        assert got == tick_pk
        reported += got
    for ch in range(n):
        assert pos[ch] * bps == raw[ch].size, "the cuts must cover the capture"
    frames = [rx.frames(ch) for ch in range(n)]
    rx.flush()
    if pinned == "registered":
        for ch in range(n):
            rx.unpin(raw[ch])
    rx.close()
    return out, frames, reported


def _check(cfg, caps, fmt, out, frames, want_llr=False, max_iter=10):
    total = 0
    for ch, raw in enumerate(caps):
        sd, _ = ol.oracle_demod(raw, fmt, cfg.Fs, cfg.Rs, cfg.M)
        ref = ol.oracle_deframe(sd, cfg.mode, max_iter=max_iter, want_llr=want_llr)
        got_sd = np.concatenate(out[ch]["sd"]) if out[ch]["sd"] else np.zeros(0, np.float32)
        assert bits_equal(got_sd, sd), f"channel {ch}: soft decisions"
        assert frames[ch] * (48 if cfg.M == 2 else 96) == sd.size
        nb = sum(len(b) for b in out[ch]["bytes"])
        assert nb == ref["n"], f"channel {ch}: {nb} packets, oracle {ref['n']}"
        if nb:
            assert (np.concatenate(out[ch]["bytes"]) == ref["bytes"]).all(), ch
            assert (np.concatenate(out[ch]["iter"]) == ref["iter"]).all(), ch
            assert (np.concatenate(out[ch]["ok"]) == ref["crc_ok"]).all(), ch
            assert (np.concatenate(out[ch]["start"]) == ref["start"]).all(), ch
            if want_llr:
                assert bits_equal(np.concatenate(out[ch]["llr"]), ref["llr"]), ch
        total += nb
    return total


def _ragged_cuts(rng, nsamp, mean):
    cuts, left = [], nsamp
    while left > 0:
        k = int(rng.integers(0, 2 * mean))
        if rng.random() < 0.1:
            k = 0                                                         # a tick in which nothing arrived for this channel
        k = min(k, left)
        cuts.append(k)
        left -= k
    return cuts


def test_128_channels_in_ragged_ticks_equal_the_oracle_one_shot():
    cfg = siggen.config_v2()
    rng = np.random.default_rng(77)
    made = [siggen.make_capture(cfg, int(rng.integers(5, 9)), 8.0, seed=7700 + ch, ppm=float(rng.choice([0.0, 60.0, -90.0]))) for ch in range(128)]
    caps = [m[0] for m in made]
    cuts = [_ragged_cuts(rng, c.size // 2, 30000) for c in caps]            # ~31 ms ticks on average, anything from 0 to 62 ms
    out, frames, reported = _run_live(cfg, caps, "cu8", cuts, want_llr=True)
    total = _check(cfg, caps, "cu8", out, frames, want_llr=True)
    assert reported == total and total > 0.7 * sum(len(m[1]) for m in made) - 128


@pytest.mark.parametrize("name,fmt,mean", [("v1", "cu8", 9000), ("v2", "cs16", 1500), ("v2", "cf32", 50000), ("4fsk", "cu8", 40000)])
def test_live_formats_framings_and_tiny_ticks(name, fmt, mean):
    """ticks much shorter than a modem frame (several pushes per frame, many with no frame at all), every input format, both framings, 4-FSK"""
    cfg = siggen.CONFIGS[name]()
    rng = np.random.default_rng(abs(hash((name, fmt))) % 1000)
    nch = 5
    caps = [siggen.make_capture(cfg, 3, 9.0, seed=900 + ch, fmt=fmt, ppm=(150.0 if ch == 1 else 0.0))[0] for ch in range(nch)]
    bps = BYTES_PER_SAMPLE[fmt]
    cuts = [_ragged_cuts(rng, ol.raw_bytes(c).size // bps, mean) for c in caps]
    out, frames, _ = _run_live(cfg, caps, fmt, cuts)
    assert _check(cfg, caps, fmt, out, frames) > 0


@pytest.mark.parametrize("name,fmt,mean,pinned", [("v2", "cu8", 20000, True), ("v1", "cs16", 700, True), ("v2", "cf32", 30001, True), ("v2", "cu8", 12000, "mixed"), ("v2", "cu8", 15000, "registered")])
def test_live_pinned_buffers_are_read_by_the_gpu_itself(name, fmt, mean, pinned):
    """chunks in pinned host memory are gathered by ONE kernel over PCIe (no copy per channel): every alignment of source (the buffers start 0..15 bytes
    into their allocations, the ticks cut them anywhere) and destination (behind whatever the last tick left), tiny and empty chunks; same results"""
    cfg = siggen.CONFIGS[name]()
    rng = np.random.default_rng(abs(hash((name, fmt, "pin"))) % 1000)
    nch = 19
    caps = [siggen.make_capture(cfg, 3, 9.0, seed=1900 + ch, fmt=fmt, ppm=(120.0 if ch == 2 else 0.0))[0] for ch in range(nch)]
    bps = BYTES_PER_SAMPLE[fmt]
    cuts = [_ragged_cuts(rng, ol.raw_bytes(c).size // bps, mean) for c in caps]
    out, frames, _ = _run_live(cfg, caps, fmt, cuts, pinned=pinned)
    assert _check(cfg, caps, fmt, out, frames) > 0


@pytest.mark.parametrize("switch", ["WENET_RX_NO_LIVE_OVERLAP", "WENET_RX_LIVE_GATHER_SHARED_CU", "WENET_RX_LIVE_COPIES"])
@pytest.mark.parametrize("pinned", [True, False])
def test_live_tick_paths_behind_their_switches(monkeypatch, switch, pinned):
    """Round 5: by default the pipelined demodulator runs BESIDE the gather kernel and takes the chunks piece by piece as they cross PCIe (arrival words, agent-scope
    loads), the gather on compute units of its own (LDS reservation), the results leave through one export kernel.  The older paths stay behind switches: the gather
    finished before the demodulator starts; gather and demodulator sharing compute units; results through five copies.  Same bits on every path."""
    monkeypatch.setenv(switch, "1")
    cfg = siggen.config_v2()
    rng = np.random.default_rng(len(switch) + int(bool(pinned)))
    caps = [siggen.make_capture(cfg, 3, 9.0, seed=2900 + ch, ppm=(150.0 if ch == 1 else 0.0))[0] for ch in range(9)]
    cuts = [_ragged_cuts(rng, c.size // 2, 60000) for c in caps]
    out, frames, _ = _run_live(cfg, caps, "cu8", cuts, pinned=pinned)
    assert _check(cfg, caps, "cu8", out, frames) > 0


def test_live_long_ticks_arrive_in_pieces_while_the_demodulator_runs():
    """ticks long enough that every piece of a chunk is many frames (the demodulator really waits for pieces in the middle of its launch), pageable and pinned
    channels side by side, a slipping channel among them; more channels than the gather has workgroups"""
    cfg = siggen.config_v2()
    rng = np.random.default_rng(41)
    caps = [siggen.make_capture(cfg, 12, 8.5, seed=4100 + ch, ppm=(-180.0 if ch == 3 else 0.0))[0] for ch in range(40)]
    cuts = [_ragged_cuts(rng, c.size // 2, 220000) for c in caps]           # ~0.19 s per tick on average
    out, frames, _ = _run_live(cfg, caps, "cu8", cuts, pinned="mixed")
    assert _check(cfg, caps, "cu8", out, frames) > 0


def test_three_handles_tick_at_once_and_fill_the_device_together():
    """Three handles of 128 channels each, ticking at the same time from three threads: 384 demodulator workgroups that wait for their chunks on a device of 256 compute
    units, and three gathers that want compute units to themselves.  Every demodulator is launched behind the gate (a workgroup of its tick's gather is resident), so all
    sets of streams finish every tick; results equal those of a handle that ticks alone."""
    import threading
    import torch
    cfg = siggen.config_v2()
    base = [np.ascontiguousarray(siggen.make_capture(cfg, 12, 9.0, seed=5200 + k)[0]).view(np.uint8).reshape(-1) for k in range(8)]
    n = min(b.size for b in base)
    tick = 2 * 28800                                                        # bytes per tick: 30 ms
    nt = n // tick

    def run(out, idx, go):
        keep = [torch.from_numpy(base[(ch + idx) % 8][:n].copy()).pin_memory() for ch in range(128)]
        host = [t.numpy() for t in keep]
        ptr = np.array([h.ctypes.data for h in host], np.uint64)
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        go.wait()
        dig = []
        for t in range(nt):
            npk = rx.push_ptrs(ptr + np.uint64(t * tick), np.full(128, tick // 2, np.int64), "cu8")
            dig.append((npk, rx.result_digest()))
        rx.flush(); rx.close()
        out[idx] = dig

    alone, go = {}, threading.Event()
    go.set()
    for i in range(3):
        run(alone, i, go)
    assert sum(d[0] for d in alone[0]) > 128 * 8
    for rep in range(3):
        both, go = {}, threading.Event()
        th = [threading.Thread(target=run, args=(both, i, go)) for i in range(3)]
        for t in th: t.start()
        go.set()
        for t in th: t.join()
        assert all(both[i] == alone[i] for i in range(3)), rep


@pytest.mark.parametrize("pinned", [True, "mixed"])
def test_live_channels_through_the_three_capture_kernel(monkeypatch, pinned):
    """the three-captures-per-workgroup pipelined kernel (what 1.5 - 3 channels per compute unit take) with carried state, its chunks arriving in pieces beside it
    (round 5); 23 channels: the last workgroup carries two captures"""
    monkeypatch.setenv("WENET_RX_TRI", "1")
    cfg = siggen.config_v2()
    rng = np.random.default_rng(15)
    caps = [siggen.make_capture(cfg, 6, 8.5, seed=3300 + ch, ppm=(200.0 if ch % 5 == 0 else 0.0))[0] for ch in range(23)]
    cuts = [_ragged_cuts(rng, c.size // 2, 90000) for c in caps]
    out, frames, _ = _run_live(cfg, caps, "cu8", cuts, pinned=pinned)
    assert _check(cfg, caps, "cu8", out, frames) > 0


def test_live_many_channels_through_the_batch_demodulator(monkeypatch):
    """enough channels that the per-tick launch takes the batch demodulator (one wavefront per capture) with carried state"""
    monkeypatch.setenv("WENET_RX_OCT", "7")
    cfg = siggen.config_v2()
    rng = np.random.default_rng(5)
    caps = [siggen.make_capture(cfg, 3, 8.5, seed=300 + ch, ppm=(200.0 if ch % 5 == 0 else 0.0))[0] for ch in range(23)]
    cuts = [_ragged_cuts(rng, c.size // 2, 25000) for c in caps]
    out, frames, _ = _run_live(cfg, caps, "cu8", cuts)
    assert _check(cfg, caps, "cu8", out, frames) > 0


def test_push_rejects_a_changed_channel_set_and_flush_reopens():
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 2, 12.0, seed=3)
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    rx.push([raw[:40000], raw[:40000]], "cu8")
    with pytest.raises(RuntimeError):
        rx.push([raw[:40000]], "cu8")                                       # one channel where two are open
    with pytest.raises(RuntimeError):
        rx.push([raw[:40000], raw[:40000]], "cs16")                         # another sample format
    rx.flush()
    got = rx.push([raw], "cu8")                                            # a fresh stream: the whole capture in one tick = the batch result
    sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    ref = ol.oracle_deframe(sd, cfg.mode)
    assert got == ref["n"] and bits_equal(rx.soft(0), sd) and (rx.packets(0)["bytes"] == ref["bytes"]).all()
    # a batch on the same handle ends the live streams and still works
    rx.process([raw], "cu8")
    assert bits_equal(rx.soft(0), sd)
    rx.close()
