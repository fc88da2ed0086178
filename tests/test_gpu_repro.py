"""GPU: the chain is REPRODUCIBLE -- one batch pushed through the demodulator, deframer and decoder several times gives the same packets (bytes, iteration counts, CRC flags,
positions) and the same soft decisions every time.  (Round 4: a faster phi0 table was rejected because some builds of it decoded a few packets in ten million differently from run
to run -- tools/experiments/README.md; tools/gpu_repro.py is the long form of this test.)  Results must not depend on which workgroup takes which packet or on timing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_same_batch_same_packets_every_pass():
    import torch
    from wenet_amd import siggen
    from wenet_amd.rx import RxBatch
    from wenet_amd.tx import Tx

    cfg = siggen.config_v2()
    B, secs, passes = 768, 1.0, 12
    dev = torch.device("cuda:0")
    nsamp = int(secs * cfg.Fs)
    nsym = nsamp // (cfg.Fs // cfg.Rs)
    tx = Tx.from_config(cfg)
    spp = tx.symbols_per_packet
    nfr = nsym // spp + 1
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
    symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
    tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
    caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
    # a spread of signal qualities: packets that decode in 3 iterations, packets that need 10, packets that fail
    tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps],
                       [6.0 + 4.0 * (i % 16) / 15.0 for i in range(B)], seeds=[900 + i for i in range(B)])
    torch.cuda.synchronize()
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    ptrs = [int(c.data_ptr()) for c in caps]
    ref = None
    for it in range(passes):
        rx.enqueue_device(ptrs, [nsamp] * B, "cu8")
        rx.collect()
        snap = []
        for ch in range(B):
            p = rx.packets(ch)
            snap.append((p["bytes"].copy(), p["iter"].copy(), p["crc_ok"].copy(), p["start"].copy()))
        soft = [rx.soft(ch).copy() for ch in range(0, B, 97)]
        if ref is None:
            ref, ref_soft = snap, soft
            npk = sum(len(s[1]) for s in snap)
            iters = np.concatenate([s[1] for s in snap])
            assert npk > 5 * B and len(set(iters.tolist())) >= 5 and 0 < int(sum(s[2].sum() for s in snap)) < npk      # easy, hard and failing packets are all there
            continue
        for ch in range(B):
            for a, b in zip(ref[ch], snap[ch]):
                assert a.shape == b.shape and (a == b).all(), f"pass {it}, capture {ch}: packets differ from the first pass"
        for a, b in zip(ref_soft, soft):
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"pass {it}: soft decisions differ"
    rx.close()


def test_ten_million_packets_decode_the_same_every_pass():
    """the 12-pass test above sees 3*10^5 packets -- the deviation round 4 met came once in ~10^7.  Here one batch of 3584 captures x 2 s (244 002 packets) goes
    through the chain 45 times = 1.1*10^7 packets; every pass must give the digest of the first (wenet_rx_result_digest: every packet's bytes, CRC flag, iteration count and
    position), the same as an undisturbed second handle, and the decoder's agreement guard must not have had to decode anything again (the cause round 5 found is fixed in the
    kernel: with it in place the guard counted ~20 repeats per pass in the most exposed build, tools/experiments/README.md)."""
    import torch
    from wenet_amd import siggen
    from wenet_amd.rx import RxBatch
    from wenet_amd.tx import Tx

    cfg = siggen.config_v2()
    B, secs, passes = 3584, 2.0, 45
    dev = torch.device("cuda:0")
    nsamp = int(secs * cfg.Fs)
    nsym = nsamp // (cfg.Fs // cfg.Rs)
    tx = Tx.from_config(cfg)
    spp = tx.symbols_per_packet
    nfr = nsym // spp + 1
    g = torch.Generator(device=dev)
    g.manual_seed(2001)
    payloads = torch.randint(0, 256, (B * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
    symbols = torch.empty(B * nfr * spp, dtype=torch.uint8, device=dev)
    tx.frame_packets_device(payloads.data_ptr(), B * nfr, symbols.data_ptr())
    caps = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(B)]
    tx.modulate_device([symbols.data_ptr() + i * nfr * spp for i in range(B)], [nsym] * B, [c.data_ptr() for c in caps], [8.0] * B, seeds=[7000 + i for i in range(B)])
    torch.cuda.synchronize()
    rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
    ptrs = [int(c.data_ptr()) for c in caps]
    first = None
    total = 0
    for it in range(passes):
        rx.enqueue_device(ptrs, [nsamp] * B, "cu8")
        rx.collect()
        d = rx.result_digest()
        total += d[1]
        if first is None:
            first = d
            assert d[1] > 240_000 and 0.98 * d[1] < d[2] < d[1]              # (8 dB: ~99 % of the packets pass the CRC gate)
        assert d == first, f"pass {it}: digest / packets / valid {d} differ from the first pass {first}"
    assert total > 10_000_000
    assert rx.decoder_repeats() == 0
    rx.close()
