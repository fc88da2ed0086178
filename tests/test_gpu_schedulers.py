"""GPU: the batch schedulers at scale.  A ragged device-resident batch larger than one round of the batch demodulator -- 3 800 captures of 0.2-1.0 s -- must
give the SAME packets and soft decisions per capture whether the captures are dealt to the workgroups by length or as given, whether the launch is cut into time slices inside one
launch, into forced short slices, or not at all; and a spread of captures must equal the oracle (reference semantics per capture: src/fsk_demod.c:270-413, one process per
capture).  Everything is seeded: tools/gpu_ragged.py is the long form of this test and writes profiles/r05_ragged.txt."""
import numpy as np
import pytest

import oracle_lib as ol
from wenet_amd import siggen
from wenet_amd.rx import RxBatch

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def ragged_batch(B, seed=11, lo=0.2, hi=1.0, nbase=48):
    import torch
    from wenet_amd.tx import Tx
    cfg = siggen.config_v2()
    dev = torch.device("cuda:0")
    nsym = int(hi * cfg.Rs); nsamp = nsym * (cfg.Fs // cfg.Rs)
    tx = Tx.from_config(cfg); spp = tx.symbols_per_packet; nfr = nsym // spp + 1
    g = torch.Generator(device=dev); g.manual_seed(seed)
    pay = torch.randint(0, 256, (nbase * nfr, 256), dtype=torch.uint8, device=dev, generator=g)
    sym = torch.empty(nbase * nfr * spp, dtype=torch.uint8, device=dev)
    tx.frame_packets_device(pay.data_ptr(), nbase * nfr, sym.data_ptr())
    base = [torch.empty(2 * nsamp, dtype=torch.uint8, device=dev) for _ in range(nbase)]
    tx.modulate_device([sym.data_ptr() + i * nfr * spp for i in range(nbase)], [nsym] * nbase, [c.data_ptr() for c in base],
                       [7.0 + 3.0 * (i % 7) / 6.0 for i in range(nbase)], seeds=[4000 + seed * 100 + i for i in range(nbase)], ppm=[(-60.0, 0.0, 45.0)[i % 3] for i in range(nbase)])
    torch.cuda.synchronize()
    rng = np.random.default_rng(seed)
    ns = [int(x) for x in rng.integers(int(lo * cfg.Fs), nsamp, B)]
    ns[5] = 0; ns[17] = 300                                               # an empty capture and one shorter than a frame ride along
    which = [int(x) for x in rng.integers(0, nbase, B)]
    return cfg, base, which, ns


SCHEDULES = [("by length, time slices inside one launch", {}), ("as given", {"WENET_RX_NO_SORT": "1"}), ("no time slices", {"WENET_RX_NO_DEV_SLICES": "1"}),
             ("slices of 60 000 samples forced", {"WENET_RX_DEV_SLICE_SAMPLES": "60000"})]


def test_ragged_batch_beyond_one_round_every_schedule_the_same_and_equal_to_the_oracle(monkeypatch):
    B = 3800
    cfg, base, which, ns = ragged_batch(B)
    ptrs = [int(base[w].data_ptr()) for w in which]
    picks = sorted(set(list(range(0, B, 119)) + [5, 17, B - 1]))                     # 35 captures for the oracle
    soft_picks = list(range(0, B, 7))
    ref = None
    for tag, env in SCHEDULES:
        for k in ("WENET_RX_NO_SORT", "WENET_RX_NO_DEV_SLICES", "WENET_RX_DEV_SLICE_SAMPLES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rx = RxBatch(cfg.Fs, cfg.Rs, cfg.M, framing=cfg.mode)
        rx.enqueue_device(ptrs, ns, "cu8")
        rx.collect()
        assert rx.last_kernel() == "wenet_demod_oct_kernel"
        got = (rx.result_digest(), [rx.frames(i) for i in range(B)], [rx.soft(i).copy() for i in soft_picks])
        if ref is None:
            ref = got
            assert got[0][1] > 20 * B / 10 and got[0][2] > 0
            for i in picks:                                                            # ... and the reference's semantics, capture by capture
                raw = base[which[i]][: 2 * ns[i]].cpu().numpy()
                sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
                want = ol.oracle_deframe(sd, cfg.mode)
                assert bits_equal(rx.soft(i), sd), (tag, i)
                p = rx.packets(i)
                assert p["n"] == want["n"] and (p["bytes"] == want["bytes"]).all() and (p["iter"] == want["iter"]).all(), (tag, i)
        else:
            assert got[0] == ref[0], f"{tag}: digest / packets / valid {got[0]} against {ref[0]}"
            assert got[1] == ref[1], tag
            for a, b in zip(got[2], ref[2]):
                assert bits_equal(a, b), tag
        assert rx.decoder_repeats() == 0
        rx.close()
