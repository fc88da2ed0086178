"""CPU, world_size 2 over gloo: the N>1 path -- captures sharded round-robin, each rank decodes only its own
(here with the CPU oracle standing in for the per-rank GPU chain), counters combined by all_reduce; the union
equals the single-process result and no rank touches another rank's captures."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CAPS = 5


def _decode_count(i):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from wenet_amd import siggen
    cfg = siggen.config_v2()
    raw, _ = siggen.make_capture(cfg, 2 + (i % 2), 9.0, seed=300 + i)
    sd, _ = ol.oracle_demod(raw, "cu8", cfg.Fs, cfg.Rs, cfg.M)
    d = ol.oracle_deframe(sd, cfg.mode)
    return int(d["crc_ok"].sum())


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from wenet_amd.shard import gather_counts, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(N_CAPS, rank, world)
    counts = [_decode_count(i) for i in mine]
    dist.barrier()
    total = gather_counts(counts, N_CAPS, rank, world, dist)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # the bench's max-over-ranks timing reduction
    q.put((rank, mine, total, float(t.item())))
    dist.destroy_process_group()


def test_round_robin_sharding_two_ranks():
    from wenet_amd.shard import shard_indices
    assert shard_indices(5, 0, 2) == [0, 2, 4] and shard_indices(5, 1, 2) == [1, 3]
    assert sorted(shard_indices(7, 0, 3) + shard_indices(7, 1, 3) + shard_indices(7, 2, 3)) == list(range(7))
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = [_decode_count(i) for i in range(N_CAPS)]
    for rank, mine, total, tmax in res:
        assert total == single                     # union of the shards == single-process result
        assert tmax == 2.0
    assert sorted(res[0][1] + res[1][1]) == list(range(N_CAPS))
