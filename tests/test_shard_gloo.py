"""CPU, world_size 2 over gloo: the channel-sharding plumbing the multi-GPU bench uses (block partition, descriptor
broadcast, channel-ordered gather, max-over-ranks timing).  The per-rank compute in this test is the CPU oracle —
there is no GPU here — so what is being tested is the distribution logic, not the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ddn_shard
import orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_channels, n, blk, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = ddn_shard.broadcast_descriptor({"B": n_channels, "n": n, "blk": blk} if rank == 0 else None)
    first, count = ddn_shard.channel_range(rank, world, desc["B"])
    iq = orc.synth_c4fm_cu8(first, count, desc["n"])
    local = torch.from_numpy(orc.oracle_batch_cu8(iq, desc["blk"]))
    full = ddn_shard.gather_channel_major(local, desc["B"])
    tmax = ddn_shard.reduce_max_seconds(0.1 * (rank + 1), torch.device("cpu"))
    assert abs(tmax - 0.1 * world) < 1e-9
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


def test_partition_covers_everything():
    for world in (1, 2, 3, 8):
        for B in (1, 7, 8, 4096, 32768 + 5):
            spans = [ddn_shard.channel_range(r, world, B) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_gather_matches_single_process(built, tmp_path):
    B, n, blk = 7, 3000, 1024   # odd channel count: ranks own 4 and 3 channels
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), B, n, blk, out), nprocs=2, join=True)
    got = np.load(out)
    want = orc.oracle_batch_cu8(orc.synth_c4fm_cu8(0, B, n), blk)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
