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


# ---- mixed-protocol batches (BASELINE configs[3]): [P25 | DMR | NXDN48] channel groups split over the ranks ----------------------
def test_mixed_partition_covers_every_group(built):
    import ddn
    for world in (1, 2, 3, 8):
        for groups in ((1366, 1365, 1365), (5, 0, 3), (0, 0, 1), (4096 * 8 + 1, 4096 * 8, 4096 * 8 - 1), (1, 1, 1)):
            total = sum(groups)
            got = [ddn.mixed_partition(*groups, r, world) for r in range(world)]
            sizes = [sum(c for _, c in g) for g in got]
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
            for k in range(3):      # every group: the ranks' ranges are contiguous, in rank order, and cover it once
                pos = 0
                for g in got:
                    f, c = g[k]
                    if c:
                        assert f == pos
                        pos += c
                assert pos == groups[k]
            # the global index a rank owns is one contiguous block
            starts = [0, groups[0], groups[0] + groups[1]]
            for r, g in enumerate(got):
                idx = [starts[k] + f + j for k, (f, c) in enumerate(g) for j in range(c)]
                assert idx == list(range(idx[0], idx[0] + len(idx))) if idx else True
                assert ddn_shard.channel_range(r, world, total) == ((idx[0] if idx else ddn_shard.channel_range(r, world, total)[0]), len(idx))


def _mixed_worker(rank, world, port, groups, n, out_path):
    """each rank runs the CPU oracles of ITS channels of every group (what the GPU ranks do with ddn_mixed_chain), rank 0 gathers the
    per-channel symbol counts in global channel order"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ddn
    import rx4
    desc = ddn_shard.broadcast_descriptor({"groups": groups, "n": n} if rank == 0 else None)
    part = ddn.mixed_partition(*desc["groups"], rank, world)
    rows = []
    for k, (first, count) in enumerate(part):
        for c in range(first, first + count):
            rows.append(_mixed_channel(k, c, desc["n"]))
    local = torch.tensor(rows, dtype=torch.int64).reshape(-1, 2)
    full = ddn_shard.gather_channel_major(local, sum(desc["groups"]))
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


def _mixed_channel(group, c, n):
    """-> [symbols, syncs] of channel c of group 0 (P25), 1 (DMR), 2 (NXDN48) through the receive-loop oracle with the handlers in it"""
    import rx4
    import p25gen
    rng = np.random.default_rng(1000 * group + c)
    if group == 0:
        dib = p25gen.make_frames(rng, 6, 0x293, crc=True, blocks=1 + c % 3)[0]
        rx = orc.OracleP25Rx(lock_symbols=-1, use_filter=1)
        sym, _, fl = rx.run(p25gen.modulate_disc(dib, lead=200 + 17 * c, noise=200.0, seed=c)[:n])
        return [len(sym), int((fl & 2).sum())]
    proto = rx4.PROTO_DMR if group == 1 else rx4.PROTO_NXDN48
    x = rng.normal(0, 3000, n).astype(np.float32)
    w = rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=2 if group == 1 else 0, handler=1)).run(x, max_sync=64)
    return [len(w["sym"]), len(w["sync_pos"])]


def test_two_rank_mixed_batch_matches_single_process(built, tmp_path):
    groups, n = (3, 2, 2), 6000         # 7 channels: rank 0 owns P25 0-2 + DMR 0, rank 1 DMR 1 + NXDN48 0-1
    out = str(tmp_path / "mixed.npy")
    mp.spawn(_mixed_worker, args=(2, _free_port(), groups, n, out), nprocs=2, join=True)
    got = np.load(out)
    want = np.array([_mixed_channel(k, c, n) for k in range(3) for c in range(groups[k])], np.int64)
    assert np.array_equal(got, want) and got[:3, 1].min() >= 1


def test_c_node_partition_is_the_same_rule(built):
    """ddn_node_partition (include/ddn_node.h, the C driver of a node's devices) == ddn_shard.channel_range (one process per GPU)"""
    import ddn
    for n, w in ((7, 3), (32768, 8), (5, 8), (0, 2), (9, 1)):
        got = [ddn.node_partition(n, r, w) for r in range(w)]
        assert sum(c for _, c in got) == n and got[0][0] == 0
        assert all(got[r + 1][0] == got[r][0] + got[r][1] for r in range(w - 1))
        assert max(c for _, c in got) - min(c for _, c in got) <= 1 and [c for _, c in got] == sorted((c for _, c in got), reverse=True)
        assert got == [ddn_shard.channel_range(r, w, n) for r in range(w)]
