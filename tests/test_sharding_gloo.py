"""N>1 path on CPU: world_size-2 gloo processes exercise the table sharding and the episode-statistics
all-reduce (the only collective of the path).  No GPU needed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pulselib_amd.sharding import EpisodeStats, shard_tables


def test_shard_tables_partitions_exactly():
    for n_total in (0, 1, 7, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_tables(n_total, world, r) for r in range(world)]
            assert sum(n for n, _ in spans) == n_total
            pos = 0
            for n, t0 in spans:
                assert t0 == pos
                pos += n
            assert max(n for n, _ in spans) - min(n for n, _ in spans) <= 1
    assert shard_tables(1048576, 8, 3) == (131072, 393216)       # BASELINE.json config 4
    with pytest.raises(ValueError):
        shard_tables(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_local, t0 = shard_tables(1001, world, rank)
        stats = EpisodeStats(torch.device("cpu"))
        # each rank reports statistics of its own shard; the sum must be the whole job's
        done = torch.arange(t0, t0 + n_local) % 3 == 0
        stats.set(done.sum().item(), float(torch.arange(t0, t0 + n_local).sum()), -2.5 * (rank + 1))
        total = stats.all_reduce_async().wait()
        out[rank] = total.tolist()
    finally:
        dist.destroy_process_group()


def test_episode_stats_all_reduce_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    want_done = float((torch.arange(1001) % 3 == 0).sum())
    want_sum = float(torch.arange(1001).sum())
    for r in range(world):
        assert out[r] == [want_done, want_sum, -7.5]


def test_episode_stats_without_process_group_is_local():
    stats = EpisodeStats(torch.device("cpu"))
    stats.set(3, 1.5, 0.25)
    assert stats.all_reduce_async().wait().tolist() == [3.0, 1.5, 0.25]
