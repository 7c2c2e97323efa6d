"""Multi-GPU layout of the env-step path: independent tables shard embarrassingly, one process per
GPU, no collective in the data path.  The only cross-GPU exchange is the all-reduce of episode
statistics (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests), a few scalars per
episode -- latency-bound, launched asynchronously.

The reference is single-device (no torch.distributed anywhere); the quantities reduced here are the
ones its trainer computes per episode: terminated fraction for the stop rule
(scripts/Poker/trainGPU.py:27-33), summed reward and Q-seat profit (trainGPU.py:96,104)."""
from __future__ import annotations

import torch


def shard_tables(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous table range of `rank`: (n_local, table_id0).  Global table ids key the Philox
    streams (decks, scripted-opponent picks), so any sharding reproduces the same per-table game."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_total, world)
    n_local = base + (1 if rank < extra else 0)
    table_id0 = rank * base + min(rank, extra)
    return n_local, table_id0


class EpisodeStats:
    """[n_done, reward_sum, q_profit_sum] accumulated on the local shard, summed over ranks on demand."""

    def __init__(self, device):
        self.local = torch.zeros(3, dtype=torch.float64, device=device)
        self._work = None

    def set(self, n_done, reward_sum, q_profit_sum=0.0):
        self.local[0] = n_done
        self.local[1] = reward_sum
        self.local[2] = q_profit_sum

    def all_reduce_async(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self._work = dist.all_reduce(self.local, op=dist.ReduceOp.SUM, group=group, async_op=True)
        return self

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.local
