"""Chunk launch (5 steps), two vs four lanes per table, for tables of 10 / 12 / 16 seats: microseconds per chunk."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
for P in (10, 12, 16):
    types = ([1, 3, 2, 2, 4, 3, 1, 4, 5, 3, 2, 3, 4, 5, 1, 2])[:P]
    for four in (False, True):
        env = PokerGPU(device=dev, agents=[], n_players=P, max_players=P, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3, K=100, alpha=50, seed=1)
        env.chunk_four_lanes = four
        actions = torch.zeros(N, dtype=torch.long, device=dev)
        tot, n, g = 0.0, 0, 0
        for ep in range(10):
            env.reset(options={"active_players": P - ep % 4, "rotation": ep})
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for c in range(7):
                env.rollout(types, actions, 5, g)
                g += 5
            b.record()
            torch.cuda.synchronize()
            if ep >= 2:
                tot += a.elapsed_time(b)
                n += 7
        print(f"{P:2d} seats, {'four' if four else 'two '} lanes per table: {tot / n * 1e3:7.2f} us per 5-step chunk at {N} tables")
