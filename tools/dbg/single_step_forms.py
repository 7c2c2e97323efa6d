"""One step per launch, three ways: the single-step kernel (4 lanes per table), the chunk kernel with n = 1 (2 lanes per
table, read-only rows staged in LDS), and env.step with caller-made actions -- microseconds per launch at N tables."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
native, q_seat, rot = bench.native_types_for_episode(0)
actions = torch.zeros(N, dtype=torch.long, device=dev)
for label, chunked in (("single-step kernel", False), ("chunk kernel, n = 1", True)):
    env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3, K=100, alpha=50, seed=1)
    env.chunked_rollout = chunked
    tot, n = 0.0, 0
    g = 0
    for ep in range(12):
        env.reset(options={"active_players": 10 - ep % 5, "rotation": ep})
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in range(30):
            env.rollout(native, actions, 1, g)
            g += 1
        b.record()
        torch.cuda.synchronize()
        if ep >= 2:
            tot += a.elapsed_time(b)
            n += 30
    print(f"{label:22s} {tot / n * 1e3:7.2f} us per step at {N} tables")
