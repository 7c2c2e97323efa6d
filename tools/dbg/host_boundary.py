"""Host time of the episode boundary in the env-only bench loop (enqueue only, no syncs): how long the interpreter
takes between learning that an episode is over and having the next episode's first launch queued."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402
from pulselib_amd.stoprule import LaggedDoneCount  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3, K=100, alpha=50, seed=1)
rule = LaggedDoneCount(dev, N, 0.8, lag=1)
actions = torch.zeros(N, dtype=torch.long, device=dev)
red = bench.EpisodeStatsReducer(env, dev, 1)
acc = {"reset": 0.0, "stats": 0.0, "new_episode_rest": 0.0, "n": 0}
orig_reset = env.reset


def timed_reset(*a, **k):
    t = time.perf_counter()
    r = orig_reset(*a, **k)
    acc["reset"] += time.perf_counter() - t
    return r


env.reset = timed_reset


def on_end(loop):
    t = time.perf_counter()
    red(loop)
    acc["stats"] += time.perf_counter() - t
    acc["n"] += 1


loop = bench.EpisodeLoop(env, rule, actions, 40, on_episode_end=on_end)
orig_new = loop.new_episode


def timed_new():
    t = time.perf_counter()
    orig_new()
    acc["new_episode_rest"] += time.perf_counter() - t


loop.new_episode = timed_new
loop.run_steps(300)
torch.cuda.synchronize()
for k in acc:
    acc[k] = 0
t0 = time.perf_counter()
loop.run_steps(6000)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
n = acc["n"]
print(f"{n} episodes, wall {wall * 1e3:.1f} ms = {wall / n * 1e6:.1f} us per episode")
print(f"  stats enqueue      {acc['stats'] / n * 1e6:6.1f} us per episode")
print(f"  reset (python+C)   {acc['reset'] / n * 1e6:6.1f} us per episode")
print(f"  new_episode total  {acc['new_episode_rest'] / n * 1e6:6.1f} us per episode (incl. reset)")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
loop.run_steps(3000)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
