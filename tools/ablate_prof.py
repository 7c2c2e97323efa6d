"""Run each phase-ablated variant of the fused step on one mid-episode snapshot, many times, WITHOUT host
timing: meant to run under `rocprofv3 --kernel-trace --stats`, which reports the true kernel duration
per template instantiation (the mask is part of the kernel name)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from pulselib_amd import _native  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = 65536
dev = torch.device("cuda:0")
env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
               K=100, alpha=50, seed=1)
native, q_seat, rot = bench.native_types_for_episode(0)
actions = torch.zeros(N, dtype=torch.long, device=dev)
packed = 0
for i, t in enumerate(native):
    packed |= (t & 15) << (4 * i)
names = ["pots", "stages", "deck_positions", "idx", "highest", "agg", "acted", "last_raise_size", "prev_stacks", "prev_invested",
         "is_done", "equity_dirty", "stacks", "current_round_bet", "total_invested", "status", "board", "equities", "obs"]
PH = _native
masks = [PH.PH_STEP, PH.PH_STEP & ~PH.PH_EQUITY, PH.PH_STEP & ~PH.PH_SHOWDOWN, PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN),
         PH.PH_STEP & ~PH.PH_REWARD, PH.PH_STEP & ~PH.PH_OBS, PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN | PH.PH_REWARD),
         PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN | PH.PH_REWARD | PH.PH_OBS), PH.PH_CAPTURE]
lib = _native.lib()
env.reset(options={"active_players": 8})
env.rollout(native, actions, 20, 100)
torch.cuda.synchronize()
print("done", env.is_done.float().mean().item(), "dirty", env.equity_dirty.float().mean().item(),
      "stages", torch.bincount(env.stages, minlength=6).tolist())
snap = {n: getattr(env, n).clone() for n in names}
rew = torch.zeros(N, device=dev)
v = env._view(inplace=True)
stream = torch.cuda.current_stream().cuda_stream
for mask in masks:
    for rep in range(30):
        for n in names:
            getattr(env, n).copy_(snap[n])
        _native.check(lib.pulse_poker_ablate(C.byref(v), mask, actions.data_ptr(), rew.data_ptr(), packed, 1000 + rep, stream))
torch.cuda.synchronize()
