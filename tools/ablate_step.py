"""Price the phases of the fused step kernel: run it on the SAME mid-episode snapshot with phases
compiled out (pulse_poker_ablate) and time each variant with HIP events.  Diagnostic only."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from pulselib_amd import _native  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
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
masks = {
    "full": PH.PH_STEP,
    "-equity": PH.PH_STEP & ~PH.PH_EQUITY,
    "-showdown": PH.PH_STEP & ~PH.PH_SHOWDOWN,
    "-equity-showdown": PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN),
    "-reward": PH.PH_STEP & ~PH.PH_REWARD,
    "-obs": PH.PH_STEP & ~PH.PH_OBS,
    "-eq-sd-reward": PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN | PH.PH_REWARD),
    "-eq-sd-reward-obs": PH.PH_STEP & ~(PH.PH_EQUITY | PH.PH_SHOWDOWN | PH.PH_REWARD | PH.PH_OBS),
    "capture-only": PH.PH_CAPTURE,
}
lib = _native.lib()
for A in (10, 6):
    for warm_steps in (8, 22, 34):
        env.reset(options={"active_players": A})
        env.rollout(native, actions, warm_steps, 100)
        torch.cuda.synchronize()
        snap = {n: getattr(env, n).clone() for n in names}
        rew = torch.zeros(N, device=dev)
        v = env._view(inplace=True)
        stream = torch.cuda.current_stream().cuda_stream
        print(f"A={A} after {warm_steps} steps: done {env.is_done.float().mean().item():.2f} dirty {env.equity_dirty.float().mean().item():.3f} "
              f"stage hist {torch.bincount(env.stages, minlength=6).tolist()}")
        for name, mask in masks.items():
            ts = []
            for rep in range(12):
                for n in names:
                    getattr(env, n).copy_(snap[n])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _native.check(lib.pulse_poker_ablate(C.byref(v), mask, actions.data_ptr(), rew.data_ptr(), packed, 1000 + rep, stream))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"   {name:22s} median {ts[len(ts)//2]:7.2f} us   min {ts[0]:7.2f} us")
