"""Where a wavefront of the fused step spends its cycles: runs the diagnostic build
(libpulse_hip_stamps.so, `make -C pulselib_amd/csrc stamps`) whose kernel stores s_memtime at phase
boundaries, and prints the mean cycles per segment over all wavefronts of a few launches.
Read the SHARES, not the length (stamps forbid overlaps the real kernel has)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pulselib_amd import _native  # noqa: E402

_native._SO = ROOT / "pulselib_amd" / "libpulse_hip_stamps.so"       # load the diagnostic twin instead
import bench  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5          # steps per launch: stamps 2..9 are those of the LAST step of the chunk
dev = torch.device("cuda:0")
lib = _native.lib()
lib.pulse_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.pulse_debug_set_stamp_buffer.restype = C.c_int
env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
               K=100, alpha=50, seed=1)
native, q_seat, rot = bench.native_types_for_episode(0)
actions = torch.zeros(N, dtype=torch.long, device=dev)
n_waves = (N * 4 + 63) // 64          # sized for four lanes per table (fewer wavefronts with two)
buf = torch.zeros((n_waves, 16), dtype=torch.int64, device=dev)
names = ["launch->start", "issue loads", "wait loads+pick actor", "policy", "equities", "execute+masks", "advance/deal", "payouts",
         "reward", "obs stores", "state stores", "drain stores"]
WARMS = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
for A, warm in ([(10, w) for w in WARMS] if WARMS else ((8, 6), (8, 20), (6, 33))):
    env.reset(options={"active_players": A})
    env.rollout(native, actions, warm, 100)
    torch.cuda.synchronize()
    acc = np.zeros(11)
    total = []
    reps = 5
    for r in range(reps):
        buf.zero_()
        lib.pulse_debug_set_stamp_buffer(buf.data_ptr())
        env.rollout(native, actions, STEPS, 1000 + 10 * r)
        torch.cuda.synchronize()
        st = buf.cpu().numpy().astype(np.int64)
        st = st[st[:, 0] != 0]
        d = np.diff(st[:, :12], axis=1)
        acc += d.mean(axis=0)
        total.append((st[:, 11].max() - st[:, 0].min(), (st[:, 11] - st[:, 0]).mean(), st[:, 0].max() - st[:, 0].min()))
    lib.pulse_debug_set_stamp_buffer(None)
    acc /= reps
    print(f"[{STEPS} steps per launch] A={A} after {warm} steps: done {env.is_done.float().mean().item():.2f}; kernel span {np.mean([t[0] for t in total]):.0f} ticks, "
          f"mean wave life {np.mean([t[1] for t in total]):.0f}, start skew {np.mean([t[2] for t in total]):.0f} (ticks = 100 MHz? see note)")
    for n, c in zip(names[1:], acc):
        print(f"   {n:24s} {c:9.0f}  {100 * c / acc.sum():5.1f} %")
