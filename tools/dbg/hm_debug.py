import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pulselib_amd.environments.Poker import PokerGPU
from pulselib_amd.utils.performance import HandMetrics, calculate_q_seat_positions
dev = torch.device("cuda:0")
N, P = 4096, 10
env = PokerGPU(device=dev, agents=[], n_players=P, max_players=P, n_games=N, seed=9)
hm = HandMetrics(dev, N)
rng = np.random.default_rng(4)
for ep, (A, q_seat) in enumerate(((6, 2), (10, 7), (2, 1), (6, 0))):
    _, info = env.reset(options={"active_players": A, "q_agent_seat": q_seat, "rotation": ep})
    hm.begin_episode(env, q_seat)
    pos = calculate_q_seat_positions(env.button, q_seat=q_seat, active_players=env.active_players)
    terminated = torch.zeros(N, dtype=torch.bool, device=dev)
    hist = {}
    for step in range(45):
        actions = torch.from_numpy(rng.choice(13, N, p=[.08, .5, .08, .03, .03, .03, .03, .03, .03, .02, .02, .02, .1])).to(dev)
        _, _, dones, _, info = env.step(actions)
        before = hm.acc.clone()
        hm.update(env, dones, terminated)
        newly = dones & ~terminated
        terminated |= dones
        if newly.any():
            st = env.stages[newly].cpu().numpy(); po = pos[newly].cpu().numpy()
            diff = (hm.acc - before).cpu().numpy().reshape(16, 5, 4)[..., 0]
            want = np.zeros((16, 5), dtype=np.int64)
            b = np.where(st >= 4, 4, np.clip(st, 0, 3))
            np.add.at(want, (po, b), 1)
            if not np.array_equal(diff, want):
                print("ep", ep, "A", env.active_players, "q", q_seat, "step", step, "button", env.button[:4].tolist(), "stages hist", np.bincount(st, minlength=6).tolist())
                print("  device cells", np.argwhere(diff).tolist(), diff[diff != 0].tolist())
                print("  want   cells", np.argwhere(want).tolist(), want[want != 0].tolist())
                break
print("done")
