"""Soak run (diagnostic): many episodes of the fused trainer loop and of the env-only paired-launch loop on one GPU; fails on any
native error (a paired launch that gives up on its verdict, a hung meeting would hit the caller's timeout).
    python tools/soak.py [trainer_episodes] [env_seconds]"""
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pulselib_amd.environments.Poker import PokerGPU, PokerQNetwork, load_gpu_agents  # noqa: E402
from pulselib_amd.environments.Poker.utils import PokerAgentType  # noqa: E402
from pulselib_amd.scripts.trainGPU import train_agent_fused  # noqa: E402

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
env_seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
N = 65536
agents, types = load_gpu_agents(dev, 9, (["tight_aggressive", "heuristic_hands", "loose_passive", "random", "small_ball"] * 2)[:9], 100, 13)
q = PokerQNetwork(None, dev, gamma=.95, update_freq=20, state_dim=40, action_dim=13, learning_rate=1e-4, weight_decay=1e-5, seed=3)
agents.insert(0, q); types.insert(0, PokerAgentType.QLEARNING)
env = PokerGPU(device=dev, agents=agents, n_players=10, max_players=10, n_games=N, seed=5)
t0 = time.time()
done = 0
while done < episodes:
    out = train_agent_fused(env, agents, types, episodes=100, n_games=N, device=dev, max_episode_steps=40, reduce_stats=False)
    done += 100
    print(f"[soak] trainer: {done} episodes, {out['total_steps']} steps in the last 100, optimizer steps {q.native_steps()}, {time.time() - t0:.0f} s", flush=True)
p = torch.cat([x.detach().reshape(-1) for x in q.network.parameters()])
assert bool(torch.isfinite(p).all()), "parameters went non-finite"
print("[soak] trainer ok; env-only loop ...", flush=True)
r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--inproc", "--no-cpu-baseline", "--trainer-loop", "off", "--other-envs", "off", "--census", "off",
                    "--min-timed-ms", str(env_seconds * 1000)], capture_output=True, text=True, timeout=env_seconds * 4 + 300)
print(r.stderr[-600:])
assert r.returncode == 0, "bench failed"
print("[soak] env-only ok:", [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1][:160])
