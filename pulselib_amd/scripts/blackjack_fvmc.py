"""BASELINE.json config 1 plumbing: batched Blackjack on the GPU env (config/blackjack.yaml: BATCH_SIZE 1000)
feeding first-visit Monte-Carlo value estimation on the CPU (config/fvmc.yaml: GAMMA 0.9).  Policy of the
measurement plan: hit while the player's sum is below 17."""
from __future__ import annotations

import torch

from ..agents import FirstVisitMonteCarlo
from ..environments.blackjack import BlackJack


def run(device, batches=5, batch_size=1000, gamma=0.9, seed=1, hit_below=17):
    env = BlackJack(device, batch_size, seed=seed)
    agent = FirstVisitMonteCarlo(gamma)
    n_episodes = 0
    for _ in range(batches):
        obs, _ = env.reset()
        episodes = [[] for _ in range(batch_size)]
        alive = torch.ones(batch_size, dtype=torch.bool)
        for _step in range(12):                                   # a hand is at most ~11 hits long
            states = obs.cpu()
            actions = (obs[:, 0] >= hit_below).long()              # 0 = hit, 1 = stand (blackjack.py:116,137)
            obs, rewards, terminated, _, _ = env.step(actions)
            a, r, term = actions.cpu(), rewards.cpu(), terminated.cpu()
            for g in torch.nonzero(alive).flatten().tolist():
                episodes[g].append((tuple(states[g].tolist()), int(a[g]), int(r[g])))
            alive &= ~term
            if not alive.any():
                break
        for ep in episodes:
            agent.learn(ep)
        n_episodes += batch_size
    return agent, n_episodes


if __name__ == "__main__":
    agent, n = run(torch.device("cuda"))
    best = sorted(agent.values.items(), key=lambda kv: -kv[1])[:5]
    print(f"{n} episodes, {len(agent.values)} states; best states {best}")
