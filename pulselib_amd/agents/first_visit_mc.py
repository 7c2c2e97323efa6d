"""First-visit Monte-Carlo state-value estimation on the CPU (BASELINE.json config 1 is CPU plumbing).

Interface and arithmetic of the reference's agents/MonteCarlo/FirstVisitMonteCarlo.py:5-31: `learn` takes one
episode as [(state, action, reward), ...]; the discounted return G_t = r_t + gamma * G_{t+1} is accumulated from
the end of the episode, and for the FIRST occurrence of each state its G_t updates that state's running mean
(`returns[state] = [sum, count]`, `values[state] = sum / count`)."""
from __future__ import annotations

from collections import defaultdict


class FirstVisitMonteCarlo:
    def __init__(self, gamma: float):
        self.gamma = gamma
        self.values = defaultdict(float)                     # state -> mean first-visit return
        self.returns = defaultdict(lambda: [0.0, 0.0])       # state -> [sum of returns, number of episodes]

    def action(self, action_space):
        return action_space.sample()

    def learn(self, episode):
        # discounted return of every time step, computed backwards
        tail, discounted = 0, [0.0] * len(episode)
        for t in reversed(range(len(episode))):
            tail = self.gamma * tail + episode[t][2]
            discounted[t] = tail
        # the earliest time step of each state; the reference applies the updates walking backwards
        # (FirstVisitMonteCarlo.py:24-31), which only matters for the insertion order of new states
        first_seen = {}
        for t, step in enumerate(episode):
            first_seen.setdefault(step[0], t)
        for state, t in sorted(first_seen.items(), key=lambda kv: -kv[1]):
            total = self.returns[state]
            total[0] += discounted[t]
            total[1] += 1
            self.values[state] = total[0] / total[1]
