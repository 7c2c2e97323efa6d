"""First-visit Monte-Carlo state-value estimator: same interface and arithmetic as the reference's
agents/MonteCarlo/FirstVisitMonteCarlo.py:5-31 (CPU, pure Python -- BASELINE.json config 1 is CPU plumbing).
`learn(episode)` takes [(state, action, reward), ...]; for the first visit of each state the discounted
return from that time step is added to the running mean."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Tuple


class FirstVisitMonteCarlo:
    def __init__(self, gamma: float):
        self.values: Dict[Tuple, float] = defaultdict(float)
        self.returns: Dict[Tuple, List[float]] = defaultdict(lambda: [0.0, 0.0])   # [sum of returns, count]
        self.gamma = gamma

    def action(self, action_space):
        return action_space.sample()

    def learn(self, episode: List[tuple]):
        first_visit = {}
        for t, (state, _, _) in enumerate(episode):
            if state not in first_visit:
                first_visit[state] = t
        g = 0
        for i in range(len(episode) - 1, -1, -1):                       # FirstVisitMonteCarlo.py:24-31
            state, _, reward = episode[i]
            g = self.gamma * g + reward
            if first_visit[state] == i:
                r = self.returns[state]
                r[0] += g
                r[1] += 1
                self.values[state] = r[0] / r[1]
