"""CPU plumbing agents: first-visit Monte Carlo (reference agents/MonteCarlo/FirstVisitMonteCarlo.py:13-31)."""
from pulselib_amd.agents import FirstVisitMonteCarlo


def test_first_visit_returns_and_running_mean():
    mc = FirstVisitMonteCarlo(gamma=0.5)
    s0, s1 = (12, 0, 10), (16, 0, 10)
    # s0 visited at t=0 and t=2: only the first visit's return counts
    mc.learn([(s0, 0, 1.0), (s1, 0, 2.0), (s0, 1, 4.0)])
    # returns backwards: g2 = 4 ; g1 = 0.5*4 + 2 = 4 ; g0 = 0.5*4 + 1 = 3
    assert mc.values[s1] == 4.0 and mc.values[s0] == 3.0
    assert mc.returns[s0] == [3.0, 1.0]
    mc.learn([(s0, 1, -1.0)])
    assert mc.values[s0] == (3.0 - 1.0) / 2 and mc.returns[s0] == [2.0, 2.0]
    assert mc.values[(0, 0, 0)] == 0.0            # defaultdict(float), as in the reference
