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


def test_first_visit_mc_reproduces_the_reference_class(golden_dir):
    """tests/golden/fvmc.npz = the reference's FirstVisitMonteCarlo.learn (agents/MonteCarlo/FirstVisitMonteCarlo.py:13-31) run
    on 3 x 40 seeded blackjack-shaped episodes (tests/golden/make_golden.py: make_fvmc): after episodes 0, 7 and 39 our class
    holds the same states IN THE SAME DICT ORDER with bit-identical values and [sum, count] records."""
    import numpy as np
    g = np.load(golden_dir / "fvmc.npz")
    for gi in range(3):
        mc = FirstVisitMonteCarlo(gamma=float(g[f"g{gi}/gamma"]))
        steps, at = g[f"g{gi}/steps"], 0
        for ep, n in enumerate(g[f"g{gi}/episode_lengths"]):
            rows = steps[at:at + n]
            at += n
            mc.learn([((int(r[0]), int(r[1]), int(r[2])), int(r[3]), float(r[4])) for r in rows])
            if f"g{gi}/after{ep}/states" in g.files:
                keys = [tuple(int(x) for x in k) for k in g[f"g{gi}/after{ep}/states"]]
                assert list(mc.values) == keys, f"gamma {gi} episode {ep}: states / insertion order"
                assert np.array_equal(np.array([mc.values[k] for k in keys]), g[f"g{gi}/after{ep}/values"])
                assert np.array_equal(np.array([mc.returns[k] for k in keys], dtype=np.float64), g[f"g{gi}/after{ep}/returns"])
        assert at == len(steps)
