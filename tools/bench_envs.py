"""Step throughput of the other environments at BASELINE.json's sizes (configs 3 and 5, plus Blackjack), with the
achieved fraction of the HBM roofline on SURVEY.md 8d's algorithmic bytes per unit:
  2048 (262,144 boards 4x4): 141 B per board-step;  its tabular Q-learning roll-out step (select + step + update);
  Particle2D (1,048,576 particles): 69 B per particle-step;  Blackjack (1,048,576 games): ~65 B per game-step.
Times whole `step()` calls (host wrapper included) with HIP events over many back-to-back calls, and the kernels
alone from a rocprofv3 kernel trace when run under it.  Prints one JSON line per environment."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def line(name, units, seconds, bytes_per_unit, note=""):
    gbs = units * bytes_per_unit / seconds / 1e9
    print(json.dumps({"env": name, "units_per_step": units, "us_per_step": seconds * 1e6, "units_per_sec": units / seconds,
                      "algorithmic_bytes_per_unit": bytes_per_unit, "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                      "note": note}), flush=True)


def main():
    from pulselib_amd.agents.qlearning import QLearningBatch
    from pulselib_amd.environments.blackjack.blackjack import BlackJack
    from pulselib_amd.environments.Particle2D.Particle2D import Particle2D
    from pulselib_amd.environments.TFE.TFE import TFEBatch
    dev = torch.device("cuda", 0)

    B = 262144
    env = TFEBatch(dev, B, 4, seed=0)
    env.reset()
    acts = torch.randint(0, 4, (B,), device=dev)
    line("2048 step, 262,144 boards", B, timed(lambda: env.step(acts), 200), 141)
    env.reset()
    agent = QLearningBatch(dev, B, 4, slots=1 << 26, seed=0)      # 64 M slots (2.5 GB): room for every state 40 steps can visit
    state = {"t": 0}

    def rollout_step():
        state["t"] += 1
        a = agent.get_actions(env.boards, state["t"])
        nb, r, d, _, _ = env.step(a)
        agent.update(nb, r, d)
    line("2048 Q-learning roll-out step (select + step + update), shared table", B, timed(rollout_step, 30), 141,
         "bytes: the env step only; the hash-table lookups are cache traffic")

    P = 1 << 20
    p2 = Particle2D(dev, P)
    p2.reuse_outputs = True
    p2.reset(seed=0)
    act = torch.rand((P, 2), device=dev) * 2 - 1
    line("Particle2D step, 1,048,576 particles", P, timed(lambda: p2.step(act), 200), 69, "reuse_outputs=True: two persistent output sets alternate (the default returns fresh tensors like the reference)")

    G = 1 << 20
    bj = BlackJack(dev, G, seed=0)
    bj.reset()
    hit = torch.ones(G, dtype=torch.long, device=dev)
    line("Blackjack step, 1,048,576 games", G, timed(lambda: bj.step(hit), 100), 65)


if __name__ == "__main__":
    main()
