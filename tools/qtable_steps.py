"""Config 3 (2048 + tabular Q-learning, 262,144 boards, shared table): per-step cost of select / step / update along a
roll-out from reset -- all boards start from two-tile states, so the first steps hammer a few hundred table entries
(contention), later steps spread over millions (random access).  Run under rocprofv3 --kernel-trace for the kernels'
own durations; prints HIP-event times per step otherwise."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pulselib_amd.agents.qlearning import QLearningBatch  # noqa: E402
from pulselib_amd.environments.TFE.TFE import TFEBatch  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 120
env = TFEBatch(dev, B, 4, seed=0)
agent = QLearningBatch(dev, B, 4, slots=1 << 26, seed=0)
env.reset()
FUSED = "--fused" in sys.argv
for a in sys.argv:
    if a.startswith("--ablate="):          # timing-only diagnostics of csrc/qtable.hip (Deferred.ablate): the table is NOT valid afterwards
        agent._scratch.reserved0 = int(a.split("=")[1])
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(STEPS)]
for t in range(STEPS):
    if FUSED:
        ev[t][0].record(); ev[t][1].record(); ev[t][2].record()
        agent.rollout_step(env, t + 1)
        ev[t][3].record()
        continue
    ev[t][0].record()
    a = agent.get_actions(env.boards, t + 1)
    ev[t][1].record()
    nb, r, d, _, _ = env.step(a)
    ev[t][2].record()
    agent.update(nb, r, d)
    ev[t][3].record()
torch.cuda.synchronize()
rows = [[ev[t][k].elapsed_time(ev[t][k + 1]) * 1e3 for k in range(3)] for t in range(STEPS)]
for t in list(range(0, 12)) + list(range(15, STEPS, 15)):
    print(json.dumps({"step": t, "select_us": round(rows[t][0], 1), "env_step_us": round(rows[t][1], 1), "update_us": round(rows[t][2], 1),
                      "done_share": None}), flush=True)
keys = int((agent.keys != 0).sum())
print(json.dumps({"boards": B, "steps": STEPS, "distinct_states_in_table": keys, "tail_mean_us": {k: round(sum(r[i] for r in rows[60:]) / max(1, STEPS - 60), 1) for i, k in enumerate(("select", "env_step", "update"))}}))
