"""Instruction counts of the chunk kernel: run under
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python tools/sq_probe.py [tables]
then `python tools/sq_probe.py --summarize <dir>` prints per-kernel means (per launch and per wavefront-step).
Two environments: 10 seats of 10 (3 seats per lane) and 10 seats of max 16 (4 seats per lane): the difference
prices one seat iteration."""
import collections
import csv
import glob
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    f = glob.glob(str(Path(sys.argv[2]) / "*" / "*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "poker_step_kernel" not in name and "poker_reset" not in name:
            continue
        key = name.split("(")[0].replace("void (anonymous namespace)::", "")[:80]
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        waves = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"]) if "SQ_WAVES" in d else 0
        print(k, "launches", len(next(iter(d.values()))))
        for c, v in sorted(d.items()):
            m = sum(v) / len(v)
            print(f"    {c:22s} {m:14.1f}   per wave {m / waves if waves else 0:10.1f}   per wave-step(5) {m / waves / 5 if waves else 0:9.1f}")
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
from pulselib_amd.environments.Poker import PokerGPU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536

dev = torch.device("cuda:0")
native, q_seat, rot = bench.native_types_for_episode(0)
actions = torch.zeros(N, dtype=torch.long, device=dev)
for max_players in (10, 16):
    env = PokerGPU(device=dev, agents=[], n_players=10, max_players=max_players, n_games=N, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
                   K=100, alpha=50, seed=1)
    for ep, A in enumerate((10, 8, 6, 4)):
        env.reset(options={"active_players": A, "rotation": ep})
        for c in range(7):
            env.rollout(native, actions, 5, 100 * ep + 5 * c)
    torch.cuda.synchronize()
    print("max_players", max_players, "done", env.is_done.float().mean().item())
