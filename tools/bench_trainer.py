"""Trainer-loop env-steps/sec (SURVEY.md 8d second line, 8f.1): the Poker roll-out WITH the learner in the loop --
PokerQNetwork acting (MFMA kernel) and learning (train_step_masked, PyTorch-ROCm autograd) every step -- counted the
reference's way (n_tables x steps per episode / wall time, scripts/Poker/trainGPU.py:108,116).

    python tools/bench_trainer.py [--tables 65536] [--episodes 20] [--warmup 3] [--loop fused|reference]

`--loop reference` runs train_agent, the loop with the reference's host syncs (boolean-mask indexing, blocking stop
rule), on the same kernels for comparison.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

AGENTS = ["tight_aggressive", "heuristic_hands", "heuristic_hands", "loose_passive", "tight_aggressive",
          "random", "loose_passive", "small_ball", "tight_aggressive"]   # reference config/pokerGPU.yaml:5-14


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--episodes", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loop", choices=["fused", "reference"], default="fused")
    ap.add_argument("--learner", choices=["native", "torch"], default="native", help="fused loop only")
    ap.add_argument("--max-episode-steps", type=int, default=40)
    ap.add_argument("--fuse-act-step", action="store_true", help="act + env step as ONE launch (PokerGPU.act_policy_step)")
    args = ap.parse_args()
    from pulselib_amd.environments.Poker import PokerGPU, PokerQNetwork, load_gpu_agents
    from pulselib_amd.environments.Poker.utils import PokerAgentType
    from pulselib_amd.scripts.trainGPU import train_agent, train_agent_fused
    # one process per GPU under torchrun (tables sharded, the learner data-parallel: gradient all-reduce over RCCL);
    # PULSE_BENCH_ONE_DEVICE=1 rehearses that on one GPU with gloo
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("PULSE_BENCH_ONE_DEVICE") == "1"
    device = torch.device("cuda", 0 if one_device else local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo") if one_device else dist.init_process_group(backend="nccl", device_id=device)
    agents, types = load_gpu_agents(device, 9, AGENTS, 100, 13)
    torch.manual_seed(20260401)                   # identical initial weights on every rank
    q = PokerQNetwork(None, device, gamma=.95, update_freq=20, state_dim=40, action_dim=13, learning_rate=2e-4, weight_decay=1e-5,
                      seed=20260401, table_id0=rank * args.tables)
    agents.insert(0, q)
    types.insert(0, PokerAgentType.QLEARNING)
    env = PokerGPU(device=device, agents=agents, n_players=10, max_players=10, n_games=args.tables, starting_bbs=100, max_bbs=1000,
                   w1=.5, w2=.3, K=100, alpha=50, seed=20260401, table_id0=rank * args.tables)
    run = train_agent_fused if args.loop == "fused" else train_agent
    kw = dict(max_episode_steps=args.max_episode_steps, reduce_stats=world > 1)
    if args.loop == "fused":
        kw["learner"] = args.learner
        kw["fuse_act_step"] = args.fuse_act_step
    run(env, agents, types, args.warmup, args.tables, device, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(env, agents, types, args.episodes, args.tables, device, **kw)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    steps = out["total_steps"] // (args.tables * world)       # (train_agent_fused counts the tables of the whole job)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "trainer-loop env-steps/sec, Poker batched tables (learner acting and learning every step)",
                      "value": out["total_steps"] / elapsed, "unit": "env-steps/sec", "n_gpus": world, "loop": args.loop, "learner": args.learner if args.loop == "fused" else "torch",
                      "tables": args.tables, "episodes": args.episodes, "step_calls_counted": steps,
                      "ms_per_step": elapsed / max(steps, 1) * 1e3, "learner_calls": q.step_count,
                      "mean_episode_reward": sum(out["episode_rewards"]) / max(len(out["episode_rewards"]), 1)}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
