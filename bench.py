#!/usr/bin/env python
"""bench.py -- env-steps/sec of the Poker batched env-step hot path on MI355X.

Workload (BASELINE.json configs[1], config/pokerGPU.yaml of the reference): 65,536 parallel tables per
GPU, 10 seats (the 9 scripted opponents of pokerGPU.yaml:5-14 + the Q seat, played by `random` in this
env-only measurement), STARTING_BBS 100, W1 .5, W2 .3, K 100, ALPHA 50; decks shuffled on device
(reference: rand().argsort() per reset, PokerGPU.py:86) with Philox keyed by (seed 20260401, global table id,
episode); `active_players` sampled 2..10 per episode (PokerGPU.py:76-80) from a host RNG seeded 0; seat rotation
per episode as scripts/Poker/trainGPU.py:58-72; episode stop rule of trainGPU.py:27-33 (every 5th step, >80 % of
ALL the job's tables done), evaluated without a host sync exactly one 5-step chunk late (--stop-rule sync gives
the reference's blocking check).

One "step" = one pass of the hot path over the batch = scripted-opponent policy + PokerGPU.step for every table.
Steps are counted like the reference: n_tables x step calls, finished tables included (trainGPU.py:108).  Resets run
inside the timed region and are not counted.  The roll-out enqueues a 5-step chunk as ONE launch
(pulse_poker_rollout: state in registers across the steps, every step's observation / reward / done / action stored).

`--steps K --warmup W`: W untimed steps, then blocks of EXACTLY K steps: barrier, synchronize, clock, K steps, synchronize,
clock -- the closing barrier comes AFTER the clock is read (an 8-rank barrier is an all-reduce + host sync, tens of
microseconds: inside a 0.2 ms block it would be a "scaling loss" made by the stopwatch); the block's time is the MAX
over the ranks.  A 20-step block is ~0.2 ms, so the block is repeated (`config.repeats`) until the blocks sum to >= 1 s
of GPU time and span >= 8 episodes; `ms_per_step` / `value` come from the MEDIAN block, the mean and the spread are in
`config.blocks`.  Two workloads are timed: `active_players` sampled 2..10 per episode as the
reference's trainer does (`value`), and forced to 10 (`active_players_10`, with its own roofline).

Launching: `python bench.py --gpus N` spawns N rank processes itself (this parent never touches the GPU);
under torchrun (WORLD_SIZE set) the process is a rank.  Rank 0 prints ONE JSON line.  `roofline` prices the chunk
kernel: algorithmic bytes per launch (453 B per table-step x tables x steps in the launch) over
its mean duration from HIP event pairs recorded on the launch stream around every launch of every 4th EPISODE of the
timed blocks (whole episodes: a launch costs 0.55-1x the mean depending on the phase of the episode it falls in).
`cpu_baseline` (N = 1 only) times the oracle (oracle/poker_oracle.c, the CPU restatement of the same policy + step +
reset + shuffle, same seeds => the same games; tests/test_poker_gpu_parity.py holds the two legs to the same episode
lengths, done counts and reward sums) in a short-lived child process of its own, so that no OpenMP pool ever
lives in a process that holds the GPU.  `trainer_loop` (N = 1 only) is the second line SURVEY.md 8d asks for: the same
environment with the learner (PokerQNetwork) acting and training every step, at 65,536 tables and at the 2,000,000 the
reference's published run used.  `other_envs` (N = 1 only): 2048 step, its tabular Q-learning roll-out step, Particle2D
and Blackjack at BASELINE.json's sizes, each against its HBM roofline and with an oracle-port CPU baseline.
`finished_tables` (N = 1 only): the share of the counted table-steps that ran on tables already finished.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

AGENTS = ["tight_aggressive", "heuristic_hands", "heuristic_hands", "loose_passive", "tight_aggressive",
          "random", "loose_passive", "small_ball", "tight_aggressive"]   # reference config/pokerGPU.yaml:5-14
BYTES_PER_TABLE_STEP = 453          # 173 + 28*P at P = 10
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CHECK_INTERVAL = 5                  # trainGPU.py:31
TERMINATION_THRESHOLD = 0.8         # trainGPU.py:76
SEED = 20260401
KERNEL_SOURCES = ("pulselib_amd/csrc/poker_step.hip", "pulselib_amd/csrc/poker_device.h")   # what the chunk kernel is compiled from


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--tables", type=int, default=65536, help="tables per GPU (weak scaling)")
    ap.add_argument("--active-players", choices=["both", "sampled", "10"], default="both",
                    help="sampled: 2..10 per episode like the reference's trainer (= `value`); 10: every seat in every hand "
                         "(the P = 10 case the 453 B/step figure prices); both: `value` from sampled, a sub-record for 10")
    ap.add_argument("--stop-rule", choices=["lagged", "sync"], default="lagged")
    ap.add_argument("--max-episode-steps", type=int, default=40,
                    help="episode cap: the reference's close-on-aggressor rule livelocks tables whose last ACTIVE seat "
                         "keeps calling against all-ins (SURVEY.md A.3), so >20 %% of tables may never finish; its "
                         "published runs average 31-35 steps per episode (results/PokerGPU/runs/run_2..8.yaml)")
    ap.add_argument("--min-timed-ms", type=float, default=2000.0, help="repeat the K-step block until the blocks sum to this")
    ap.add_argument("--min-episodes", type=int, default=8, help="... and span at least this many episodes")
    ap.add_argument("--max-repeats", type=int, default=20000)
    ap.add_argument("--per-step-launches", action="store_true", help="one launch per step instead of one per chunk (A/B)")
    ap.add_argument("--stop-exchange", choices=["auto", "host", "shm", "rccl"], default="auto",
                    help="how the ranks exchange the stop rule's counts (auto: shared memory with the nccl backend, torch.distributed with gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--trainer-loop", choices=["auto", "on", "off"], default="auto", help="auto = on (N > 1: data-parallel learner, the per-GPU size only)")
    ap.add_argument("--trainer-episodes", type=int, default=12)
    ap.add_argument("--trainer-tables-large", type=int, default=2000000, help="second trainer-loop size (the reference's published N_GAMES); 0 = skip")
    ap.add_argument("--other-envs", choices=["auto", "on", "off"], default="auto", help="2048 / Q-learning / Particle2D / Blackjack records; auto: at N = 1")
    ap.add_argument("--census", choices=["auto", "on", "off"], default="auto", help="share of table-steps on finished tables; auto: at N = 1")
    ap.add_argument("--role", choices=["launcher", "rank", "cpu"], default="launcher", help=argparse.SUPPRESS)
    ap.add_argument("--inproc", action="store_true", help="measure in this process (what a profiler wraps); same as --role rank")
    return ap.parse_args(argv)


_TYPES_CACHE = {}


def native_types_for_episode(episode: int):
    """Seat -> PULSE_AGENT_* for this episode: Q seat = episode % 10, list rotated like get_rotated_agents."""
    hit = _TYPES_CACHE.get(episode % 10)          # the rotation has period 10 (ten seats)
    if hit is not None:
        return hit
    out = _native_types_for_episode(episode % 10)
    _TYPES_CACHE[episode % 10] = out
    return out


def _native_types_for_episode(episode: int):
    from pulselib_amd.environments.Poker.utils import NATIVE_TYPE, PokerAgentType, get_rotated_agents
    names = ["qlearning"] + AGENTS
    types = [PokerAgentType(n) for n in names]
    _, rotated, q_seat, rotation = get_rotated_agents(list(range(10)), types, episode_idx=episode, q_agent_idx=0)
    native = [NATIVE_TYPE[t] for t in rotated]
    native[q_seat] = NATIVE_TYPE[PokerAgentType.RANDOM]      # env-only: the learner's seat plays `random`
    return native, q_seat, rotation


def _log(msg):
    """Progress marks on stderr (stdout carries only the JSON line): a silent bench cannot be told from a hung one."""
    print(f"[bench {time.strftime('%H:%M:%S')} pid {os.getpid()}] {msg}", file=sys.stderr, flush=True)


def sample_active_players(host_rng, mode, n_players=10):
    """`active_players` of the next episode: PokerGPU.py:77 draws randint(2, n_players + 1) on the device and syncs; here
    a host RNG (no .item() sync), or the fixed 10 of the second workload."""
    return host_rng.randint(2, n_players) if mode == "sampled" else int(mode)


# ------------------------------------------------------------------------------------------------ episode loop
# The kernel timer (HIP events on the launch stream) brackets every chunk launch of every TIME_EVERY-th EPISODE, whole
# episodes only: a launch costs 0.55x..1x of the mean depending on the phase of the episode it falls in, so brackets
# that start at arbitrary chunks and end with the episode would over-sample its cheap tail.  (An event record is a
# barrier packet, ~7 us of queue gap: with every episode bracketed the loop ran 4 % slower.)
TIME_EVERY = 4


class EpisodeLoop:
    """Episode logic of scripts/Poker/trainGPU.py:57-108 without the learner, over chunks of CHECK_INTERVAL steps.
    `env` needs reset(options) and rollout(types, actions, n, step0, timer=, stop_rule=); `rule` is a
    stoprule.LaggedDoneCount (or anything with over() / drain()); `on_episode_end` runs at every boundary (the
    per-episode statistics all-reduce).  Every rank of a job makes the same sequence of calls: the stop verdicts come
    from the job-wide count at a fixed lag, everything else is a function of (episode, host RNG seed)."""

    def __init__(self, env, rule, actions, max_episode_steps, on_episode_end=None, host_seed=0, n_players=10, active_players="sampled"):
        self.env, self.rule, self.actions = env, rule, actions
        self.max_episode_steps, self.on_episode_end = max_episode_steps, on_episode_end
        self.host_rng = random.Random(host_seed)
        self.n_players = n_players
        self.active_mode = active_players
        self.episode = 0
        self.global_step = 0           # Philox offset of the scripted policies
        self.steps_in_episode = 0
        self.calls = 0
        self.new_episode()

    def new_episode(self):
        self.native, q_seat, rotation = native_types_for_episode(self.episode)
        A = sample_active_players(self.host_rng, self.active_mode, self.n_players)
        options = {"rotation": rotation, "active_players": int(A), "q_agent_seat": q_seat}
        extra = getattr(self.on_episode_end, "reset_options", None)
        if extra is not None and self.episode > 0:          # the ended episode's statistics ride on the reset launch
            options.update(extra(self))
        self.rule.drain()                     # (before the reset is enqueued: nothing but the first launch's enqueue should follow it)
        self.env.reset(options=options)
        # host-expensive follow-ups of the episode that just ended (a collective's enqueue) run while the GPU resets
        after = getattr(self.on_episode_end, "after_reset", None)
        if after is not None and self.episode > 0:
            after(self)
        self.episode += 1
        self.steps_in_episode = 0

    def run_steps(self, k, timer=None, time_every=0):
        """Run exactly k counted steps (episodes roll over inside)."""
        done = 0
        native_loop = hasattr(self.env, "rollout_until") and getattr(self.rule, "handle", None) is not None and self.rule.exchange != "host"
        while native_loop and done < k:
            # the chunks of the episode and the rule's verdicts run in one native call (no interpreter per chunk)
            timed = timer is not None and time_every > 0 and self.episode % time_every == 0
            n, over = self.env.rollout_until(self.native, self.actions, CHECK_INTERVAL, min(k - done, self.max_episode_steps - self.steps_in_episode),
                                             self.global_step, self.rule, timer=timer if timed else None, time_every=1 if timed else 0)
            self.calls += -(-n // CHECK_INTERVAL)
            self.global_step += n
            self.steps_in_episode += n
            done += n
            if over or self.steps_in_episode >= self.max_episode_steps:
                if self.on_episode_end is not None:
                    self.on_episode_end(self)
                self.new_episode()
        while done < k:
            n = min(CHECK_INTERVAL, k - done, self.max_episode_steps - self.steps_in_episode)
            tm = timer if (timer is not None and time_every > 0 and self.episode % time_every == 0) else None
            self.env.rollout(self.native, self.actions, n, self.global_step, timer=tm, stop_rule=self.rule)
            self.calls += 1
            self.global_step += n
            self.steps_in_episode += n
            done += n
            # trainGPU.py:99: the check happens at idx % 5 == 0, i.e. after steps 1, 6, 11, ...; chunks of five
            # steps check after steps 5, 10, ... -- same cadence, first check four steps later.
            if self.rule.over() or self.steps_in_episode >= self.max_episode_steps:
                if self.on_episode_end is not None:
                    self.on_episode_end(self)
                self.new_episode()
        return done


REPORT_EVERY = 10      # scripts/Poker/trainGPU.py:110: the reference reports every 10th episode


class EpisodeStatsReducer:
    """The only cross-GPU exchange of the data path besides the stop rule's count: the episode statistics {sum of the last
    step's rewards, tables done}.  Every rank accumulates its own CUMULATIVE totals on the device (inside the reset launch);
    they are all-reduced (RCCL over xGMI, asynchronous) at the reference's reporting cadence -- every REPORT_EVERY-th
    episode -- and once at the end, not at every episode: at 65,536 tables an episode is ~270 us, and an all-reduce per
    episode would be ~3,700 collectives per second and rank, each with two cross-stream dependencies, for totals nobody reads."""

    def __init__(self, env, device, world):
        import torch
        self.env, self.device, self.world = env, device, world
        self.local = env.new_episode_stats()                                  # cumulative over episodes (256 spread accumulators)
        self.reduced = torch.zeros_like(self.local)
        self.work = None
        self.collectives = 0

    def __call__(self, loop):
        """At the boundary, BEFORE the reset: the previous episode's all-reduce must have read its buffer."""
        if self.world > 1 and self.work is not None:
            self.work.wait()                         # stream-ordered for RCCL: the buffer is about to be rewritten
            self.work = None

    def reset_options(self, loop):
        """This episode's sums join the running totals inside the reset launch (PulsePokerResetOpts.stats_*): the reset
        kernel reads every table's done flag and last reward before it clears them -- no statistics launch (it was
        4.7 us per episode, 1.5 % of the loop)."""
        env = self.env
        return {"episode_stats": (env._rewards[1 - env._pp], self.local)}     # the reward set the episode's last step wrote

    def after_reset(self, loop):
        """... and AFTER the reset was enqueued: the all-reduce (its enqueue costs the host tens of microseconds, which
        the GPU spends resetting instead of waiting for the next episode's first launch)."""
        if self.world == 1 or loop.episode % REPORT_EVERY != 0:      # (loop.episode: episodes ended so far -- the same on every rank)
            return
        self._reduce(asynchronous=True)

    def _reduce(self, asynchronous):
        import torch.distributed as dist
        if dist.get_backend() == "gloo":             # one-GPU rehearsal: gloo reduces host copies
            host = self.local.to("cpu", copy=True)    # (a copy even when the sums already live on the host: the rank's own totals stay its own)
            dist.all_reduce(host)
            self.reduced.copy_(host)
        else:
            self.reduced.copy_(self.local)           # (after the reset launch that completed the sums, in stream order)
            work = dist.all_reduce(self.reduced, async_op=True)
            if asynchronous:
                self.work = work
            else:
                work.wait()
        self.collectives += 1

    def totals(self):
        """Job-wide totals up to the last episode boundary (called by every rank at the same point of its loop)."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.world > 1:
            self._reduce(asynchronous=False)         # the episodes since the last report
        t = self.env.episode_stats_totals(self.reduced if self.world > 1 else self.local).cpu().tolist()
        return {"last_step_reward_sum": t[0], "tables_done_at_episode_end": t[1], "episode_collectives": self.collectives,
                "reduced_every_episodes": REPORT_EVERY}


# ------------------------------------------------------------------------------------------------ CPU baseline (child)
def cpu_episode_trace(n_tables, max_episodes, lag, max_episode_steps, threads, seconds=None, active_players="sampled", table_id0=0):
    """The GPU leg's episodes on the oracle (CPU restatement): same seeds, same Philox decks, same scripted-opponent draws,
    same `active_players` sequence and rotation, the stop rule of trainGPU.py:27-33 at the same fixed lag.  Returns one
    record per episode {steps, done, reward_sum, seconds}; stops after `max_episodes` or once `seconds` of CPU time are
    spent.  (tests/test_poker_gpu_parity.py compares these records with the GPU leg's.)"""
    import numpy as np
    from oracle import oracle as orc
    env = orc.OraclePokerEnv(n_players=10, max_players=10, n_games=n_tables, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
                             K=100, alpha=50, n_threads=threads)
    host_rng = random.Random(0)
    actions = np.zeros(n_tables, dtype=np.int64)
    out, elapsed, gstep = [], 0.0, 0
    for episode in range(max_episodes):
        if seconds is not None and elapsed >= seconds:
            break
        native, q_seat, rotation = native_types_for_episode(episode)
        A = sample_active_players(host_rng, active_players)
        t0 = time.perf_counter()
        decks = orc.shuffle_decks(SEED, table_id0, episode, n_tables)        # what the reset kernel draws for (seed, table, episode)
        env.reset(options={"rotation": rotation, "active_players": A, "q_agent_seat": q_seat, "prefixed_decks": decks})
        idx, verdicts = 0, []
        while True:
            env.policy_step(native, SEED, gstep, actions, table_id0=table_id0)
            gstep += 1
            idx += 1
            if idx % CHECK_INTERVAL == 0:
                verdicts.append(env.is_done.mean() > TERMINATION_THRESHOLD)
                if len(verdicts) > lag and verdicts[-1 - lag]:          # the GPU leg's fixed-lag rule
                    break
            if idx >= max_episode_steps:
                break
        dt = time.perf_counter() - t0
        elapsed += dt
        out.append({"steps": idx, "done": int(env.is_done.sum()), "reward_sum": float(env.rewards.astype(np.float64).sum()), "seconds": dt})
    return out


def cpu_baseline(args):
    """Oracle (CPU restatement) timed on the host cores over a bounded sample of the same workload."""
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    mode = "10" if args.active_players == "10" else "sampled"
    trace = cpu_episode_trace(args.tables, 5000, 0 if args.stop_rule == "sync" else 1, args.max_episode_steps, threads,
                              seconds=args.cpu_seconds, active_players=mode)
    total_steps = sum(e["steps"] for e in trace) * args.tables
    elapsed = sum(e["seconds"] for e in trace)
    out = {"value": total_steps / elapsed, "unit": "env-steps/sec", "cores": threads, "kind": "port",
           "sample": f"{len(trace)} episodes x {args.tables} tables, {total_steps} table-steps in {elapsed:.1f} s "
                     f"(oracle/poker_oracle.c shuffle+reset+policy+step, OpenMP over tables, same seeds / decks / draws / "
                     f"active_players sequence ({mode}) as the GPU leg, stop rule as trainGPU.py:27-33 one check late, "
                     f"cap {args.max_episode_steps} steps/episode)"}
    if args.other_envs != "off":
        from tools.bench_envs import cpu_records
        out["other_envs"] = cpu_records(seconds_each=2.0)
    return out


# ------------------------------------------------------------------------------------------------ trainer loop leg
def trainer_loop_leg(args, device, tables, rank=0, world=1, dist=None):
    """The second line of SURVEY.md 8d: the reference trainer's loop (scripts/Poker/trainGPU.py:57-108) -- the learner
    acting and training every step -- on the native path (scripts/trainGPU.py: train_agent_fused).
    N > 1 (one process per GPU): `tables` tables per rank with the job's global table ids, the learner data-parallel --
    pulse_qnet_train_grads on the rank's rows, ONE all-reduce of the 130 KB gradient sum + row count (RCCL over xGMI),
    pulse_qnet_train_apply: the identical AdamW step on every rank -- the stop rule on the job-wide done count, episode sums
    all-reduced.  value = tables of the whole job x steps / the slowest rank's time."""
    import torch
    from pulselib_amd.environments.Poker import PokerAgentType, PokerGPU, PokerQNetwork, load_gpu_agents
    from pulselib_amd.scripts.trainGPU import train_agent_fused
    agents, types = load_gpu_agents(device, 9, AGENTS, 100, 13)
    torch.manual_seed(SEED)                                                  # identical initial weights on every rank
    q = PokerQNetwork(None, device, gamma=.95, update_freq=20, state_dim=40, action_dim=13, learning_rate=2e-4, weight_decay=1e-5,
                      seed=SEED, table_id0=rank * tables)
    agents.insert(0, q)
    types.insert(0, PokerAgentType.QLEARNING)
    env = PokerGPU(device=device, agents=agents, n_players=10, max_players=10, n_games=tables, starting_bbs=100, max_bbs=1000,
                   w1=.5, w2=.3, K=100, alpha=50, seed=SEED, table_id0=rank * tables)
    kw = dict(max_episode_steps=args.max_episode_steps, reduce_stats=world > 1)
    episodes = args.trainer_episodes if tables <= 262144 else max(3, args.trainer_episodes // 3)
    train_agent_fused(env, agents, types, 2, tables, device, **kw)          # warm-up episodes
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = train_agent_fused(env, agents, types, episodes, tables, device, **kw)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:                                                     # the slowest rank's clock
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_steps = out["total_steps"]                                         # job-wide (train_agent_fused counts the rule's n_global)
    steps = total_steps // (tables * world)
    weights = torch.cat([p.detach().reshape(-1) for p in q.network.parameters()]).double()
    digest = [float(weights.sum()), float(weights.abs().sum())]
    if dist is not None:                                                     # every rank must hold the same network after the run
        lo = torch.tensor(digest + [-d for d in digest], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(lo, op=dist.ReduceOp.MAX)
        same = bool(lo[0] == -lo[2]) and bool(lo[1] == -lo[3])
    else:
        same = True
    del env, q, agents
    torch.cuda.empty_cache()
    return {"value": total_steps / elapsed, "unit": "env-steps/sec", "ms_per_step": elapsed / max(steps, 1) * 1e3,
            "tables": tables * world, "tables_per_gpu": tables, "n_gpus": world, "episodes": episodes, "steps": steps,
            "episode_reward_sums": out["episode_rewards"][:4], "weights_equal_on_all_ranks": same,
            "learner": "PokerQNetwork 40-128-128-64-32-13, fp32 MFMA kernels (act + row lists, train, reduce + AdamW), acting and training every step"
                       + ("" if world == 1 else "; data-parallel: gradient sum + row count all-reduced every step, identical AdamW step on every rank"),
            "counts_as": "n_games x steps incl. finished tables (trainGPU.py:108), episodes timed end to end incl. resets and the per-episode read-back",
            "reference_published": {"value": 2.5e7, "tables": 2000000, "hardware": "unnamed CUDA GPU",
                                    "source": "results/PokerGPU/runs/run_2.yaml:21,35 (BASELINE.md section 1)"}}


# ------------------------------------------------------------------------------------------------ census
def finished_tables_census(args, device, episodes=8):
    """How many of the counted table-steps ran on tables that were already finished (the reference counts them,
    trainGPU.py:108; so do we).  The first `episodes` episodes of the sampled workload replayed with one launch per
    step (bit-identical to the chunks: tests) so that the done flags can be summed before every step; the stop rule is
    the timed loop's."""
    import torch
    from pulselib_amd.environments.Poker import PokerGPU
    from pulselib_amd.stoprule import LaggedDoneCount
    N = args.tables
    env = PokerGPU(device=device, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100, max_bbs=1000,
                   w1=.5, w2=.3, K=100, alpha=50, seed=SEED)
    rule = LaggedDoneCount(device, N, TERMINATION_THRESHOLD, lag=0 if args.stop_rule == "sync" else 1)
    host_rng = random.Random(0)
    actions = torch.zeros(N, dtype=torch.long, device=device)
    before = torch.zeros((), dtype=torch.int64, device=device)
    gstep, lengths, at_end = 0, [], []
    for e in range(episodes):
        native, q_seat, rotation = native_types_for_episode(e)
        env.reset(options={"rotation": rotation, "active_players": sample_active_players(host_rng, "sampled"), "q_agent_seat": q_seat})
        rule.drain()
        idx = 0
        while True:
            for i in range(CHECK_INTERVAL):
                before += env.is_done.sum()
                env.rollout(native, actions, 1, gstep, stop_rule=rule if i == CHECK_INTERVAL - 1 else None)
                gstep += 1
                idx += 1
            if rule.over() or idx >= args.max_episode_steps:
                break
        lengths.append(idx)
        at_end.append(int(env.is_done.sum()))
    rule.close()
    total = N * sum(lengths)
    return {"share_of_counted_table_steps_on_finished_tables": int(before) / total, "episodes": episodes, "episode_steps": lengths,
            "tables_done_at_episode_end": at_end, "tables": N,
            "note": "finished tables are stepped and counted as the reference does (trainGPU.py:108: total_steps += n_games * idx); "
                    "a step on a finished table still writes its observation, reward (0) and done flag"}


# ------------------------------------------------------------------------------------------------ one rank
def _sha_of_kernel_sources():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


def recorded_traffic(tables, steps_per_launch, active_players):
    """HBM-side bytes per launch of the chunk kernel.  NOT measured in this run (PMC counters need rocprofv3): read from
    the newest committed summary (profiles/rNN/step_kernel_profile*.json: FETCH_SIZE / WRITE_SIZE in separate passes,
    corrected by the dword-stream calibration recorded with them) that matches this workload AND was collected from the
    kernel source this build was compiled from (sha256 over KERNEL_SOURCES recorded at collection time); otherwise None
    with the reason.  The summary with the closest mean steps per launch is taken (up to 2.5 away: short blocks cut some
    launches to one check interval) and quoted AS IT IS, with both launch mixes named: a launch's fabric traffic is its state
    in and out plus what the caches do not absorb of the per-step outputs, and barely moves with the steps it runs (r04:
    78.7 MB at 9.97 steps per launch, 77.5 MB at 8.82) -- scaling it by the ratio, as round 3 did, was wrong.
    Returns (bytes or None, source string)."""
    want_sha = _sha_of_kernel_sources()
    best, best_rank, why = None, None, "no profiles/r*/step_kernel_profile*.json matches this workload"
    for f in sorted((ROOT / "profiles").glob("r*/step_kernel_profile*.json")):
        try:
            d = json.loads(f.read_text())
            have = float(d.get("steps_per_launch_mean", d.get("steps_per_launch", 1)))
            if int(d.get("tables_per_launch", -1)) != tables or abs(have - steps_per_launch) > 2.5:
                continue
            if str(d.get("active_players", "sampled")) != str(active_players):
                continue
            rel = str(f.relative_to(ROOT))
            if d.get("kernel_source_sha256_16") != want_sha:
                why = f"stale: {rel} was collected from other kernel sources ({d.get('kernel_source_sha256_16')} != {want_sha}); re-run tools/collect_profiles.sh"
                continue
            rank_of = (f.parent.name, -abs(have - steps_per_launch))           # the newest round's summary with the closest launch mix
            if best_rank is not None and rank_of < best_rank:
                continue
            best_rank = rank_of
            best, why = float(d["traffic_bytes_per_launch"]), rel
            if abs(have - steps_per_launch) > 0.05:
                why += f" (collected at {have:.2f} steps per launch, this run {steps_per_launch:.2f}: not scaled)"
        except Exception:
            pass
    return (best, why) if best is not None else (None, why)


def roofline_record(args, N, sum_ms, n_launches, n_steps_timed, active_players):
    if n_launches <= 0:
        return None
    kernel_s = sum_ms * 1e-3 / n_launches
    steps_per_launch = n_steps_timed / n_launches
    alg = BYTES_PER_TABLE_STEP * N * steps_per_launch
    achieved = alg / kernel_s / 1e9
    traffic, source = recorded_traffic(N, steps_per_launch, active_players)
    rec = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": source,
           "traffic_GBps": None if traffic is None else traffic / kernel_s / 1e9,
           "traffic_frac": None if traffic is None else traffic / kernel_s / 1e9 / HBM_PEAK_GBS,
           "kernel": "poker_step_kernel<PH_STEP, POLICY, MULTI>" if not args.per_step_launches else "poker_step_kernel<PH_STEP, POLICY>",
           "kernel_us": kernel_s * 1e6, "launches_timed": n_launches, "steps_per_launch": steps_per_launch,
           "algorithmic_bytes_per_launch": alg,
           "note": "achieved / frac price ALGORITHMIC bytes (453 B per table-step, SURVEY.md 8d, x tables x steps per launch) as the "
                   "bench contract asks; a chunk keeps the state in registers between its steps, so the bytes that really cross "
                   "the fabric (traffic, from the committed rocprofv3 counter summary named in traffic_source -- a file constant, "
                   "not measured in this run) are about half of that: traffic_frac is the kernel's real HBM utilisation. "
                   "The kernel is bound by instruction issue / dependent latency, not by HBM (DESIGN.md section 6)"}
    return rec


def main_rank(args):
    import faulthandler
    # a bench that takes minutes is a bug: dump every thread's stack and exit instead of hanging the box
    faulthandler.dump_traceback_later(int(os.environ.get("PULSE_BENCH_WATCHDOG_S", "480")), exit=True)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knob for a one-GPU box: PULSE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 with the gloo
    # backend (RCCL refuses two ranks on one GPU); the driver's real runs use one GPU per rank over RCCL
    one_device = os.environ.get("PULSE_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # PULSE_BENCH_FORCE_DIST=1: initialise the process group for a world of one as well (what a one-GPU box can rehearse
    # of the RCCL leg: backend init, barrier, the MAX all-reduce)
    if world > 1 or os.environ.get("PULSE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_min_over_ranks(x: float):
        """(max, min) over the ranks in ONE all-reduce (MAX of {x, -x})."""
        if dist is None:
            return x, x
        t = torch.tensor([x, -x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo = t.tolist()
        return float(hi), float(-lo)

    def max_over_ranks(x: float) -> float:
        return max_min_over_ranks(x)[0]

    def gather_over_ranks(x: float):
        if dist is None:
            return [x]
        t = torch.zeros(world, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        t[rank] = x
        dist.all_reduce(t)
        return [float(v) for v in t.tolist()]

    from pulselib_amd.environments.Poker import PokerGPU
    from pulselib_amd.stoprule import LaggedDoneCount, RolloutTimer
    N = args.tables
    env = PokerGPU(device=device, agents=[], n_players=10, max_players=10, n_games=N, starting_bbs=100,
                   max_bbs=1000, w1=.5, w2=.3, K=100, alpha=50, seed=SEED, table_id0=rank * N)
    env.chunked_rollout = not args.per_step_launches
    if one_device and world > 2:
        # Three or more ranks on ONE device (rehearsal only): a paired launch waits for its host's verdict, which waits for
        # every rank's launch to have started -- and the launches of three processes at this size do not all fit on one GPU
        # at once (two do).  One check interval per launch never waits inside the kernel.  (One process per GPU: no such
        # coupling.)
        env.paired_launches = False
    rule = LaggedDoneCount(device, N, TERMINATION_THRESHOLD, lag=0 if args.stop_rule == "sync" else 1, n_global=N * world,
                           exchange=None if args.stop_exchange == "auto" else args.stop_exchange)
    actions = torch.zeros(N, dtype=torch.long, device=device)
    timer = RolloutTimer()

    def measure(mode):
        """Warm-up, then blocks of exactly args.steps steps: barrier -> sync -> t0 -> steps -> sync -> dt; the MAX over
        the ranks of dt is the block's time.  No collective and no barrier between the two clock reads."""
        stats = EpisodeStatsReducer(env, device, world)
        loop = EpisodeLoop(env, rule, actions, args.max_episode_steps, on_episode_end=stats, active_players=mode)
        if rank == 0:
            _log(f"active_players {mode}: warm-up {args.warmup} steps ...")
        barrier()                              # the ranks start together (a rank seconds behind would stall every other rank's launches)
        loop.run_steps(args.warmup)
        torch.cuda.synchronize()
        timer.collect()
        if rank == 0:
            _log(f"active_players {mode}: timing blocks of {args.steps} steps ...")
        blocks, fastest, own, episodes0 = [], [], [], loop.episode
        while True:
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ran = loop.run_steps(args.steps, timer=timer, time_every=TIME_EVERY)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0              # read BEFORE anything collective
            assert ran == args.steps
            hi, lo = max_min_over_ranks(dt)
            blocks.append(hi); fastest.append(lo); own.append(dt)      # (hi is identical on every rank: so is the decision below)
            enough = sum(blocks) * 1e3 >= args.min_timed_ms and loop.episode - episodes0 >= args.min_episodes
            if enough or len(blocks) >= args.max_repeats:
                break
        sum_ms, n_launches, n_steps_timed = timer.collect()
        med = statistics.median(blocks)
        total_tables = N * world
        rec = {"active_players": "sampled 2..10 per episode (host RNG seeded 0)" if mode == "sampled" else "10 in every episode",
               "value": total_tables * args.steps / med, "ms_per_step": med / args.steps * 1e3,
               "repeats": len(blocks), "episodes_timed": loop.episode - episodes0,
               "blocks": {"median_ms": med * 1e3, "mean_ms": statistics.fmean(blocks) * 1e3, "min_ms": min(blocks) * 1e3,
                          "max_ms": max(blocks) * 1e3, "sum_ms": sum(blocks) * 1e3,
                          "value_from_mean": total_tables * args.steps / statistics.fmean(blocks)},
               "episode_stats": stats.totals(),
               # a straggling host shows here: per block the slowest and the fastest rank's time, and every rank's own sum
               "rank_spread": {"slowest_over_fastest_mean": statistics.fmean(h / max(l, 1e-12) for h, l in zip(blocks, fastest)),
                               "slowest_over_fastest_max": max(h / max(l, 1e-12) for h, l in zip(blocks, fastest)),
                               "per_rank_sum_ms": [v * 1e3 for v in gather_over_ranks(sum(own))]},
               "stop_rule": rule.stats() if hasattr(rule, "stats") else None,
               "roofline": roofline_record(args, N, sum_ms, n_launches, n_steps_timed, mode) if rank == 0 else None}
        if rank == 0:
            _log(f"active_players {mode}: {len(blocks)} blocks, {sum(blocks) * 1e3:.1f} ms, {rec['value']:.4g} env-steps/s")
        if world > 1:                                        # every rank: a rehearsal's log shows that the ranks stayed in step
            _log(f"rank {rank}/{world} active_players {mode}: episodes {loop.episode}, blocks {len(blocks)}, own time {sum(own) * 1e3:.1f} ms, "
                 f"job totals {rec['episode_stats']}, stop rule {rec['stop_rule']}")
        return rec

    if rank == 0:
        _log(f"environment ready ({N} tables/GPU x {world}, stop-rule exchange: {rule.exchange})")
    modes = ["sampled", "10"] if args.active_players == "both" else [args.active_players]
    records = {m: measure(m) for m in modes}
    main = records[modes[0]]

    trainer = census = other = None
    solo = world == 1
    if args.census == "on" or (args.census == "auto" and solo):
        if rank == 0:
            _log("finished-tables census ...")
            census = finished_tables_census(args, device)
    if args.trainer_loop in ("on", "auto"):                  # auto: at every N (N > 1: the data-parallel learner, per-GPU size only)
        rule.close()
        del env
        torch.cuda.empty_cache()
        sizes = [args.tables] + ([args.trainer_tables_large] if solo and args.trainer_tables_large and args.trainer_tables_large != args.tables else [])
        trainer = []
        for tables in sizes:
            if rank == 0:
                _log(f"trainer-loop leg, {tables} tables per GPU x {world} ...")
            trainer.append(trainer_loop_leg(args, device, tables, rank, world, dist))
            if rank == 0:
                _log(f"trainer loop: {trainer[-1]['value']:.3g} env-steps/s")
            if world > 1:
                t = trainer[-1]
                _log(f"rank {rank}/{world} trainer loop: episodes {t['episodes']}, steps {t['steps']}, first episode reward sums (job-wide) "
                     f"{[round(x, 3) for x in t['episode_reward_sums']]}, weights equal on all ranks: {t['weights_equal_on_all_ranks']}")
    if (args.other_envs == "on" or (args.other_envs == "auto" and solo)) and rank == 0:
        _log("other environments ...")
        from tools.bench_envs import gpu_records
        other = gpu_records(device)

    if rank == 0:
        def workload(mode):
            return (f"Poker {N} tables/GPU x {world} GPU, 10 seats, config/pokerGPU.yaml opponents, env-only (policy+step fused, "
                    f"5-step chunks), device-shuffled decks, active_players {'sampled 2..10' if mode == 'sampled' else '10'}")
        out = {
            "metric": "env-steps/sec (whole node), Poker batched tables", "value": main["value"], "unit": "env-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload(modes[0]),
                       "tables_per_gpu": N, "n_players": 10, "stop_rule": args.stop_rule, "stop_rule_exchange": rule.exchange,
                       "max_episode_steps": args.max_episode_steps, "steps_per_launch": CHECK_INTERVAL if not args.per_step_launches else 1,
                       "repeats": main["repeats"], "episodes_timed": main["episodes_timed"], "blocks": main["blocks"],
                       "episode_stats": main["episode_stats"], "min_timed_ms": args.min_timed_ms,
                       # two check intervals per launch unless the rule fell back: a launch that waited in vain for its
                       # host's verdict (a stalled rank) is counted here and the run goes on with one check interval per launch
                       "paired_launches": bool(main["stop_rule"]["pairs"]) if main.get("stop_rule") else False,
                       "verdict_timeouts": int(main["stop_rule"]["verdict_timeouts"]) if main.get("stop_rule") else 0,
                       "rank_spread": main["rank_spread"],
                       "timed_window": "barrier, sync, t0, K steps, sync, dt (no barrier or collective between the clock reads); MAX over ranks",
                       "parallelism": f"tables sharded x{world}, no data-path collective; stop-rule count exchanged per check point, episode "
                                      f"statistics all-reduced every {REPORT_EVERY}th episode and at the end"},
            "roofline": main["roofline"], "trainer_loop": trainer, "finished_tables": census, "other_envs": other,
        }
        if len(modes) > 1:
            sub = dict(records[modes[1]])
            sub["workload"] = workload(modes[1])
            out["active_players_10"] = sub
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ launcher (parent)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _passthrough_args(argv):
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--role":
            skip = True
            continue
        if a.startswith("--role=") or a == "--inproc":
            continue
        out.append(a)
    return out


def run_cpu_child(args, argv):
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    threads = int(os.environ.get("PULSE_CPU_THREADS", min(threads, 16)))       # the 1-GPU box's CPU share is 16 cores
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    _log(f"cpu_baseline (oracle port, {threads} threads) in a child process ...")
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), *_passthrough_args(argv), "--role", "cpu"],
                           stdout=subprocess.PIPE, text=True, env=env, timeout=args.cpu_seconds * 6 + 120)
    except subprocess.TimeoutExpired:
        _log("cpu_baseline child timed out")
        return None
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        _log(f"cpu_baseline child failed (exit {p.returncode})")
        return None
    cpu = json.loads(lines[-1])
    _log(f"cpu_baseline done: {cpu['value']:.3g} steps/s on {cpu['cores']} cores")
    return cpu


def launcher(args, argv):
    """This process never touches the GPU: it runs the CPU baseline in one child, then N rank children
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each), and prints rank 0's JSON line.  A rank set that
    produces no line within the limit is killed, its stderr (with the watchdog's stack dump) is kept under
    gpurun_out/bench_wedge_<time>.log, and the measurement is repeated ONCE; the line says so (`attempts`)."""
    cpu = None
    if args.gpus == 1 and not args.no_cpu_baseline:
        cpu = run_cpu_child(args, argv)
    limit = int(os.environ.get("PULSE_BENCH_ATTEMPT_S", "420"))
    base = _passthrough_args(argv)
    for attempt in (1, 2):
        port = _free_port()
        procs, errs = [], []
        for r in range(args.gpus):
            # a clean rendezvous of our own: nothing of an enclosing torchrun worker's (TORCHELASTIC_USE_AGENT_STORE would
            # make rank 0 look for the agent's store on OUR port and wait for ever)
            env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC")) and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")}
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), PULSE_BENCH_WATCHDOG_S=str(max(30, limit - 30)))
            err = open(f"/tmp/pulse_bench_{os.getpid()}_a{attempt}_r{r}.err", "w+")
            errs.append(err)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *base, "--role", "rank"],
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err, text=True, env=env))
        out, timed_out = "", False
        try:
            out, _ = procs[0].communicate(timeout=limit)
            for p in procs[1:]:
                p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            timed_out = True
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    pass
        logs = []
        for r, err in enumerate(errs):
            err.seek(0)
            logs.append(f"==== rank {r} (exit {procs[r].returncode}) ====\n{err.read()}")
            err.close()
            os.unlink(err.name)
        sys.stderr.write("\n".join(logs) + "\n")
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        ok = not timed_out and all(p.returncode == 0 for p in procs) and lines
        if ok:
            line = json.loads(lines[-1])
            if cpu and cpu.get("other_envs") is not None:           # the child's per-environment CPU samples join their GPU records
                per_env = cpu.pop("other_envs")
                for rec in line.get("other_envs") or []:
                    rec["cpu_baseline"] = per_env.get(rec.get("name"))
            line["cpu_baseline"] = cpu
            line["attempts"] = attempt
            print(json.dumps(line), flush=True)
            return 0
        wedge = ROOT / "gpurun_out" / f"bench_wedge_{time.strftime('%Y%m%d_%H%M%S')}_a{attempt}.log"
        try:
            wedge.parent.mkdir(exist_ok=True)
            wedge.write_text(f"attempt {attempt}: timed_out={timed_out}, exits={[p.returncode for p in procs]}\n" + "\n".join(logs) + "\nstdout:\n" + out)
            _log(f"attempt {attempt} failed (timed out: {timed_out}); record kept in {wedge}")
        except OSError:
            _log(f"attempt {attempt} failed (timed out: {timed_out})")
        if any(p.returncode == 2 for p in procs):                    # argparse / usage error: repeating will not help
            return 2
    return 1


if __name__ == "__main__":
    _args = parse_args()
    if _args.role == "cpu":
        print(json.dumps(cpu_baseline(_args)), flush=True)
    elif (_args.role == "rank" or _args.inproc or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PULSE_BENCH_INPROC") == "1"
          or "TORCHELASTIC_RUN_ID" in os.environ):          # a torchrun worker IS a rank, also in a world of one
        main_rank(_args)
    else:
        sys.exit(launcher(_args, sys.argv[1:]))
