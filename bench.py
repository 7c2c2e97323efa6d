#!/usr/bin/env python
"""bench.py -- env-steps/sec of the Poker batched env-step hot path on MI355X.

Workload (BASELINE.json configs[1], config/pokerGPU.yaml of the reference): 65,536 parallel tables per
GPU, 10 seats (the 9 scripted opponents of pokerGPU.yaml:5-14 + the Q seat, played by `random` in this
env-only measurement), STARTING_BBS 100, W1 .5, W2 .3, K 100, ALPHA 50; decks shuffled on device
(reference: rand().argsort() per reset, PokerGPU.py:86); `active_players` sampled 2..10 per episode
(PokerGPU.py:76-80) from a host RNG seeded 0; seat rotation per episode as scripts/Poker/trainGPU.py:58-72;
episode stop rule of trainGPU.py:27-33 (every 5th step, >80 % of tables done), evaluated without a host
sync one 5-step chunk late (--stop-rule sync gives the reference's blocking check).

One "step" = one pass of the hot path over the batch = scripted-opponent policy + PokerGPU.step, ONE fused
HIP launch (pulse_poker_policy_step).  Steps are counted like the reference: n_tables x step calls,
finished tables included (trainGPU.py:108).  Resets run inside the timed region and are not counted.

Prints ONE JSON line (rank 0).  `roofline` prices the fused step kernel: algorithmic bytes per launch
(453 B per table-step) over the kernel's mean duration from HIP event pairs recorded
on the launch stream around every 8th 5-launch chunk of the timed region (kernel boundaries included; bracketing every
chunk cost 10 % of the throughput it was measuring).  `cpu_baseline` times the oracle
(oracle/poker_oracle.c, the CPU restatement of the same policy+step) on the host cores, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

AGENTS = ["tight_aggressive", "heuristic_hands", "heuristic_hands", "loose_passive", "tight_aggressive",
          "random", "loose_passive", "small_ball", "tight_aggressive"]   # reference config/pokerGPU.yaml:5-14
BYTES_PER_TABLE_STEP = 453          # 173 + 28*P at P = 10
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CHECK_INTERVAL = 5                  # trainGPU.py:31
TERMINATION_THRESHOLD = 0.8         # trainGPU.py:76


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--tables", type=int, default=65536, help="tables per GPU (weak scaling)")
    ap.add_argument("--stop-rule", choices=["lagged", "sync"], default="lagged")
    ap.add_argument("--launcher", choices=["native", "python"], default="native",
                    help="native: 5-step chunks enqueued by pulse_poker_rollout; python: one ctypes call per step")
    ap.add_argument("--max-episode-steps", type=int, default=40,
                    help="episode cap: the reference's close-on-aggressor rule livelocks tables whose last ACTIVE seat "
                         "keeps calling against all-ins (SURVEY.md A.3), so >20 %% of tables may never finish; its "
                         "published runs average 31-35 steps per episode (results/PokerGPU/runs/run_2..8.yaml)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inproc", action="store_true", help="measure in this process (no guard child); see guarded()")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def native_types_for_episode(episode: int):
    """Seat -> PULSE_AGENT_* for this episode: Q seat = episode % 10, list rotated like get_rotated_agents."""
    from pulselib_amd.environments.Poker.utils import NATIVE_TYPE, PokerAgentType, get_rotated_agents
    names = ["qlearning"] + AGENTS
    types = [PokerAgentType(n) for n in names]
    _, rotated, q_seat, rotation = get_rotated_agents(list(range(10)), types, episode_idx=episode, q_agent_idx=0)
    native = [NATIVE_TYPE[t] for t in rotated]
    native[q_seat] = NATIVE_TYPE[PokerAgentType.RANDOM]      # env-only: the learner's seat plays `random`
    return native, q_seat, rotation


class Runner:
    """Episode loop of scripts/Poker/trainGPU.py:57-108 without the learner."""

    def __init__(self, args, rank, world, device):
        from pulselib_amd.environments.Poker import PokerGPU
        self.args, self.rank, self.world, self.device = args, rank, world, device
        self.N = args.tables
        self.env = PokerGPU(device=device, agents=[], n_players=10, max_players=10, n_games=self.N, starting_bbs=100,
                            max_bbs=1000, w1=.5, w2=.3, K=100, alpha=50, seed=20260401, table_id0=rank * self.N)
        self.actions = torch.zeros(self.N, dtype=torch.long, device=device)
        self.host_rng = random.Random(0)
        self.episode = 0
        self.global_step = 0           # Philox offset of the scripted policies
        self.steps_in_episode = 0
        from pulselib_amd.stoprule import LaggedDoneCount
        self.done_count = LaggedDoneCount(device, self.N, TERMINATION_THRESHOLD)
        self.episode_stats = torch.zeros(2, dtype=torch.float64, device=device)          # all-reduced copy (cumulative over episodes)
        self.episode_stats_local = torch.zeros(2, dtype=torch.float64, device=device)
        self.new_episode()

    def new_episode(self):
        self.native, q_seat, rotation = native_types_for_episode(self.episode)
        A = self.host_rng.randint(2, 10)                       # PokerGPU.py:77 (host RNG: no .item() sync)
        self.env.reset(options={"rotation": rotation, "active_players": int(A), "q_agent_seat": q_seat})
        self.episode += 1
        self.steps_in_episode = 0
        self.done_count.drain()

    def run_steps(self, k, time_every=0):
        """Run exactly k counted steps (episodes roll over inside)."""
        done = 0
        env = self.env
        while done < k:
            n = min(CHECK_INTERVAL, k - done)
            if self.args.launcher == "native":      # the chunk's launches and its done-count in one native call
                env.rollout(self.native, self.actions, n, self.global_step, time_every=time_every, stop_rule=self.done_count)
            else:
                for i in range(n):
                    env.policy_step(self.native, self.actions, self.global_step + i)
                self.done_count.submit(env.is_done)
            self.global_step += n
            self.steps_in_episode += n
            done += n
            # trainGPU.py:99: the check happens at idx % 5 == 0, i.e. after steps 1, 6, 11, ...; chunks of five
            # steps check after steps 5, 10, ... -- same cadence, first check four steps later.
            if self.done_count.over(blocking=self.args.stop_rule == "sync") or self.steps_in_episode >= self.args.max_episode_steps:
                self.end_episode()
        return done

    def end_episode(self):
        if self.world > 1:
            import torch.distributed as dist
            # the only cross-GPU exchange of the path: episode statistics (RCCL all-reduce over xGMI)
            if getattr(self, "_stats_work", None) is not None:
                self._stats_work.wait()              # stream-ordered for RCCL: the buffer is about to be rewritten
                self._stats_work = None
            # one launch: {sum of the last step's rewards, tables done} added into a cumulative double[2]
            env = self.env
            env._lib.pulse_poker_stats(env.is_done.data_ptr(), env._rewards[0].data_ptr(), None, self.N, None,
                                       self.episode_stats_local.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            if dist.get_backend() == "gloo":      # one-GPU rehearsal: gloo reduces host copies
                host = self.episode_stats_local.cpu()
                dist.all_reduce(host)
            else:
                self.episode_stats.copy_(self.episode_stats_local)
                self._stats_work = dist.all_reduce(self.episode_stats, async_op=True)
        self.new_episode()


def cpu_baseline(args, n_tables):
    """Oracle (CPU restatement) timed on the host cores over a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as orc
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    threads = int(os.environ.get("PULSE_CPU_THREADS", min(threads, 16)))   # the 1-GPU box's CPU share is 16 cores
    env = orc.OraclePokerEnv(n_players=10, max_players=10, n_games=n_tables, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
                             K=100, alpha=50, n_threads=threads)
    rng = np.random.default_rng(0)
    host_rng = random.Random(0)
    actions = np.zeros(n_tables, dtype=np.int64)
    # decks: the oracle takes injected decks; build one seeded set outside the timed region and reuse it
    decks = np.argsort(rng.random((n_tables, 52)), axis=1).astype(np.int32) + 1
    total_steps, elapsed, episode, gstep = 0, 0.0, 0, 0
    while elapsed < args.cpu_seconds and episode < 5000:
        native, q_seat, rotation = native_types_for_episode(episode)
        A = host_rng.randint(2, 10)
        t0 = time.perf_counter()
        env.reset(options={"rotation": rotation, "active_players": A, "q_agent_seat": q_seat, "prefixed_decks": decks})
        idx = 0
        while True:
            env.policy_step(native, 20260401, gstep, actions)
            gstep += 1
            idx += 1
            if idx % CHECK_INTERVAL == 0 and env.is_done.mean() > TERMINATION_THRESHOLD:
                break
            if idx >= args.max_episode_steps:
                break
        elapsed += time.perf_counter() - t0
        total_steps += idx * n_tables
        episode += 1
    return {"value": total_steps / elapsed, "unit": "env-steps/sec", "cores": threads, "kind": "port",
            "sample": f"{episode} episodes x {n_tables} tables, {total_steps} table-steps in {elapsed:.1f} s "
                      f"(oracle/poker_oracle.c policy+step, OpenMP over tables, stop rule as trainGPU.py:27-33, "
                      f"cap {args.max_episode_steps} steps/episode)"}


def recorded_traffic(tables):
    """HBM-side bytes per launch of the fused step kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN/step_kernel_profile.json: FETCH_SIZE / WRITE_SIZE in separate passes, corrected by the
    dword-stream calibration recorded with them).  None if no summary matches this workload."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*/step_kernel_profile.json")):
        try:
            d = json.loads(f.read_text())
            if int(d.get("tables_per_launch", -1)) == tables:
                best = float(d["traffic_bytes_per_launch"])
        except Exception:
            pass
    return best


def _log(msg):
    """Progress marks on stderr (stdout carries only the JSON line): a silent bench cannot be told from a hung one."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    import faulthandler
    # a bench that takes minutes is a bug: dump every thread's stack and exit instead of hanging the box
    faulthandler.dump_traceback_later(int(os.environ.get("PULSE_BENCH_WATCHDOG_S", "360")), exit=True)
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knob for a one-GPU box: PULSE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 with the gloo
    # backend (RCCL refuses two ranks on one GPU); the driver's real runs use one GPU per rank over RCCL
    one_device = os.environ.get("PULSE_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _log("cpu_baseline (oracle port) ...")
        cpu = cpu_baseline(args, args.tables)
        _log(f"cpu_baseline done: {cpu['value']:.3g} steps/s on {cpu['cores']} cores")

    runner = Runner(args, rank, world, device)
    lib = runner.env._lib
    if rank == 0:
        _log(f"environment ready ({args.tables} tables/GPU x {world}); warm-up {args.warmup} steps ...")
    runner.run_steps(args.warmup)
    torch.cuda.synchronize()
    if rank == 0:
        _log(f"timing {args.steps} steps ...")
    s_ms, n_t = C.c_float(0), C.c_int32(0)
    lib.pulse_rollout_timing_collect(C.byref(s_ms), C.byref(n_t))     # drop warm-up samples

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # HIP events around every 8th five-launch chunk of the timed region (>= 75 chunks sampled at the default 3,000 steps)
    ran = runner.run_steps(args.steps, time_every=8 if args.launcher == "native" else 0)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert ran == args.steps
    if rank == 0:
        _log(f"timed region done: {elapsed * 1e3:.1f} ms")

    lib.pulse_rollout_timing_collect(C.byref(s_ms), C.byref(n_t))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_tables = args.tables * world
        value = total_tables * args.steps / elapsed
        roofline = None
        if n_t.value > 0:
            kernel_s = (s_ms.value / n_t.value) * 1e-3
            achieved = BYTES_PER_TABLE_STEP * args.tables / kernel_s / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": recorded_traffic(args.tables), "kernel": "poker_step_kernel<PH_STEP, POLICY>", "kernel_us": kernel_s * 1e6,
                        "launches_timed": n_t.value, "algorithmic_bytes_per_launch": BYTES_PER_TABLE_STEP * args.tables}
        out = {
            "metric": "env-steps/sec (whole node), Poker batched tables", "value": value, "unit": "env-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"Poker {args.tables} tables/GPU x {world} GPU, 10 seats, config/pokerGPU.yaml opponents, "
                                   f"env-only (policy+step fused), device-shuffled decks, active_players 2..10",
                       "tables_per_gpu": args.tables, "n_players": 10, "stop_rule": args.stop_rule, "max_episode_steps": args.max_episode_steps, "launcher": args.launcher,
                       "episodes": runner.episode, "parallelism": f"tables sharded x{world}, no data-path collective"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def guarded():
    """Single-GPU default: run the measurement in a child process (this one never touches the GPU) so that a
    run that wedges -- seen twice on fresh boxes, cause unknown, never in a re-run -- is killed and repeated once
    instead of costing the round its number.  `--inproc` (or torchrun ranks, or PULSE_BENCH_INPROC=1) measures in
    this process; profilers wrap that form (tools/collect_profiles.sh)."""
    import subprocess
    limit = int(os.environ.get("PULSE_BENCH_ATTEMPT_S", "300"))
    env = dict(os.environ, PULSE_BENCH_WATCHDOG_S=str(max(30, limit - 20)))
    for attempt in (1, 2):
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:], "--inproc"],
                                 stdout=subprocess.PIPE, text=True, env=env)
        try:
            out, _ = child.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            _log(f"attempt {attempt}: no result after {limit} s, killing pid {child.pid}")
            child.kill()
            try:
                child.communicate(timeout=20)
            except subprocess.TimeoutExpired:
                pass
            continue
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if child.returncode == 0 and lines:
            print(lines[-1], flush=True)
            return 0
        _log(f"attempt {attempt}: exit code {child.returncode}")
        sys.stdout.write(out)
        if child.returncode == 2:                    # argparse: repeating will not help
            return 2
    return 1


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--inproc" in sys.argv or os.environ.get("PULSE_BENCH_INPROC") == "1":
        main()
    else:
        sys.exit(guarded())
