"""The N > 1 episode logic on CPU (gloo, world size 2): bench.py's EpisodeLoop and the trainer's stop rule decide on
the JOB-WIDE done count at a fixed lag, so ranks whose own tables finish at different speeds still end every episode at
the same step and issue the same number of collectives (scripts/Poker/trainGPU.py:27-33,99 made multi-process).  The
device side (counting, side stream, RCCL) is replaced by stoprule.HostCounts; the decision / exchange code is the
product's (stoprule.LaggedDoneCount with exchange="host")."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pulselib_amd.stoprule import HostCounts, LaggedDoneCount

N_LOCAL = 1000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_host_counts_fixed_lag_matches_the_native_rule_contract():
    """Verdict of check point c comes right after check point c + lag was submitted; an episode boundary silences
    everything submitted before it (the same cases tests/test_poker_gpu_parity.py runs against the native handle)."""
    for lag in (0, 1, 2):
        r = LaggedDoneCount(torch.device("cpu"), 10000, 0.8, lag=lag, backend=HostCounts(lag))
        assert r.exchange == "local" and r.over() is False
        fracs = [0.0, 0.5, 0.9, 0.1, 0.85, 0.8, 0.2, 0.95]
        verdicts = []
        for f in fracs:
            r.backend.submit_count(int(f * 10000))
            verdicts.append(r.over())
        assert verdicts == [False] * lag + [f > 0.8 for f in fracs[:len(fracs) - lag]]
        r.backend.submit_count(10000)
        r.drain()
        after = []
        for _ in range(lag + 2):
            r.backend.submit_count(0)
            after.append(r.over())
        assert after == [False] * (lag + 2)


class _FakeEnv:
    """Tables finish at a per-rank rate: after s steps of an episode, min(1, rate * s) of this rank's tables are done."""

    def __init__(self, rate, log):
        self.rate, self.log, self.steps, self.episode = rate, log, 0, -1

    def reset(self, options=None):
        self.steps = 0
        self.episode += 1
        self.log.append(("reset", options["active_players"], options["rotation"], options["q_agent_seat"]))

    def rollout(self, types, actions, n, step0, timer=None, stop_rule=None):
        self.steps += n
        self.log.append(("rollout", n, step0))
        stop_rule.backend.submit_count(int(min(1.0, self.rate * self.steps) * N_LOCAL))


def _bench_worker(rank, world, port, lag, exchange, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        log, collectives = [], []
        # rank 0's tables finish three times as fast as rank 1's: a rank-local rule would end episodes at different steps
        env = _FakeEnv(rate=(0.09, 0.03)[rank], log=log)
        rule = LaggedDoneCount(torch.device("cpu"), N_LOCAL, 0.8, lag=lag, n_global=N_LOCAL * world, backend=HostCounts(lag),
                               exchange=exchange)
        assert rule.exchange == (exchange or "host")

        def on_end(loop):                     # the per-episode statistics all-reduce of bench.py's EpisodeStatsReducer
            t = torch.tensor([float(env.steps), 1.0], dtype=torch.float64)
            dist.all_reduce(t)
            collectives.append(t.tolist())

        loop = bench.EpisodeLoop(env, rule, actions=None, max_episode_steps=40, on_episode_end=on_end)
        for k in (20, 20, 35, 5, 300):        # blocks like the driver's (--steps 20) and longer ones, with odd sizes
            assert loop.run_steps(k) == k
        # what this rank alone would have decided (its own count against its own tables) at the first check of episode 0
        out[rank] = {"log": log, "collectives": collectives, "episodes": loop.episode, "exchanges": rule.exchanges,
                     "decisions": rule.decisions, "global_step": loop.global_step}
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_different_done_rates_end_every_episode_together():
    """exchange None = torch.distributed all-reduce of the count (gloo); "shm" = the native shared-memory exchange
    (pulse_shm_*), the default of the nccl backend -- real cross-process traffic in both cases."""
    for lag, exchange in ((1, None), (0, None), (2, None), (1, "shm"), (0, "shm")):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_bench_worker, args=(2, _free_port(), lag, exchange, out), nprocs=2, join=True)
        a, b = out[0], out[1]
        assert a["log"] == b["log"], "ranks issued different reset / roll-out sequences"
        assert a["collectives"] == b["collectives"] and len(a["collectives"]) == a["episodes"] - 1
        assert a["episodes"] == b["episodes"] > 5 and a["exchanges"] == b["exchanges"] > 0 and a["decisions"] == b["decisions"]
        assert a["global_step"] == b["global_step"] == 380
        # episodes end on the global count (min(1, .09 s) + min(1, .03 s)) / 2 > 0.8: first true at the check after
        # step 25 (0.875), seen `lag` checks later; rank 0 alone (0.09 s > 0.8) would have stopped at the check after
        # step 10, rank 1 alone (0.03 s > 0.8) at the one after step 30 -- rank-local rules would have diverged
        second_reset = next(i for i, x in enumerate(a["log"]) if i > 0 and x[0] == "reset")
        assert sum(x[1] for x in a["log"][1:second_reset]) == min(40, 25 + 5 * lag)


def _trainer_rule_worker(rank, world, port, out):
    """The fused trainer's cadence (scripts/trainGPU.py: submit at idx % 5 == 0, verdict one check late) on per-rank
    `terminated` tensors: both ranks break at the same idx."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rule = LaggedDoneCount(torch.device("cpu"), N_LOCAL, 0.8, lag=1, backend=HostCounts(1))       # n_global by all-reduce
        assert rule.n_global == world * N_LOCAL
        ends = []
        for episode in range(4):
            rule.drain()
            terminated = torch.zeros(N_LOCAL, dtype=torch.bool)
            idx = 0
            while True:
                terminated[:min(N_LOCAL, int((0.05 + 0.04 * rank) * N_LOCAL * (idx + 1)))] = True
                if idx % 5 == 0:
                    rule.submit(terminated)
                    if rule.over():
                        break
                idx += 1
                if idx >= 60:
                    break
            ends.append(idx)
        out[rank] = ends
    finally:
        dist.destroy_process_group()


def _shm_worker(rank, world, name, out):
    from pulselib_amd.stoprule import ShmExchange
    x = ShmExchange(name, rank, world)
    got = [x.all_sum(i, (rank + 1) * 1000 + i) for i in range(2000)]        # far more rounds than slots: they are reused safely
    x.close()
    out[rank] = got


def test_shared_memory_exchange_sums_over_three_processes():
    """pulse_shm_all_sum: every rank contributes one value per index and reads the sum; 2,000 rounds through 4 slots."""
    world, name = 3, f"/pulse_test_shm_{os.getpid()}"
    mgr = mp.Manager()
    out = mgr.dict()
    try:
        mp.spawn(_shm_worker, args=(world, name, out), nprocs=world, join=True)
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
    want = [sum((r + 1) * 1000 + i for r in range(world)) for i in range(2000)]
    assert out[0] == out[1] == out[2] == want


def test_trainer_cadence_breaks_at_the_same_step_on_both_ranks():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_rule_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == out[1] and all(0 < e < 60 for e in out[0])
    # mean rate 0.07 per step: > 0.8 from idx 11 on -> first check that sees it is idx 15, its verdict comes at idx 20
    assert out[0] == [20, 20, 20, 20]


def test_shm_segment_tells_ranks_that_share_a_device():
    """The shared-memory exchange carries the GPU each rank computes on (pulse_shm_set_device: the PCI bus id).  Ranks whose
    device is also another rank's must not use launches that wait for each other's hosts (csrc/stoprule.hip:
    stoprule_pairs_supported): device_is_private answers 0 for them, 0 for a rank that has not named its device and while
    another rank has not (bounded wait), 1 once every rank has named a device of its own."""
    import ctypes as C
    from pulselib_amd import _native
    lib = _native.lib()
    name = f"/pulse_test_devices_{os.getpid()}".encode()
    try:
        os.unlink("/dev/shm" + name.decode())
    except OSError:
        pass
    hs = []
    for r in range(3):
        h = C.c_void_p()
        _native.check(lib.pulse_shm_create(name, r, 3, C.byref(h)), "pulse_shm_create")
        hs.append(h)
    os.unlink("/dev/shm" + name.decode())
    try:
        assert lib.pulse_shm_device_is_private(hs[0], 10) == 0            # has not named its own device yet
        _native.check(lib.pulse_shm_set_device(hs[0], b"0000:05:00.0"), "pulse_shm_set_device")
        assert lib.pulse_shm_device_is_private(hs[0], 10) == 0            # ranks 1, 2 still unknown: treated as shared
        _native.check(lib.pulse_shm_set_device(hs[1], b"0000:26:00.0"), "pulse_shm_set_device")
        _native.check(lib.pulse_shm_set_device(hs[2], b"0000:05:00.0"), "pulse_shm_set_device")
        assert lib.pulse_shm_device_is_private(hs[1], 10) == 1            # nobody else on 26:00.0
        assert lib.pulse_shm_device_is_private(hs[0], 10) == 0 and lib.pulse_shm_device_is_private(hs[2], 10) == 0   # ranks 0 and 2 share 05:00.0
        _native.check(lib.pulse_shm_set_device(hs[2], b"0000:45:00.0"), "pulse_shm_set_device")
        assert [lib.pulse_shm_device_is_private(h, 10) for h in hs] == [1, 1, 1]
        # the sums still work beside the device records
        total = C.c_int64(0)
        import threading
        out = [0, 0, 0]

        def rank(r):
            t = C.c_int64(0)
            _native.check(lib.pulse_shm_all_sum(hs[r], 0, 10 ** r, C.byref(t)), "pulse_shm_all_sum")
            out[r] = t.value
        ts = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
        [t.start() for t in ts]; [t.join() for t in ts]
        assert out == [111, 111, 111]
    finally:
        for h in hs:
            lib.pulse_shm_destroy(h)


class _StatsEnv(_FakeEnv):
    """_FakeEnv + what bench.EpisodeStatsReducer touches of PokerGPU: the cumulative accumulator (here on the CPU; the reset
    "launch" adds the ended episode's sums to it) and the reward buffers."""

    def __init__(self, rate, log, rank):
        super().__init__(rate, log)
        self.rank, self._pp = rank, 0
        self._rewards = [torch.zeros(N_LOCAL), torch.zeros(N_LOCAL)]

    def new_episode_stats(self):
        return torch.zeros((4, 2), dtype=torch.float64)

    @staticmethod
    def episode_stats_totals(acc):
        return acc.sum(dim=0)

    def reset(self, options=None):
        stats = (options or {}).get("episode_stats")
        if stats is not None:                               # the ended episode's sums join the running totals (pulse_poker_reset does this)
            _, acc = stats
            acc[self.episode % 4, 0] += 100.0 * (self.rank + 1) + self.episode
            acc[self.episode % 4, 1] += float(min(1.0, self.rate * self.steps) * N_LOCAL)
        super().reset(options)


def _stats_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        env = _StatsEnv(rate=(0.09, 0.03)[rank], log=[], rank=rank)
        rule = LaggedDoneCount(torch.device("cpu"), N_LOCAL, 0.8, lag=1, n_global=N_LOCAL * world, backend=HostCounts(1))
        stats = bench.EpisodeStatsReducer(env, torch.device("cpu"), world)
        loop = bench.EpisodeLoop(env, rule, actions=None, max_episode_steps=40, on_episode_end=stats)
        assert loop.run_steps(1000) == 1000
        ended = loop.episode - 1                             # episodes whose sums have joined the totals (the running one has not)
        during = stats.collectives
        totals = stats.totals()
        out[rank] = {"ended": ended, "during": during, "collectives": stats.collectives, "totals": totals, "local": stats.local.sum(dim=0).tolist()}
    finally:
        dist.destroy_process_group()


def test_episode_statistics_are_all_reduced_every_tenth_episode_and_at_the_end():
    """bench.py's EpisodeStatsReducer at N > 1: the ranks' CUMULATIVE episode sums cross the ranks at
    the reference's reporting cadence (every 10th episode, scripts/Poker/trainGPU.py:110) and once more when the totals are
    asked for -- not at every episode -- and the totals are the sum of the ranks' own accumulators, identical on every rank."""
    import bench
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_stats_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["ended"] == b["ended"] > 25
    assert a["during"] == b["during"] == a["ended"] // bench.REPORT_EVERY          # one collective per ten ended episodes ...
    assert a["collectives"] == b["collectives"] == a["during"] + 1                 # ... and the final one
    assert a["totals"] == b["totals"]
    assert abs(a["totals"]["last_step_reward_sum"] - (a["local"][0] + b["local"][0])) < 1e-9
    assert abs(a["totals"]["tables_done_at_episode_end"] - (a["local"][1] + b["local"][1])) < 1e-9
    assert a["local"][0] != b["local"][0]                                          # (the ranks' own sums differ: the totals are a real reduction)
