"""Step throughput of the other environments at BASELINE.json's sizes (configs 3 and 5, plus Blackjack), with the
achieved fraction of the HBM roofline on SURVEY.md 8d's algorithmic bytes per unit:
  2048 (262,144 boards 4x4): 141 B per board-step;  its tabular Q-learning roll-out step (select + step + update);
  Particle2D (1,048,576 particles): 69 B per particle-step;  Blackjack (1,048,576 games): ~65 B per game-step.
`gpu_records(device)`: times whole `step()` calls (host wrapper included) with HIP events over many back-to-back calls.
`cpu_records(seconds_each)`: the oracle's scalar restatements (oracle/envs_oracle.c; the Q-learning roll-out with the
reference's Python dict per state, QLearningNumba.py:10-37) on bounded samples -- TEST INFRASTRUCTURE timed as the CPU
baseline, never on the product path.  bench.py puts both into its line (`other_envs`); run as a script it prints one
JSON line per environment (under rocprofv3: the kernels alone from the kernel trace)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0
TFE_BOARDS, PARTICLES, BJ_GAMES = 262144, 1 << 20, 1 << 20
BYTES = {"tfe_step": 141, "tfe_qlearning_rollout_step": 141, "particle2d_step": 69, "blackjack_step": 65}


def timed(fn, reps):
    import torch
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


def record(name, workload, units, seconds, note=""):
    gbs = units * BYTES[name] / seconds / 1e9
    return {"name": name, "workload": workload, "units_per_step": units, "us_per_step": seconds * 1e6, "value": units / seconds,
            "unit": "unit-steps/sec", "algorithmic_bytes_per_unit": BYTES[name],
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None},
            "timed": "whole step() calls incl. the Python wrapper, HIP events over back-to-back calls", "note": note, "cpu_baseline": None}


def gpu_records(dev):
    import torch
    from pulselib_amd.agents.qlearning import QLearningBatch
    from pulselib_amd.environments.blackjack.blackjack import BlackJack
    from pulselib_amd.environments.Particle2D.Particle2D import Particle2D
    from pulselib_amd.environments.TFE.TFE import TFEBatch
    out = []
    B = TFE_BOARDS
    env = TFEBatch(dev, B, 4, seed=0)
    env.reset()
    acts = torch.randint(0, 4, (B,), device=dev)
    out.append(record("tfe_step", "2048, 262,144 boards 4x4 (BASELINE config 3)", B, timed(lambda: env.step(acts), 200)))
    env.reset()
    agent = QLearningBatch(dev, B, 4, slots=1 << 26, seed=0)      # 64 M slots (2.5 GB): room for every state 40 steps can visit
    state = {"t": 0}

    def rollout_step():
        state["t"] += 1
        a = agent.get_actions(env.boards, state["t"])
        nb, r, d, _, _ = env.step(a)
        agent.update(nb, r, d)
    sep = timed(rollout_step, 30)
    env.reset()
    agent2 = QLearningBatch(dev, B, 4, slots=1 << 26, seed=0)
    state["t"] = 0

    def fused_step():
        state["t"] += 1
        agent2.rollout_step(env, state["t"])
    rec = record("tfe_qlearning_rollout_step", "2048 + tabular Q-learning roll-out step, 262,144 boards, shared table of 2^26 entries", B,
                 timed(fused_step, 30), "QLearningBatch.rollout_step: select + move + update in ONE launch (the separate calls: "
                 f"{sep * 1e6:.1f} us); steps 6..35 after a reset, i.e. including the steps where thousands of boards share a state; "
                 "bytes: the env step only, the hash-table lines are extra traffic")
    for _ in range(60):
        fused_step()
    rec["steady_state_us_per_step"] = timed(fused_step, 30) * 1e6            # boards spread over distinct states
    out.append(rec)
    del agent2
    del agent, env
    torch.cuda.empty_cache()
    P = PARTICLES
    p2 = Particle2D(dev, P)
    p2.reuse_outputs = True
    p2.reset(seed=0)
    act = torch.rand((P, 2), device=dev) * 2 - 1
    out.append(record("particle2d_step", "Particle2D, 1,048,576 particles (BASELINE config 5)", P, timed(lambda: p2.step(act), 200),
                      "reuse_outputs=True: two persistent output sets alternate (the default returns fresh tensors like the reference)"))
    G = BJ_GAMES
    bj = BlackJack(dev, G, seed=0)
    bj.reset()
    hit = torch.ones(G, dtype=torch.long, device=dev)
    out.append(record("blackjack_step", "Blackjack, 1,048,576 games", G, timed(lambda: bj.step(hit), 100)))
    return out


def _cpu(value, sample):
    return {"value": value, "unit": "unit-steps/sec", "cores": 1, "kind": "port", "sample": sample}


def cpu_records(seconds_each=3.0):
    """Scalar oracle ports (one thread) on bounded samples of the same workloads."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as orc
    out = {}
    rng = np.random.default_rng(0)
    # 2048 step
    B, n = TFE_BOARDS, 4
    boards = np.zeros((B, n, n), dtype=np.int32); score = np.zeros(B, dtype=np.int64)
    rew = np.zeros(B, dtype=np.int32); done = np.zeros(B, dtype=np.uint8)
    orc.tfe_reset(boards, score, n, 0)
    acts = rng.integers(0, 4, B).astype(np.int64)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each:
        steps += 1
        orc.tfe_step(boards, score, acts, rew, done, n, 0, steps)
    dt = time.perf_counter() - t0
    out["tfe_step"] = _cpu(B * steps / dt, f"{steps} steps x {B} boards in {dt:.1f} s (oracle/envs_oracle.c: oracle_tfe_step, scalar)")
    # Q-learning roll-out step: the reference agent's Python dict + the numba helpers' restatements, per board
    Bq = 4096
    boards = np.zeros((Bq, n, n), dtype=np.int32); score = np.zeros(Bq, dtype=np.int64)
    rew = np.zeros(Bq, dtype=np.int32); done = np.zeros(Bq, dtype=np.uint8)
    orc.tfe_reset(boards, score, n, 0)
    table, lib = {}, orc.lib()
    pows = (4 ** np.arange(16)).astype(np.uint64) ** 2                     # 16^i
    def keys_of(b):
        e = np.where(b > 0, np.log2(np.maximum(b, 1)).astype(np.uint64), 0).reshape(b.shape[0], -1)
        return (e * pows).sum(axis=1)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each:
        steps += 1
        ks = keys_of(boards)
        a = np.zeros(Bq, dtype=np.int64)
        p = rng.random(Bq); r_int = rng.integers(0, 1 << 32, Bq, dtype=np.uint64)
        qs = [table.setdefault(int(k), np.zeros(4, dtype=np.float64)) for k in ks]
        for g in range(Bq):
            a[g] = lib.oracle_select_action_epsilon_greedy(qs[g].ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(float(p[g])),
                                                           C.c_uint32(int(r_int[g])))
        orc.tfe_step(boards, score, a, rew, done, n, 0, steps)
        nk = keys_of(boards)
        for g in range(Bq):
            nxt = table.setdefault(int(nk[g]), np.zeros(4, dtype=np.float64))
            lib.oracle_update_q_entry(qs[g].ctypes.data_as(C.c_void_p), int(a[g]), nxt.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1),
                                      C.c_double(float(rew[g])), C.c_double(0.99), int(done[g]))
    dt = time.perf_counter() - t0
    out["tfe_qlearning_rollout_step"] = _cpu(Bq * steps / dt, f"{steps} steps x {Bq} boards in {dt:.1f} s (Python dict per state as in QLearningNumba.py:10-37 + "
                                             f"oracle_select_action_epsilon_greedy / oracle_update_q_entry / oracle_tfe_step, scalar; {len(table)} states)")
    # Particle2D
    P = PARTICLES
    state = np.concatenate([5 * rng.standard_normal((P, 2)), np.zeros((P, 2))], axis=1).astype(np.float32)
    steps_arr = np.zeros(P, dtype=np.int32)
    act = rng.uniform(-1, 1, (P, 2)).astype(np.float32)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each:
        steps += 1
        orc.particle2d_step(state, act, steps_arr, 0.1, 200)
    dt = time.perf_counter() - t0
    out["particle2d_step"] = _cpu(P * steps / dt, f"{steps} steps x {P} particles in {dt:.1f} s (oracle_particle2d_step, scalar)")
    # Blackjack: always hit; a reset with fresh decks whenever every game has ended (not timed apart: part of the loop)
    G = BJ_GAMES // 4
    bj = orc.OracleBlackjack(G)
    decks = np.argsort(rng.random((G, 52)), axis=1).astype(np.int32)
    bj.reset(decks)
    hit = np.ones(G, dtype=np.int64)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_each:
        steps += 1
        bj.step(hit)
    dt = time.perf_counter() - t0
    out["blackjack_step"] = _cpu(G * steps / dt, f"{steps} steps x {G} games in {dt:.1f} s (oracle_blackjack_step, scalar, always hit)")
    return out


def main():
    import torch
    for rec in gpu_records(torch.device("cuda", 0)):
        print(json.dumps(rec), flush=True)
    if "--cpu" in sys.argv:
        print(json.dumps(cpu_records()), flush=True)


if __name__ == "__main__":
    main()
