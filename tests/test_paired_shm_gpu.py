"""The combination an 8-GPU job runs and a one-GPU box can still rehearse with TWO processes: the bench's episode loop with
paired launches (two check intervals per launch, verdicts taken by the launch) AND the shared-memory
exchange of the stop rule's counts between ranks.  Two ranks with 8,192 tables each (global table ids, one device, gloo for
the rendezvous) must end every episode on the step on which ONE process with all 16,384 tables ends it -- the rule decides
on the job-wide count -- and each rank's tables must come out as its half of the single run."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N_LOCAL, EPISODES, CAP = 8192, 7, 40
NAMES = ("stacks", "status", "pots", "stages", "idx", "is_done", "board", "current_round_bet")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(n_tables, table_id0, n_global, exchange, paired=True, allow_shared=True):
    """EPISODES episodes of bench.EpisodeLoop; returns per-episode (steps, local done count) and the final state."""
    import bench
    from pulselib_amd.environments.Poker import PokerGPU
    from pulselib_amd.stoprule import LaggedDoneCount
    dev = torch.device("cuda:0")
    env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=n_tables, starting_bbs=100, max_bbs=1000, w1=.5, w2=.3,
                   K=100, alpha=50, seed=bench.SEED, table_id0=table_id0)
    env.paired_launches = paired
    rule = LaggedDoneCount(dev, n_tables, bench.TERMINATION_THRESHOLD, lag=1, n_global=n_global, exchange=exchange)
    if exchange == "shm" and allow_shared:
        # both ranks sit on cuda:0: the rule would refuse to pair (their launches wait for hosts that wait for both launches to
        # have started); two grids of 8,192 tables do fit the device together, so the test may lift that
        rule.set_option(LaggedDoneCount.OPT_ALLOW_SHARED_DEVICE_PAIRS, 1)
    got = []

    def at_end(loop):
        got.append((loop.steps_in_episode, int(loop.env.is_done.sum())))

    actions = torch.zeros(n_tables, dtype=torch.long, device=dev)
    loop = bench.EpisodeLoop(env, rule, actions, CAP, on_episode_end=at_end, active_players="sampled")
    while len(got) < EPISODES:
        loop.run_steps(13)                    # call boundaries fall inside episodes: launches of one and of two check intervals
    torch.cuda.synchronize()
    state = {k: getattr(env, k).cpu().numpy().copy() for k in NAMES}
    mode = rule.native_mode
    stats = rule.stats()
    rule.close()
    return got[:EPISODES], state, mode, stats


def _worker(rank, world, port, out, allow_shared):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = _run(N_LOCAL, rank * N_LOCAL, world * N_LOCAL, "shm", allow_shared=allow_shared)
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_paired_launches_and_shm_exchange_equal_one_process():
    import torch.multiprocessing as mp
    world = 2
    want, want_state, mode, st = _run(world * N_LOCAL, 0, world * N_LOCAL, None)
    assert mode == "local" and st["paired_launches"] > 0
    unpaired, _, _, st = _run(world * N_LOCAL, 0, world * N_LOCAL, None, paired=False)
    assert unpaired == want and st["paired_launches"] == 0                   # (and pairing itself changes nothing)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
    assert all(out[r][2] == "shm" and out[r][3]["paired_launches"] > 0 and out[r][3]["verdict_timeouts"] == 0 for r in range(world))
    steps = [[e[0] for e in out[r][0]] for r in range(world)]
    assert steps[0] == steps[1] == [e[0] for e in want], (steps, want)        # every rank ends every episode where the whole job does
    for e in range(EPISODES):
        assert out[0][0][e][1] + out[1][0][e][1] == want[e][1], f"episode {e}: done counts {out[0][0][e][1]} + {out[1][0][e][1]} != {want[e][1]}"
    for r in range(world):
        for k in NAMES:
            np.testing.assert_array_equal(out[r][1][k], want_state[k][r * N_LOCAL:(r + 1) * N_LOCAL], err_msg=f"rank {r} {k}")
    lengths = {e[0] for e in want}
    assert len(lengths) > 1 and min(lengths) < CAP, f"the rule must end some episodes before the cap (lengths {sorted(lengths)})"


def test_ranks_that_share_a_device_do_not_pair_unless_told_to():
    """The guard sits in the native rule, not in the bench: two ranks on ONE GPU (shm exchange, PCI bus ids in the segment)
    run one check interval per launch -- no launch of theirs ever waits for a host -- and still play the single process's
    episodes."""
    import torch.multiprocessing as mp
    world = 2
    want, want_state, _, _ = _run(world * N_LOCAL, 0, world * N_LOCAL, None)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, False), nprocs=world, join=True)
    for r in range(world):
        assert out[r][2] == "shm" and out[r][3]["paired_launches"] == 0 and not out[r][3]["pairs"], out[r][3]
        assert [e[0] for e in out[r][0]] == [e[0] for e in want]
        for k in NAMES:
            np.testing.assert_array_equal(out[r][1][k], want_state[k][r * N_LOCAL:(r + 1) * N_LOCAL], err_msg=f"rank {r} {k}")
