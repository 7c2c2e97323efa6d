"""Data-parallel learner: two ranks (gloo rehearsal on one GPU; RCCL on the driver's 8-GPU node) train on
their own shards and all-reduce the gradient sum: every rank ends with the parameters of one process trained on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LINEARS = (0, 2, 5, 8, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(net):
    return np.concatenate([np.concatenate([net[i].weight.detach().cpu().numpy().ravel(), net[i].bias.detach().cpu().numpy().ravel()])
                           for i in LINEARS]).astype(np.float32)


def _batch(n, seed):
    rng = np.random.default_rng(seed)
    s = (rng.standard_normal((n, 40)) * 2).astype(np.float32)
    s[:, 12] = rng.integers(0, 4, n)
    return dict(states=s, next_states=(rng.standard_normal((n, 40)) * 2).astype(np.float32), actions=rng.integers(0, 13, n).astype(np.int64),
                rewards=(rng.standard_normal(n) * 3).astype(np.float32), dones=rng.random(n) < 0.3, row_mask=rng.random(n) < 0.6)


def _make(golden_path, table_id0):
    from pulselib_amd.environments.Poker import PokerQNetwork
    g = np.load(golden_path)
    q = PokerQNetwork(None, torch.device("cuda:0"), gamma=.95, update_freq=2, state_dim=40, learning_rate=2e-4, weight_decay=1e-5,
                      seed=31, table_id0=table_id0)
    q.network.load_state_dict({k.split("/")[-1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("s40/w0/")})
    q.target_network.load_state_dict(q.network.state_dict())
    return q


def _steps(q, lo, hi, n_total):
    for it in range(3):
        b = _batch(n_total, 500 + it)
        dev = {k: torch.from_numpy(x[lo:hi]).to("cuda:0") for k, x in b.items()}
        rep = q.train_step_native(dev["states"], dev["actions"], dev["rewards"], dev["next_states"], dev["dones"], dev["row_mask"],
                                  step_counter=70 + it)
    torch.cuda.synchronize()
    return _flat(q.network), _flat(q.target_network), rep.cpu().numpy(), q.native_steps()


def _worker(rank, world, port, golden_path, n_total, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        half = n_total // world
        q = _make(golden_path, table_id0=rank * half)
        out[rank] = _steps(q, rank * half, (rank + 1) * half, n_total)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_whole_batch(golden_dir):
    import torch.multiprocessing as mp
    n_total, world = 6000, 2
    path = str(golden_dir / "qnetwork.npz")
    want_p, want_t, want_rep, want_steps = _steps(_make(path, 0), 0, n_total, n_total)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), path, n_total, out), nprocs=world, join=True)
    for r in range(world):
        p, t, rep, steps = out[r]
        np.testing.assert_allclose(p, want_p, rtol=0, atol=1e-5, err_msg=f"rank {r} parameters")
        np.testing.assert_allclose(t, want_t, rtol=0, atol=1e-5, err_msg=f"rank {r} target")
        assert steps == want_steps == 3 and rep[0] == want_rep[0]              # global row count in every rank's report
        assert abs(rep[1] - want_rep[1]) < 1e-4 * max(1.0, want_rep[1])
    np.testing.assert_array_equal(out[0][0], out[1][0])                        # the ranks stay bit-identical to each other
