"""Runs the op-list scenarios of tests/scenarios_ops.py on an environment behind a small adapter: the oracle
(numpy, CPU) or the HIP path (torch tensors on the GPU).  Data plumbing only."""
from __future__ import annotations

import numpy as np


class OracleAdapter:
    """oracle.OraclePokerEnv: numpy state, decks injected (the oracle has no RNG of its own)."""

    def __init__(self, table):
        self.table = table

    def make(self, sc):
        from oracle import oracle as orc
        self.n_games = sc["n_games"]
        return orc.OraclePokerEnv(n_players=sc["n_players"], max_players=sc["max_players"], n_games=sc["n_games"], hand_ranks_table=self.table)

    def reset(self, env, options, force_players=None):
        rng = np.random.default_rng(getattr(env, "_resets", 0) + 17)
        env._resets = getattr(env, "_resets", 0) + 1
        opts = dict(options or {})
        opts.pop("_rotation_kwarg", None)
        if force_players is not None:
            opts["active_players"] = int(force_players)
        opts["prefixed_decks"] = np.stack([rng.permutation(52) + 1 for _ in range(self.n_games)]).astype(np.int32)
        env.reset(options=opts)

    def read(self, env, name):
        if name == "active_players":
            return np.asarray(env.active_players)
        return np.asarray(getattr(env, name))

    def poke(self, env, name, index, value):
        arr = getattr(env, name)
        if index is None:
            arr[...] = value
        else:
            arr[index] = value

    def set_active(self, env, n):
        env.active_players = int(n)

    def scalars(self, env, d):
        for k, v in d.items():
            setattr(env, k, v)

    def call(self, env, method, *args):
        if method == "execute_actions":
            return env.execute_actions(np.asarray(args[0], dtype=np.int64))
        return getattr(env, method)()

    def step(self, env, actions):
        _, rew, dones, _, info = env.step(np.asarray(actions, dtype=np.int64))
        return np.array(rew, copy=True), np.array(dones, copy=True)


class HipAdapter:
    """pulselib_amd PokerGPU on the GPU (same pokes the reference's tests make on its tensors)."""

    def __init__(self, device="cuda:0"):
        import torch
        self.torch, self.dev = torch, torch.device(device)

    def make(self, sc):
        from pulselib_amd.environments.Poker import PokerGPU
        return PokerGPU(device=self.dev, agents=[], n_players=sc["n_players"], max_players=sc["max_players"], n_games=sc["n_games"])

    def reset(self, env, options, force_players=None):
        torch = self.torch
        opts = None if options is None else dict(options)
        kw = {}
        if opts is not None and "_rotation_kwarg" in opts:
            kw["rotation"] = opts.pop("_rotation_kwarg")
        if force_players is None:
            env.reset(options=opts, **kw)
            return
        orig = torch.randint
        torch.randint = lambda low, high, size, device=None: torch.tensor([force_players], device=device)   # the reference tests' patch
        try:
            env.reset(options=opts, **kw)
        finally:
            torch.randint = orig

    def read(self, env, name):
        if name == "active_players":
            return np.asarray(env.active_players)
        return getattr(env, name).detach().cpu().numpy()

    def poke(self, env, name, index, value):
        torch = self.torch
        t = getattr(env, name)
        val = torch.as_tensor(value, dtype=t.dtype, device=t.device)
        if index is None:
            t[...] = val
        else:
            t[index] = val

    def set_active(self, env, n):
        env.active_players = int(n)

    def scalars(self, env, d):
        torch = self.torch
        for k, v in d.items():          # re-assigned 0-d tensors, as tests/poker/test_poker_gpu_round_progression.py:207-210 does
            setattr(env, k, torch.tensor(v, device=self.dev, dtype=torch.float32 if k in ("w1", "w2") else torch.int32))

    def call(self, env, method, *args):
        if method == "execute_actions":
            return env.execute_actions(self.torch.tensor(args[0], dtype=self.torch.long, device=self.dev))
        return getattr(env, method)()

    def step(self, env, actions):
        _, rew, dones, _, info = env.step(self.torch.tensor(actions, dtype=self.torch.long, device=self.dev))
        return rew.detach().cpu().numpy().copy(), dones.detach().cpu().numpy().copy()


def _pick(arr, index):
    return arr if index is None else arr[index]


def run_ops(sc, ad):
    env = ad.make(sc)
    last = {}
    snap = {}
    where = sc["name"]

    def value(name):
        if name in ("rewards", "dones"):
            return last[name]
        if name == "seat_idx":                       # info["seat_idx"] is a live reference to idx (PokerGPU.py:181-186)
            return ad.read(env, "idx")
        if name == "obs_shape":
            return np.asarray(ad.read(env, "obs").shape)
        if name == "stacks_sum":
            return ad.read(env, "stacks").sum(axis=1)
        return ad.read(env, name)

    for n_op, op in enumerate(sc["ops"]):
        kind, ctx = op[0], f"{where} (op {n_op}: {op[:3]})"
        if kind == "reset":
            ad.reset(env, op[1], op[2] if len(op) > 2 else None)
        elif kind == "poke":
            ad.poke(env, op[1], op[2], op[3])
        elif kind == "active":
            ad.set_active(env, op[1])
        elif kind == "scalars":
            ad.scalars(env, op[1])
        elif kind == "call":
            ad.call(env, op[1], *op[2:])
        elif kind == "step":
            last["rewards"], last["dones"] = ad.step(env, op[1])
        elif kind == "snapshot":
            snap = {n: np.array(ad.read(env, n), copy=True) for n in op[1]}
        elif kind == "expect":
            got = np.asarray(_pick(value(op[1]), op[2])).astype(np.int64)
            assert np.array_equal(got, np.asarray(op[3]).astype(np.int64)), f"{ctx}: got {got.tolist()}, want {op[3]}"
        elif kind == "approx":
            got = np.asarray(_pick(value(op[1]), op[2]), dtype=np.float64)
            assert np.allclose(got, np.asarray(op[3], dtype=np.float64), rtol=0, atol=op[4]), f"{ctx}: got {got.tolist()}, want {op[3]}"
        elif kind == "positive":
            got = np.asarray(_pick(value(op[1]), op[2]))
            assert (got > 0).all(), f"{ctx}: got {got.tolist()}"
        elif kind == "between":
            got = np.asarray(_pick(value(op[1]), op[2]), dtype=np.float64)
            assert ((got >= op[3]) & (got <= op[4])).all(), f"{ctx}: got {got.tolist()}"
        elif kind == "same":
            got, want = _pick(value(op[1]), op[2]), _pick(snap[op[1]], op[2])
            assert np.array_equal(np.asarray(got), np.asarray(want)), f"{ctx}: changed: {np.asarray(got).tolist()} vs {np.asarray(want).tolist()}"
        elif kind == "obs_hand":
            obs, hands = ad.read(env, "obs"), ad.read(env, "hands")
            assert obs[op[1], 5:7].astype(np.int64).tolist() == hands[op[1], op[2]].astype(np.int64).tolist(), ctx
        elif kind == "decks_are_permutations":
            decks, hands, A = ad.read(env, "decks"), ad.read(env, "hands"), int(ad.read(env, "active_players"))
            for g in range(decks.shape[0]):
                assert sorted(decks[g].tolist()) == list(range(1, 53)), ctx
                dealt = hands[g, :A].reshape(-1).tolist()
                assert len(dealt) == len(set(dealt)), ctx
        else:
            raise ValueError(f"unknown op {kind}")
