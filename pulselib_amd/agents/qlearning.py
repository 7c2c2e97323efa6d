"""Batched tabular Q-learning for the 2048 roll-out: the GPU form of the reference's
agents/TemperalDifference/QLearningNumba.py:10-37 (+ utils/numba.py:5-39), one learner state per
launch for B boards.  `get_action` / `update` keep the reference's meaning per board; the Python dict
becomes a device hash table (csrc/qtable.hip) of 64-byte entries {key, q[4]}: one memory line per state.

private_tables=True  : every board learns in its own region (B independent copies of the reference
                       agent: bit-exact against the oracle, no races) -- the parity mode;
private_tables=False : all boards share one table (what a batched learner wants).  Boards that update the same
                       entry in the same launch: one of them applies its update alone, the others' are combined
                       (the k transitions one after another, each with the mean of their targets).

`rollout_step(env)` is get_actions + env.step + update in ONE launch (same draws, same results)."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native

ENTRY_WORDS = 8          # int64 words per 64-byte entry: key, q[0..3] (as float64 bits), three spare


class QLearningBatch:
    def __init__(self, device, batch_size, board_size=4, config=None, private_tables=False, slots=None, seed=0, board_id0=0):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"pulselib_amd.QLearningBatch runs on an MI355X ('cuda' device); got '{device}'. No CPU fallback.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        config = config or {}
        self.alpha = float(config.get("ALPHA", 0.1))          # config/qlearning.yaml:6-8
        self.gamma = float(config.get("GAMMA", 0.99))
        self.epsilon = float(config.get("EPSILON", 0.1))
        self.device, self.batch_size, self.n = device, batch_size, board_size
        self.seed, self.board_id0 = int(seed), int(board_id0)
        self._lib = _native.lib()
        if private_tables:
            self.region_slots = int(slots or 1024)
            self.capacity = self.region_slots * batch_size
        else:
            self.region_slots = 0
            self.capacity = int(slots or (1 << 24))
        for x in (self.capacity if not private_tables else self.region_slots,):
            if x & (x - 1):
                raise ValueError("slots must be a power of two")
        self.entries = torch.zeros((self.capacity, ENTRY_WORDS), dtype=torch.int64, device=device)
        self.keys = self.entries[:, 0]                                      # views of the entries (strided)
        self.values = self.entries[:, 1:5].view(torch.float64)
        self.actions = torch.zeros(batch_size, dtype=torch.int64, device=device)
        self.slots = torch.full((batch_size,), -2, dtype=torch.int64, device=device)
        self._q = _native.QTable(self.entries.data_ptr(), self.capacity, self.region_slots)
        self._scratch, self._scratch_tensors, self._launches, self._carried = None, None, 0, False
        if not private_tables:                                              # shared table: the deferred-update scratch (pulse_env.h)
            n = -(-batch_size // 256) * 256                                 # the deferred list is kept in segments of 256 (pulse_env.h)
            acc = 1
            while acc < 2 * max(n, 1):
                acc *= 2
            t = dict(count=torch.zeros(64 + 2 * (n // 256), dtype=torch.int32, device=device), cells=torch.zeros(n, dtype=torch.int64, device=device),
                     targets=torch.zeros(n, dtype=torch.float64, device=device), owner=torch.zeros(n, dtype=torch.int32, device=device),
                     acc_key=torch.zeros(acc, dtype=torch.int64, device=device), acc_cnt=torch.zeros(acc, dtype=torch.int32, device=device),
                     acc_sum=torch.zeros(acc, dtype=torch.float64, device=device))
            self._scratch_tensors = t
            self._scratch = _native.QTableScratch(*(t[k].data_ptr() for k in ("count", "cells", "targets", "owner", "acc_key", "acc_cnt", "acc_sum")), n, acc)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _scratch_ref(self):
        return C.byref(self._scratch) if self._scratch is not None else None

    def get_actions(self, boards, step_counter):
        """epsilon-greedy action per board (QLearningNumba.py:24-27); remembers the looked-up states for update()."""
        boards = boards if boards.is_contiguous() else boards.contiguous()
        _native.check(self._lib.pulse_qtable_select(C.byref(self._q), boards.data_ptr(), self.batch_size, self.n, self.epsilon,
                                                    self.seed, self.board_id0, int(step_counter), self.actions.data_ptr(),
                                                    self.slots.data_ptr(), self._stream()), "pulse_qtable_select")
        self._carried = False
        return self.actions

    def update(self, next_boards, rewards, terminated):
        """Q update for the transitions (state looked up by the last get_actions, its action) -> next_boards."""
        next_boards = next_boards if next_boards.is_contiguous() else next_boards.contiguous()
        term = terminated.view(torch.uint8) if terminated.dtype == torch.bool else terminated.to(torch.uint8)
        _native.check(self._lib.pulse_qtable_update(C.byref(self._q), self._scratch_ref(), self._launches, self.slots.data_ptr(),
                                                    self.actions.data_ptr(), rewards.data_ptr(), next_boards.data_ptr(), term.data_ptr(),
                                                    self.batch_size, self.n, self.alpha, self.gamma, self._stream()), "pulse_qtable_update")
        self._launches += 1
        self._carried = False         # (the separate calls look every state up; only rollout_step carries s' over)

    def rollout_step(self, env, step_counter, carry_states=True):
        """One roll-out step of every board in ONE launch: `a = get_actions(env.boards, step_counter); nb, r, d, _, _ =
        env.step(a); update(nb, r, d)` -- same draws, same results (tests), the board in registers throughout, and the
        entry of s' found by the update serves the next call's lookup (carry_states; reset it with `forget_states()`
        whenever the boards change behind the agent's back, e.g. env.reset()).  Returns what env.step returns."""
        if env.batch_size != self.batch_size or env.n != self.n or env.device != self.device:
            raise ValueError("rollout_step: the environment's batch does not match the agent's")
        if not (carry_states and self._carried):
            self.slots.fill_(-2)
        env.step_counter += 1
        dones = env.dones.view(torch.uint8)
        _native.check(self._lib.pulse_qtable_rollout_step(
            C.byref(self._q), self._scratch_ref(), self._launches, env.boards.data_ptr(), env.total_score.data_ptr(), self.batch_size, self.n,
            self.epsilon, self.alpha, self.gamma, self.seed, int(step_counter), env.seed, env.step_counter, self.board_id0,
            self.actions.data_ptr(), env.rewards.data_ptr(), dones.data_ptr(), self.slots.data_ptr(), self._stream()), "pulse_qtable_rollout_step")
        self._launches += 1
        self._carried = True
        return env.boards, env.rewards, env.dones, env.truncated, {"score": env.total_score}

    def forget_states(self):
        self._carried = False

    def table(self, board=None):
        """{packed state key: q-values} of one board's region (private tables) or of the shared table."""
        if self.region_slots:
            lo, hi = board * self.region_slots, (board + 1) * self.region_slots
        else:
            lo, hi = 0, self.capacity
        keys = self.keys[lo:hi].cpu().numpy()
        vals = self.values[lo:hi].cpu().numpy()
        return {int(k) & (2**64 - 1): vals[i].copy() for i, k in enumerate(keys) if k != 0}
