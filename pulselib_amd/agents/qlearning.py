"""Batched tabular Q-learning for the 2048 roll-out: the GPU form of the reference's
agents/TemperalDifference/QLearningNumba.py:10-37 (+ utils/numba.py:5-39), one learner state per
launch for B boards.  `get_action` / `update` keep the reference's meaning per board; the Python dict
becomes a device hash table (csrc/qtable.hip).

private_tables=True  : every board learns in its own region (B independent copies of the reference
                       agent: bit-exact against the oracle, no races) -- the parity mode;
private_tables=False : all boards share one table (what a batched learner wants), concurrent updates
                       of one entry are applied atomically in arrival order."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native


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
        self.keys = torch.zeros(self.capacity, dtype=torch.int64, device=device)
        self.values = torch.zeros((self.capacity, 4), dtype=torch.float64, device=device)
        self.actions = torch.zeros(batch_size, dtype=torch.int64, device=device)
        self.slots = torch.zeros(batch_size, dtype=torch.int64, device=device)
        self._q = _native.QTable(self.keys.data_ptr(), self.values.data_ptr(), self.capacity, self.region_slots)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def get_actions(self, boards, step_counter):
        """epsilon-greedy action per board (QLearningNumba.py:24-27); remembers the looked-up states for update()."""
        boards = boards if boards.is_contiguous() else boards.contiguous()
        _native.check(self._lib.pulse_qtable_select(C.byref(self._q), boards.data_ptr(), self.batch_size, self.n, self.epsilon,
                                                    self.seed, self.board_id0, int(step_counter), self.actions.data_ptr(),
                                                    self.slots.data_ptr(), self._stream()), "pulse_qtable_select")
        return self.actions

    def update(self, next_boards, rewards, terminated):
        """Q update for the transitions (state looked up by the last get_actions, its action) -> next_boards."""
        next_boards = next_boards if next_boards.is_contiguous() else next_boards.contiguous()
        term = terminated.view(torch.uint8) if terminated.dtype == torch.bool else terminated.to(torch.uint8)
        _native.check(self._lib.pulse_qtable_update(C.byref(self._q), self.slots.data_ptr(), self.actions.data_ptr(),
                                                    rewards.data_ptr(), next_boards.data_ptr(), term.data_ptr(), self.batch_size,
                                                    self.n, self.alpha, self.gamma, self._stream()), "pulse_qtable_update")

    def table(self, board=None):
        """{packed state key: q-values} of one board's region (private tables) or of the shared table."""
        if self.region_slots:
            lo, hi = board * self.region_slots, (board + 1) * self.region_slots
        else:
            lo, hi = 0, self.capacity
        keys = self.keys[lo:hi].cpu().numpy()
        vals = self.values[lo:hi].cpu().numpy()
        return {int(k) & (2**64 - 1): vals[i].copy() for i, k in enumerate(keys) if k != 0}
