"""Episode stop rule of the reference trainer without its host sync.

scripts/Poker/trainGPU.py:27-33 ends an episode when, at every 5th step, `terminated.float().mean() > 0.8` -- a
blocking device->host read in the middle of the loop.  `LaggedDoneCount` keeps the rule and drops the wait: after a
chunk of steps the number of finished tables is counted on the device (pulse_poker_stats adds into a cumulative
counter: no memset in the loop), copied to pinned host memory on a side stream, and the decision is taken on the
newest count that has ALREADY arrived -- normally the one of the previous chunk.  `blocking=True` waits for the
current chunk's count instead (the reference's behaviour)."""
from __future__ import annotations

import torch

from . import _native


class LaggedDoneCount:
    def __init__(self, device, n_tables: int, threshold: float = 0.8):
        self.device, self.n, self.threshold = device, int(n_tables), float(threshold)
        self.side = torch.cuda.Stream(device=device)
        self.counts_dev = torch.zeros(2, dtype=torch.int64, device=device)
        self.counts_host = torch.zeros(2, dtype=torch.int64).pin_memory()
        self.copy_events = [torch.cuda.Event(), torch.cuda.Event()]
        self.seen = [0, 0]             # cumulative counts already consumed per slot
        self.pending = []              # chunk ids whose count is in flight
        self.chunk = 0
        self._late_over = False
        self._lib = _native.lib()

    def _pop(self) -> bool:
        c = self.pending.pop(0)
        total = int(self.counts_host[c & 1].item())
        n_done = total - self.seen[c & 1]
        self.seen[c & 1] = total
        return n_done > self.threshold * self.n

    def submit(self, is_done: torch.Tensor) -> None:
        """Count the set flags of `is_done` (bool/uint8[n]) in stream order and start the copy to the host."""
        slot = self.chunk & 1
        while len(self.pending) >= 2:                      # bounded run-ahead: never reuse a slot still in flight
            self.copy_events[self.pending[0] & 1].synchronize()
            self._late_over = self._pop() or self._late_over
        _native.check(self._lib.pulse_poker_stats(is_done.data_ptr(), None, None, self.n, self.counts_dev[slot:].data_ptr(), None,
                                                  torch.cuda.current_stream(self.device).cuda_stream), "pulse_poker_stats")
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.counts_host[slot:slot + 1].copy_(self.counts_dev[slot:slot + 1], non_blocking=True)
            self.copy_events[slot].record(self.side)
        self.pending.append(self.chunk)
        self.chunk += 1

    def over(self, blocking: bool = False) -> bool:
        """True if any count that has reached the host since the last call exceeds the threshold."""
        over, self._late_over = self._late_over, False
        while self.pending:
            ev = self.copy_events[self.pending[0] & 1]
            if blocking:
                ev.synchronize()
            elif not ev.query():
                break
            over = self._pop() or over
        return over

    def drain(self) -> None:
        """Episode boundary: consume what is in flight so the cumulative counters stay consistent."""
        while self.pending:
            self.copy_events[self.pending[0] & 1].synchronize()
            self._pop()
        self._late_over = False
