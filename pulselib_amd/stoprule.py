"""Episode stop rule of the reference trainer without its host sync.

scripts/Poker/trainGPU.py:27-33 ends an episode when, at every 5th step, `terminated.float().mean() > 0.8` -- a
blocking device->host read in the middle of the loop.  `LaggedDoneCount` keeps the rule and drops the wait: after a
chunk of steps the number of finished tables is counted on the device (cumulative counter: no memset in the loop),
copied to pinned host memory on a side stream, and the decision is taken on the newest count that has ALREADY
arrived -- normally the one of the previous chunk.  `blocking=True` waits for the current chunk's count instead (the
reference's behaviour).  The mechanics live in the native library (pulse_stoprule_*, csrc/poker.hip): done through
torch, the chunk boundary cost more host time than the chunk's five step launches take on the GPU."""
from __future__ import annotations

import ctypes as C

import torch

from . import _native


class LaggedDoneCount:
    def __init__(self, device, n_tables: int, threshold: float = 0.8):
        self.device, self.n, self.threshold = device, int(n_tables), float(threshold)
        self._lib = _native.lib()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _native.check(self._lib.pulse_stoprule_create(self.n, self.threshold, C.byref(h)), "pulse_stoprule_create")
        self.handle = h

    def submit(self, is_done: torch.Tensor) -> None:
        """Count the set flags of `is_done` (bool/uint8[n]) in stream order and start the copy to the host.
        (PokerGPU.rollout(..., stop_rule=self) does this inside the same native call as the step launches.)"""
        _native.check(self._lib.pulse_stoprule_submit(self.handle, is_done.data_ptr(),
                                                      torch.cuda.current_stream(self.device).cuda_stream), "pulse_stoprule_submit")

    def over(self, blocking: bool = False) -> bool:
        """True if any count that has reached the host since the last call exceeds the threshold."""
        flag = C.c_int32(0)
        _native.check(self._lib.pulse_stoprule_over(self.handle, 1 if blocking else 0, C.byref(flag)), "pulse_stoprule_over")
        return bool(flag.value)

    def drain(self) -> None:
        """Episode boundary: consume what is in flight so the cumulative counters stay consistent."""
        _native.check(self._lib.pulse_stoprule_drain(self.handle), "pulse_stoprule_drain")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.pulse_stoprule_destroy(self.handle)
            self.handle = None

    # no __del__: at interpreter shutdown the HIP runtime may already be gone, and an unclosed handle only leaks a side
    # stream, four events and 16 bytes; long-lived callers call close()
