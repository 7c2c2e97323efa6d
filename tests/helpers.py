"""Shared helpers for the parity tests (data plumbing only)."""
from __future__ import annotations

import numpy as np

DYN_I32 = ("pots", "stages", "deck_positions", "button", "sb", "bb", "idx", "highest", "agg", "acted",
           "last_raise_size", "prev_stacks", "prev_invested")
DYN_BOOL = ("is_done", "equity_dirty")
DYN_ROWS = ("stacks", "current_round_bet", "total_invested", "status", "hands", "board")
INT_KEYS = DYN_I32 + DYN_BOOL + DYN_ROWS


def to_np(x):
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()


def assert_state_equal(got: dict, want: dict, keys=INT_KEYS, ctx=""):
    for k in keys:
        g = to_np(got[k]).astype(np.int64)
        w = to_np(want[k]).astype(np.int64)
        if g.shape != w.shape or not np.array_equal(g, w):
            bad = np.argwhere(g != w) if g.shape == w.shape else None
            first = bad[0] if bad is not None and len(bad) else None
            raise AssertionError(f"{ctx}: state '{k}' differs (shape {g.shape} vs {w.shape}); first mismatch at {first}: "
                                 f"got {g[tuple(first)] if first is not None else None} want {w[tuple(first)] if first is not None else None}; "
                                 f"n_bad={0 if bad is None else len(bad)}")


def reward_tol(alpha: float) -> float:
    # fp32 tanh: the HIP path and the oracle round a double-precision tanh once; torch's CPU fp32
    # tanh is within 1 ulp of that.  |reward| <= alpha, so 4 ulp of 1.0 scaled by alpha is generous.
    return 4 * 6e-8 * max(abs(alpha), 1.0) + 1e-7
