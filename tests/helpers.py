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


def reward_tol(alpha: float, want=None):
    """Bound on |reward - reference reward|, reward = fl(alpha * tanh_f32(x)) (PokerGPU.py:329).

    The HIP path and the oracle round a double-precision tanh once (correctly rounded but for ~1e-8 of arguments);
    torch's CPU fp32 tanh is within 1 ulp of that.  So the two tanh values differ by at most 1 ulp OF THE TANH VALUE,
    the product with alpha carries that over scaled by alpha, and each side rounds its product once more (1/2 ulp of
    the reward each).  Per element, with t = |want| / alpha:  tol = alpha * ulp(t) + ulp(want)  (<= 3 ulp of the reward).
    Without `want`: the scalar worst case of the same bound (|tanh| < 1, |reward| <= alpha)."""
    a = np.float32(max(abs(alpha), 1.0))
    if want is None:
        return float(a * np.spacing(np.float32(0.5)) + np.spacing(a))
    w = np.abs(np.asarray(want, dtype=np.float32))
    return (a * np.spacing(w / a) + np.spacing(w)).astype(np.float64)


def assert_rewards_close(got, want, alpha, ctx=""):
    got, want = to_np(got).astype(np.float64), to_np(want)
    tol = reward_tol(alpha, want)
    bad = np.abs(got - want.astype(np.float64)) > tol
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError(f"{ctx}: {int(bad.sum())} rewards outside 1 ulp(tanh) * alpha + 1 ulp(reward); first at {i}: got {got.flat[i]!r} "
                             f"want {float(want.flat[i])!r} tol {tol.flat[i]:.3g}")
