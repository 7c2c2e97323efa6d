"""Hand-level performance metrics without per-step host syncs.

The reference's benchmark trainer (scripts/Poker/trainGPU_performance.py:198-206) pulls, on every step,
`stacks[newly_done, q_seat]`, `stages[newly_done]` and the seat positions through boolean indexing -- a device->host
sync each -- and keeps per-hand lists that utils/performance.py reduces at the end.  Every metric it reports except
the rolling window is a function of per-group sums, so `HandMetrics` keeps exactly those on the device
(pulse_poker_hand_metrics: hands, wins, sum delta, sum delta^2 per (button-relative position, street bucket), exact
int64) and folds them into host totals per episode, keyed by the episode's player count.  The rolling window
(`rolling_window_size`, utils/performance.py:128-135) needs the hands IN ORDER: the same launch notes, per table, the
step it finished at and its delta; at the episode's end one sort by (step, table) restores the order of the reference's
per-step boolean pulls -- still no sync inside the episode.  Metric names and formulas follow utils/performance.py
(cited per function); values are plain floats."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _native

STREET_DEPTH_NAMES = {0: "preflop", 1: "flop", 2: "turn", 3: "river", 4: "showdown"}     # utils/performance.py:8-14
CONFIDENCE_Z_95 = 1.959963984540054                                                      # :15
MAX_SEATS, BUCKETS = 16, 5


def calculate_q_seat_positions(buttons: torch.Tensor, *, q_seat: int, active_players: int) -> torch.Tensor:
    """Button-relative position of the learner's seat per table (utils/performance.py:55-59)."""
    return torch.remainder(q_seat - buttons, active_players).to(torch.int64)


def bb_per_100(count: int, total: float) -> float:
    """mean big blinds won per 100 hands (:104-109)"""
    return 100.0 * total / count if count else 0.0


def lcb95_bb_per_100(count: int, total: float, total_sq: float) -> float:
    """lower 95 % confidence bound of BB/100 with the population std (:112-125)"""
    if count == 0:
        return 0.0
    mean = total / count
    if count == 1:
        return 100.0 * mean
    var = max(total_sq / count - mean * mean, 0.0)
    return 100.0 * (mean - CONFIDENCE_Z_95 * math.sqrt(var) / math.sqrt(count))


def build_prefixed_deck_batch(*, n_games: int, seed: int, device) -> torch.Tensor:
    """One deterministic shuffled deck per game from a CPU generator seeded `seed` (utils/performance.py:62-67;
    scripts/Poker/trainGPU_performance.py:52 uses seed = 20260401 + episode)."""
    generator = torch.Generator(device="cpu")
    generator.manual_seed(int(seed))
    shuffled = torch.rand((n_games, 52), generator=generator).argsort(dim=1) + 1
    return shuffled.to(device=device, dtype=torch.int32)


def calculate_rolling_window_averages(hand_bb_deltas, *, window_size: int) -> torch.Tensor:
    """Rolling mean big-blind delta over completed hands, in the order given (utils/performance.py:128-135: the fp32
    `unfold(0, W, 1).mean(dim=1)` of the flattened per-hand list).  `hand_bb_deltas`: a tensor or a list of tensors (one
    per step / per episode, in order).  Returns a float32 tensor of len(deltas) - W + 1 averages (empty if fewer hands)."""
    batches = hand_bb_deltas if isinstance(hand_bb_deltas, (list, tuple)) else [hand_bb_deltas]
    batches = [torch.as_tensor(b) for b in batches if b is not None]
    if not batches:
        return torch.empty(0, dtype=torch.float32)
    deltas = torch.cat([b.reshape(-1).to(device=batches[0].device, dtype=torch.float32) for b in batches], dim=0)
    if window_size <= 0 or deltas.numel() < window_size:
        return torch.empty(0, dtype=torch.float32, device=deltas.device)
    return deltas.unfold(0, window_size, 1).mean(dim=1)


def rolling_bb_window_summary(hand_bb_deltas, window_size: int) -> dict:
    """The `rolling_bb_window` block of calculate_final_performance_metrics (utils/performance.py:452-457)."""
    avg = calculate_rolling_window_averages(hand_bb_deltas, window_size=window_size)
    n = int(avg.numel())
    return {"window_size": int(window_size), "num_windows": n, "last_average": float(avg[-1]) if n else 0.0,
            "best_average": float(avg.max()) if n else 0.0, "values": avg.detach().cpu().numpy()}


class HandMetrics:
    """Device accumulators for one run; `begin_episode` / `update` per step / `end_episode`, then `summary()`.
    `rolling_window_size` > 0 also keeps the ordered per-hand deltas (8 bytes per table on the device, one sort per
    episode) and adds the `rolling_bb_window` block to the summary."""

    def __init__(self, device, n_tables: int, rolling_window_size: int = 0):
        self.rolling_window_size = int(rolling_window_size)
        self.hand_deltas = []                # per episode: float32 tensor of the finished hands' deltas, in the reference's order
        self._step = 0
        self._finish = self._delta = self._table_ids = None
        if self.rolling_window_size > 0:
            self._finish = torch.full((int(n_tables),), -1, dtype=torch.int32, device=device)
            self._delta = torch.zeros(int(n_tables), dtype=torch.int32, device=device)
            self._table_ids = torch.arange(int(n_tables), dtype=torch.int64, device=device)
        self.device, self.n = device, int(n_tables)
        self.acc = torch.zeros(MAX_SEATS * BUCKETS * 4, dtype=torch.int64, device=device)
        self.initial = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.totals = {}                     # (player_count, opponent-mix id) -> int64[16,5,4] on the host
        self.q_seat, self.active_players, self.mix_id = 0, 2, 0
        self._lib = _native.lib()

    def begin_episode(self, env, q_seat: int, mix_id: int = 0) -> None:
        """after env.reset: remember the learner's starting stacks (trainGPU_performance.py:165); `mix_id` labels the
        episode's opponent pool for the `opponent_mix` slices (:288-297)"""
        self.q_seat, self.active_players, self.mix_id = int(q_seat), int(env.active_players), int(mix_id)
        self.initial.copy_(env.stacks[:, self.q_seat])
        self.acc.zero_()
        self._step = 0
        if self._finish is not None:
            self._finish.fill_(-1)

    def update(self, env, dones: torch.Tensor, terminated_before: torch.Tensor | None) -> None:
        """after env.step and BEFORE `terminated |= dones`: account the hands that finished in this step (:192-206)."""
        d = dones.view(torch.uint8) if dones.dtype == torch.bool else dones
        t = None if terminated_before is None else (terminated_before.view(torch.uint8) if terminated_before.dtype == torch.bool
                                                    else terminated_before)
        _native.check(self._lib.pulse_poker_hand_metrics(
            d.data_ptr(), None if t is None else t.data_ptr(), env.stacks.data_ptr(), env.n_players, self.initial.data_ptr(),
            env.stages.data_ptr(), env.button.data_ptr(), self.q_seat, self.active_players, self.n, self.acc.data_ptr(),
            self._step, None if self._finish is None else self._finish.data_ptr(), None if self._delta is None else self._delta.data_ptr(),
            torch.cuda.current_stream(self.device).cuda_stream), "pulse_poker_hand_metrics")
        self._step += 1

    def end_episode(self) -> dict:
        """One read-back per episode (at the point where the trainer reads its episode sums anyway); returns the
        episode's summary with the keys of summarize_episode_performance_metrics (:138-167)."""
        a = self.acc.cpu().numpy().reshape(MAX_SEATS, BUCKETS, 4).copy()
        tot = self.totals.setdefault((self.active_players, self.mix_id), np.zeros((MAX_SEATS, BUCKETS, 4), dtype=np.int64))
        tot += a
        hands, wins, total = int(a[..., 0].sum()), int(a[..., 1].sum()), float(a[..., 2].sum())
        if self._finish is not None and hands:
            # the hands in the reference's order: by the step they finished at, then by table (trainGPU_performance.py:198-206)
            key = torch.where(self._finish >= 0, self._finish.to(torch.int64) * self.n + self._table_ids, torch.iinfo(torch.int64).max)
            order = torch.argsort(key)[:hands]
            self.hand_deltas.append(self._delta[order].to(torch.float32))
        return {"mean_bb_delta": total / hands if hands else 0.0, "hand_win_rate": wins / hands if hands else 0.0,
                "hands_completed": hands, "field_bb_per_100": bb_per_100(hands, total)}

    def summary(self) -> dict:
        out = summarize_totals(self.totals)
        if self.rolling_window_size > 0:
            out["rolling_bb_window"] = rolling_bb_window_summary(self.hand_deltas, self.rolling_window_size)
        return out


def accumulate_hands(deltas, stages, positions, player_counts, mix_ids=None) -> dict:
    """The sufficient statistics pulse_poker_hand_metrics keeps on the device, built on the host from per-hand lists
    (chip delta, terminal stage, button-relative position, player count and opponent-mix id of the episode):
    {(player_count, mix_id): int64[16,5,4]} with cells {hands, wins, sum delta, sum delta^2}; stage buckets as
    bucketize_terminal_stages (:170-173)."""
    totals = {}
    d = np.asarray(deltas, dtype=np.int64)
    b = np.where(np.asarray(stages) >= 4, 4, np.clip(np.asarray(stages), 0, 3))
    pos, cnt = np.asarray(positions, dtype=np.int64), np.asarray(player_counts, dtype=np.int64)
    mix = np.zeros_like(cnt) if mix_ids is None else np.asarray(mix_ids, dtype=np.int64)
    for a, x in sorted(set(zip(cnt.tolist(), mix.tolist()))):
        t = totals.setdefault((int(a), int(x)), np.zeros((MAX_SEATS, BUCKETS, 4), dtype=np.int64))
        m = (cnt == a) & (mix == x)
        np.add.at(t[..., 0], (pos[m], b[m]), 1)
        np.add.at(t[..., 1], (pos[m], b[m]), (d[m] > 0).astype(np.int64))
        np.add.at(t[..., 2], (pos[m], b[m]), d[m])
        np.add.at(t[..., 3], (pos[m], b[m]), d[m] * d[m])
    return totals


def summarize_totals(totals: dict) -> dict:
    """The hand-derived part of calculate_final_performance_metrics (:352-...): totals, BB/100 with its lower bound,
    seat-balanced BB/100 (:242-253), win share by street (:176-196), win rate by position (:199-221) and the BB/100
    slices by opponent mix / seat / player count / street depth (:256-318) with the worst slice (:321-349, first
    minimum in that family order).  `totals` = {(player_count, mix_id): int64[16,5,4]} (HandMetrics.totals /
    accumulate_hands)."""
    if not totals:
        allt = np.zeros((MAX_SEATS, BUCKETS, 4), dtype=np.int64)
    else:
        allt = sum(totals.values())
    hands, wins = int(allt[..., 0].sum()), int(allt[..., 1].sum())
    total, total_sq = float(allt[..., 2].sum()), float(allt[..., 3].sum())
    by_pos = allt.sum(axis=1)            # [16, 4]
    by_street = allt.sum(axis=0)         # [5, 4]
    seats = [p for p in range(MAX_SEATS) if by_pos[p, 0] > 0]
    seat_slices = {f"position_{p}": bb_per_100(int(by_pos[p, 0]), float(by_pos[p, 2])) for p in seats}
    def grouped(which):
        g = {}
        for key, t in totals.items():
            g[key[which]] = g.get(key[which], 0) + t
        return sorted(g.items())

    slices = {
        "opponent_mix": {f"mix_{x}": bb_per_100(int(t[..., 0].sum()), float(t[..., 2].sum())) for x, t in grouped(1) if t[..., 0].sum() > 0},
        "seat": seat_slices,
        "player_count": {f"players_{a}": bb_per_100(int(t[..., 0].sum()), float(t[..., 2].sum())) for a, t in grouped(0)
                         if t[..., 0].sum() > 0},
        "street_depth": {STREET_DEPTH_NAMES[b]: bb_per_100(int(by_street[b, 0]), float(by_street[b, 2])) for b in range(BUCKETS)
                         if by_street[b, 0] > 0},
    }
    flat = [(v, fam, name) for fam, d in slices.items() for name, v in d.items()]
    worst = min(flat, key=lambda x: x[0]) if flat else (0.0, "", "")          # first minimum, like argmin (:339-341)
    return {
        "total_hands": hands,
        "total_bb_won": total,
        "overall_hand_win_rate": wins / hands if hands else 0.0,
        "field_bb_per_100": bb_per_100(hands, total),
        "lcb95_bb_per_100": lcb95_bb_per_100(hands, total, total_sq),
        "seat_balanced_bb_per_100": float(np.mean(list(seat_slices.values()))) if seat_slices else 0.0,
        "street_win_percentages": {STREET_DEPTH_NAMES[b]: (int(by_street[b, 1]) / hands if hands else 0.0) for b in range(BUCKETS)},
        "position_win_rates": {f"position_{p}": {"hands": int(by_pos[p, 0]), "wins": int(by_pos[p, 1]),
                                                 "win_rate": int(by_pos[p, 1]) / int(by_pos[p, 0])} for p in seats},
        "slices": slices,
        "worst_slice": {"bb_per_100": worst[0], "family": worst[1], "slice": worst[2]},
    }
