"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy binding of the CPU restatements in oracle/*.c (liboracle.so).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (pulselib_amd/) never does.

`OraclePokerEnv` mirrors the state names of the reference's
environments/Poker/PokerGPU.py (numpy arrays instead of torch tensors) so parity tests can compare
attribute by attribute.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

HR_LEN = 32487834
ACTIVE, FOLDED, ALLIN, SITOUT = 0, 1, 2, 3


def build(force: bool = False) -> Path:
    so = _HERE / "liboracle.so"
    srcs = [_HERE / "handranks_oracle.c", _HERE / "poker_oracle.c", _HERE / "envs_oracle.c"]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.hr_oracle_generate.restype = C.c_int
        _LIB.oracle_reward.restype = C.c_float
        _LIB.oracle_step_table.restype = C.c_float
        _LIB.oracle_select_action_epsilon_greedy.restype = C.c_int32
    return _LIB


_HR_CACHE = None


def hand_ranks(cache_path: str | None = None) -> np.ndarray:
    """The oracle's own HandRanks table (int32[32,487,834]); generated once per process, optionally
    cached on disk (default: $PULSE_ORACLE_HR or /tmp/pulse_oracle_HandRanks.dat)."""
    global _HR_CACHE
    if _HR_CACHE is not None:
        return _HR_CACHE
    path = Path(cache_path or os.environ.get("PULSE_ORACLE_HR", "/tmp/pulse_oracle_HandRanks.dat"))
    if path.exists() and path.stat().st_size == HR_LEN * 4:
        _HR_CACHE = np.fromfile(path, dtype=np.int32)
        return _HR_CACHE
    out = np.zeros(HR_LEN, dtype=np.int32)
    rc = lib().hr_oracle_generate(out.ctypes.data_as(C.c_void_p))
    if rc != 612977:
        raise RuntimeError(f"hr_oracle_generate failed: {rc}")
    try:
        tmp = path.with_suffix(f".tmp{os.getpid()}")
        out.tofile(tmp)
        os.replace(tmp, path)
    except OSError:
        pass
    _HR_CACHE = out
    return out


class _PokerStruct(C.Structure):
    _fields_ = (
        [("n_games", C.c_int32), ("n_players", C.c_int32), ("active_players", C.c_int32), ("obs_size", C.c_int32),
         ("hr_len", C.c_int64), ("hand_ranks", C.c_void_p)]
        + [(n, C.c_void_p) for n in (
            "pots", "stages", "deck_positions", "button", "sb", "bb", "idx", "highest", "agg", "acted",
            "last_raise_size", "prev_stacks", "prev_invested", "raise_amounts",
            "is_done", "equity_dirty", "is_round_over",
            "stacks", "current_round_bet", "total_invested", "status", "hands", "board", "decks",
            "equities", "obs")]
        + [("w1", C.c_float), ("w2", C.c_float), ("K", C.c_int32), ("alpha", C.c_int32)]
    )


_SCALARS_I32 = ("pots", "stages", "deck_positions", "button", "sb", "bb", "idx", "highest", "agg", "acted",
                "last_raise_size", "prev_stacks", "prev_invested", "raise_amounts")
_SCALARS_U8 = ("is_done", "equity_dirty", "is_round_over")
_ROWS = ("stacks", "current_round_bet", "total_invested", "status")

INT_STATE = _SCALARS_I32 + _SCALARS_U8 + _ROWS + ("hands", "board", "decks")


class OraclePokerEnv:
    """Numpy twin of the reference PokerGPU (PokerGPU.py:8-633) over liboracle.so."""

    NUM_ACTIONS = 13
    ACTIVE, FOLDED, ALLIN, SITOUT = 0, 1, 2, 3

    def __init__(self, n_players=6, max_players=10, n_games=100, starting_bbs=100, max_bbs=1000,
                 w1=.5, w2=.5, K=20, alpha=300, hand_ranks_table: np.ndarray | None = None, n_threads: int = 1):
        self.n_players, self.max_players, self.n_games = n_players, max_players, n_games
        self.starting_bbs, self.max_bbs = starting_bbs, max_bbs
        self.w1, self.w2, self.K, self.alpha = float(np.float32(w1)), float(np.float32(w2)), int(K), int(alpha)
        self.obs_size = 13 + (max_players - 1) * 3
        self.hand_ranks = hand_ranks() if hand_ranks_table is None else np.ascontiguousarray(hand_ranks_table, dtype=np.int32)
        self.n_threads = n_threads
        self.active_players = n_players
        self._first = True
        N, P = n_games, n_players
        for n in _SCALARS_I32:
            setattr(self, n, np.zeros(N, dtype=np.int32))
        for n in _SCALARS_U8:
            setattr(self, n, np.zeros(N, dtype=np.uint8))
        for n in _ROWS:
            setattr(self, n, np.zeros((N, P), dtype=np.int32))
        self.hands = np.full((N, P, 2), -1, dtype=np.int32)
        self.board = np.full((N, 5), -1, dtype=np.int32)
        self.decks = np.zeros((N, 52), dtype=np.int32)
        self.equities = np.full((N, P), .5, dtype=np.float32)
        self.obs = np.zeros((N, self.obs_size), dtype=np.float32)
        self.rewards = np.zeros(N, dtype=np.float32)

    # -- plumbing -----------------------------------------------------------------------------
    def _struct(self) -> _PokerStruct:
        s = _PokerStruct()
        s.n_games, s.n_players, s.active_players, s.obs_size = self.n_games, self.n_players, self.active_players, self.obs_size
        s.hr_len = self.hand_ranks.size
        s.hand_ranks = self.hand_ranks.ctypes.data
        for n in _SCALARS_I32 + _SCALARS_U8 + _ROWS + ("hands", "board", "decks", "equities", "obs"):
            a = getattr(self, n)
            assert a.flags["C_CONTIGUOUS"], n
            setattr(s, n, a.ctypes.data)
        s.w1, s.w2, s.K, s.alpha = self.w1, self.w2, self.K, self.alpha
        return s

    # -- reference surface --------------------------------------------------------------------
    def reset(self, options=None, rng: np.random.Generator | None = None):
        """PokerGPU.py:73-157.  `options["active_players"]` may be an int here to force A (the
        reference samples it with torch.randint); True samples 2..n_players from `rng`."""
        options = options or {}
        ap = options.get("active_players", False)
        if ap is True:
            cand = int((rng or np.random.default_rng()).integers(2, self.n_players + 1))
        elif ap:
            cand = int(ap)
        else:
            cand = self.n_players
        q_seat = options.get("q_agent_seat", 0)
        self.active_players = max(cand, q_seat + 1)
        decks = options.get("prefixed_decks")
        if decks is None:
            raise ValueError("the oracle needs prefixed_decks (RNG is injected)")
        decks = np.asarray(decks, dtype=np.int32)
        if decks.shape != (self.n_games, 52):
            raise ValueError(f"prefixed_decks must have shape {(self.n_games, 52)}, got {tuple(decks.shape)}")
        self.decks = np.ascontiguousarray(decks).copy()
        rotation = int(options.get("rotation", 0))
        A = self.active_players
        if self._first:
            buttons = np.zeros(self.n_games, dtype=np.int32)
        else:
            buttons = ((self.button + 1) % A).astype(np.int32)
        self.equities = np.full((self.n_games, A), .5, dtype=np.float32)
        s = self._struct()
        lib().oracle_reset(C.byref(s), C.c_int(1 if self._first else 0), C.c_int(self.starting_bbs), C.c_int(self.max_bbs),
                           C.c_int(rotation), buttons.ctypes.data_as(C.c_void_p), C.c_int(self.n_threads))
        self._first = False
        return self.obs, {"active_players": A, "stacks": self.stacks, "seat_idx": self.idx}

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int64)
        assert actions.shape == (self.n_games,)
        s = self._struct()
        lib().oracle_step(C.byref(s), actions.ctypes.data_as(C.c_void_p), self.rewards.ctypes.data_as(C.c_void_p),
                          C.c_int(self.n_threads))
        return self.obs, self.rewards, self.is_done.astype(bool), np.zeros(self.n_games, dtype=bool), \
            {"active_players": self.active_players, "stacks": self.stacks, "seat_idx": self.idx}

    def policy(self, agent_types, seed, step_counter, actions, table_id0=0):
        """build_actions with the scripted opponents (utils.py:108-123); fills `actions` in place."""
        types = (C.c_uint8 * self.n_players)(*[int(x) for x in agent_types])
        assert actions.dtype == np.int64 and actions.flags["C_CONTIGUOUS"]
        s = self._struct()
        lib().oracle_policy(C.byref(s), types, C.c_uint64(seed), C.c_uint64(step_counter), C.c_uint64(table_id0),
                            actions.ctypes.data_as(C.c_void_p), C.c_int(self.n_threads))
        return actions

    def policy_step(self, agent_types, seed, step_counter, actions, table_id0=0):
        types = (C.c_uint8 * self.n_players)(*[int(x) for x in agent_types])
        s = self._struct()
        lib().oracle_policy_step(C.byref(s), types, C.c_uint64(seed), C.c_uint64(step_counter), C.c_uint64(table_id0),
                                 actions.ctypes.data_as(C.c_void_p), self.rewards.ctypes.data_as(C.c_void_p),
                                 C.c_int(self.n_threads))
        return self.obs, self.rewards, self.is_done.astype(bool)

    # per-table method-level entry points for white-box checks
    def _each(self, fn, *extra):
        s = self._struct()
        for t in range(self.n_games):
            fn(C.byref(s), C.c_int(t), *extra)

    def get_obs(self):
        self._each(lib().oracle_get_obs)
        return self.obs

    def calculate_equities(self):
        self._each(lib().oracle_calculate_equities)

    def execute_actions(self, actions):
        s = self._struct()
        for t in range(self.n_games):
            lib().oracle_execute_actions(C.byref(s), C.c_int(t), C.c_int64(int(actions[t])))

    def resolve_fold_winners(self):
        s = self._struct()
        for t in range(self.n_games):
            lib().oracle_resolve_fold_winner(C.byref(s), C.c_int(t), C.c_int(int(self.is_done[t])))

    def resolve_terminated_games(self):
        s = self._struct()
        for t in range(self.n_games):
            lib().oracle_resolve_terminated(C.byref(s), C.c_int(t), C.c_int(int(self.is_done[t])))

    def post_blinds(self):
        self._each(lib().oracle_post_blinds)

    def snapshot(self) -> dict:
        d = {n: getattr(self, n).copy() for n in INT_STATE}
        d["equities"] = self.equities.copy()
        d["obs"] = self.obs.copy()
        return d


def eval_hands(hr: np.ndarray, cards: np.ndarray) -> np.ndarray:
    """5/6/7-card lookups exactly as PokerGPU.py:437-444 / :500 / :521 walk the table."""
    cards = np.ascontiguousarray(cards, dtype=np.int32)
    out = np.zeros(cards.shape[0], dtype=np.int32)
    lib().oracle_eval_hands(hr.ctypes.data_as(C.c_void_p), C.c_int64(hr.size), cards.ctypes.data_as(C.c_void_p),
                            C.c_int(cards.shape[0]), C.c_int(cards.shape[1]), out.ctypes.data_as(C.c_void_p))
    return out


def scripted_actions_rows(types, c1, c2, pot, pick: int, coin: int, actions=None) -> np.ndarray:
    """oracle_scripted_action over rows with the two draw words given (oracle/poker_oracle.c)."""
    types = np.ascontiguousarray(types, dtype=np.uint8)
    c1, c2, pot = (np.ascontiguousarray(x, dtype=np.int32) for x in (c1, c2, pot))
    out = np.full(types.size, -7, dtype=np.int64) if actions is None else actions
    lib().oracle_scripted_actions_rows(types.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p), c2.ctypes.data_as(C.c_void_p),
                                       pot.ctypes.data_as(C.c_void_p), C.c_int(types.size), C.c_uint32(pick), C.c_uint32(coin),
                                       out.ctypes.data_as(C.c_void_p))
    return out


def shuffle_decks(seed: int, table_id0: int, episode: int, n_tables: int, key_bits: int = 0) -> np.ndarray:
    """The device shuffle's definition (oracle/poker_oracle.c: oracle_shuffle_decks) -> int32[n_tables, 52]."""
    decks = np.zeros((n_tables, 52), dtype=np.int32)
    lib().oracle_shuffle_decks(C.c_uint64(seed), C.c_uint64(table_id0), C.c_uint64(episode), C.c_int(key_bits),
                               C.c_int(n_tables), decks.ctypes.data_as(C.c_void_p))
    return decks


def _qnet_ptrs(weights, biases):
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    arr_w = (C.c_void_p * 5)(*[w.ctypes.data for w in ws])
    arr_b = (C.c_void_p * 5)(*[b.ctypes.data for b in bs])
    return ws, bs, arr_w, arr_b


def qnet_forward(weights, biases, states: np.ndarray) -> np.ndarray:
    """oracle/qnet_oracle.c: PokerQNetwork.network in eval mode; weights/biases = the five Linear layers (torch layout)."""
    ws, bs, arr_w, arr_b = _qnet_ptrs(weights, biases)
    states = np.ascontiguousarray(states, dtype=np.float32)
    n, k = states.shape
    assert ws[0].shape == (128, k)
    n_actions = ws[4].shape[0]
    out = np.zeros((n, n_actions), dtype=np.float32)
    lib().oracle_qnet_forward(C.c_int(k), C.c_int(n_actions), arr_w, arr_b, states.ctypes.data_as(C.c_void_p), C.c_long(k),
                              C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def qnet_act(q: np.ndarray, seat_idx, q_seat: int, epsilon: float, seed: int, step: int, table_id0: int,
             actions: np.ndarray) -> np.ndarray:
    """oracle/qnet_oracle.c: epsilon-greedy over Q rows for the rows of the learner's seat; `actions` updated in place."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    assert actions.dtype == np.int64 and actions.flags["C_CONTIGUOUS"]
    si = None if seat_idx is None else np.ascontiguousarray(seat_idx, dtype=np.int32)
    lib().oracle_qnet_act(C.c_int(q.shape[1]), q.ctypes.data_as(C.c_void_p), C.c_int(q.shape[0]),
                          None if si is None else si.ctypes.data_as(C.c_void_p), C.c_int(q_seat), C.c_float(epsilon),
                          C.c_uint64(seed), C.c_uint64(step), C.c_uint64(table_id0), actions.ctypes.data_as(C.c_void_p))
    return actions


def qnet_param_count(state_dim: int, n_actions: int) -> int:
    return 128 * state_dim + 128 + 128 * 128 + 128 + 64 * 128 + 64 + 32 * 64 + 32 + 32 * n_actions + n_actions


def qnet_split(flat: np.ndarray, state_dim: int, n_actions: int):
    """flat parameter vector (w1,b1,...,w5,b5) -> ([w1..w5], [b1..b5]) views."""
    dims = [state_dim, 128, 128, 64, 32, n_actions]
    ws, bs, o = [], [], 0
    for l in range(5):
        n = dims[l + 1] * dims[l]
        ws.append(flat[o:o + n].reshape(dims[l + 1], dims[l])); o += n
        bs.append(flat[o:o + dims[l + 1]]); o += dims[l + 1]
    assert o == flat.size
    return ws, bs


def qnet_train_grads(params: np.ndarray, target: np.ndarray, states, actions, rewards, next_states, dones, row_mask,
                     gamma: float, drop_p: float, seed: int, step: int, table_id0: int):
    """oracle/qnet_oracle.c: oracle_qnet_train_grads -> (grad_sum flat fp32, row count, sum of squared TD errors)."""
    states = np.ascontiguousarray(states, dtype=np.float32)
    next_states = np.ascontiguousarray(next_states, dtype=np.float32)
    n, k = states.shape
    params = np.ascontiguousarray(params, dtype=np.float32)
    target = np.ascontiguousarray(target, dtype=np.float32)
    n_actions = (params.size - (128 * k + 128 + 128 * 128 + 128 + 64 * 128 + 64 + 32 * 64 + 32)) // 33
    ws, bs, aw, ab = _qnet_ptrs(*qnet_split(params, k, n_actions))
    tws, tbs, atw, atb = _qnet_ptrs(*qnet_split(target, k, n_actions))
    actions = np.ascontiguousarray(actions, dtype=np.int64)
    rewards = np.ascontiguousarray(rewards, dtype=np.float32)
    dones = np.ascontiguousarray(dones, dtype=np.uint8)
    rm = None if row_mask is None else np.ascontiguousarray(row_mask, dtype=np.uint8)
    grad = np.zeros(params.size, dtype=np.float32)
    sq = C.c_float(0)
    fn = lib().oracle_qnet_train_grads
    fn.restype = C.c_int
    cnt = fn(C.c_int(k), C.c_int(n_actions), aw, ab, atw, atb, states.ctypes.data_as(C.c_void_p), C.c_long(k),
             actions.ctypes.data_as(C.c_void_p), rewards.ctypes.data_as(C.c_void_p), next_states.ctypes.data_as(C.c_void_p),
             C.c_long(k), dones.ctypes.data_as(C.c_void_p), None if rm is None else rm.ctypes.data_as(C.c_void_p), C.c_int(n),
             C.c_float(gamma), C.c_float(drop_p), C.c_uint64(seed), C.c_uint64(step), C.c_uint64(table_id0),
             grad.ctypes.data_as(C.c_void_p), C.byref(sq))
    return grad, int(cnt), float(sq.value)


def qnet_adamw(params, target, grad, m, v, count: int, t: int, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=1.0,
               update_freq=0) -> float:
    """oracle/qnet_oracle.c: oracle_qnet_adamw; params/target/m/v (contiguous fp32) are updated in place."""
    for a in (params, target, grad, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    fn = lib().oracle_qnet_adamw
    fn.restype = C.c_float
    return float(fn(C.c_int(params.size), params.ctypes.data_as(C.c_void_p), target.ctypes.data_as(C.c_void_p),
                    grad.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_int(count),
                    C.c_long(t), C.c_float(lr), C.c_float(wd), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                    C.c_float(max_norm), C.c_int(update_freq)))


def philox4x32(seed: int, subseq: int, offset: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32(C.c_uint64(seed), C.c_uint64(subseq), C.c_uint64(offset), out.ctypes.data_as(C.c_void_p))
    return out


# ---------------------------------------------------------------------------------------------
# Blackjack / 2048 / Particle2D oracles (oracle/envs_oracle.c)
class _BJStruct(C.Structure):
    _fields_ = [("batch_size", C.c_int32)] + [(n, C.c_void_p) for n in (
        "decks", "deck_positions", "players_cards", "players_card_idx", "player_card_sums",
        "dealer_cards", "dealer_card_idx", "dealer_upcard", "dealer_card_sums",
        "terminated", "has_ace", "dealer_has_ace", "rewards", "obs")]


class OracleBlackjack:
    """Numpy twin of the reference BlackJack (environments/blackjack/blackjack.py) over liboracle.so."""

    I32 = ("deck_positions", "players_card_idx", "player_card_sums", "dealer_card_idx", "dealer_upcard", "dealer_card_sums", "rewards")
    U8 = ("terminated", "has_ace", "dealer_has_ace")

    def __init__(self, batch_size):
        B = self.batch_size = batch_size
        self.decks = np.zeros((B, 52), dtype=np.int32)
        for n in self.I32:
            setattr(self, n, np.zeros(B, dtype=np.int32))
        for n in self.U8:
            setattr(self, n, np.zeros(B, dtype=np.uint8))
        self.players_cards = np.zeros((B, 20), dtype=np.int32)
        self.dealer_cards = np.zeros((B, 20), dtype=np.int32)
        self.obs = np.zeros((B, 3), dtype=np.int32)

    def _struct(self):
        s = _BJStruct()
        s.batch_size = self.batch_size
        for n in ("decks", "players_cards", "dealer_cards", "obs") + self.I32 + self.U8:
            setattr(s, n, getattr(self, n).ctypes.data)
        return s

    def reset(self, decks):
        self.decks = np.ascontiguousarray(decks, dtype=np.int32).copy()
        s = self._struct()
        lib().oracle_blackjack_reset(C.byref(s))
        return self.obs

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int64)
        s = self._struct()
        lib().oracle_blackjack_step(C.byref(s), actions.ctypes.data_as(C.c_void_p))
        return self.obs, self.rewards, self.terminated.astype(bool)


def tfe_reset(boards, total_score, n, seed, board_id0=0):
    lib().oracle_tfe_reset(boards.ctypes.data_as(C.c_void_p), total_score.ctypes.data_as(C.c_void_p), C.c_int(boards.shape[0]),
                           C.c_int(n), C.c_uint64(seed), C.c_uint64(board_id0))


def tfe_step(boards, total_score, actions, rewards, dones, n, seed, step_counter, board_id0=0):
    actions = np.ascontiguousarray(actions, dtype=np.int64)
    lib().oracle_tfe_step(boards.ctypes.data_as(C.c_void_p), total_score.ctypes.data_as(C.c_void_p),
                          actions.ctypes.data_as(C.c_void_p), rewards.ctypes.data_as(C.c_void_p),
                          dones.ctypes.data_as(C.c_void_p), C.c_int(boards.shape[0]), C.c_int(n), C.c_uint64(seed),
                          C.c_uint64(board_id0), C.c_uint64(step_counter))


def particle2d_step(state, action, steps, dt, max_steps):
    n = state.shape[0]
    obs = np.zeros_like(state)
    rewards = np.zeros(n, dtype=np.float32)
    term = np.zeros(n, dtype=np.uint8)
    action = np.ascontiguousarray(action, dtype=np.float32)
    lib().oracle_particle2d_step(state.ctypes.data_as(C.c_void_p), action.ctypes.data_as(C.c_void_p),
                                 steps.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p),
                                 rewards.ctypes.data_as(C.c_void_p), term.ctypes.data_as(C.c_void_p), C.c_int(n),
                                 C.c_float(dt), C.c_int(max_steps))
    return obs, rewards, term.astype(bool)
