"""MI355X drop-in for the reference's batched hold'em environment.

Same class name, constructor, methods, option keys, attribute names / shapes / dtypes and error
texts as /root/reference/environments/Poker/PokerGPU.py (cited as PokerGPU.py:line below), but
every method that touches table state is one launch of a hand-written gfx950 kernel through the
C ABI in include/pulse_env.h -- `step()` is a single fused launch with no host sync, against
~1,700 eager dispatches and ~80 syncs in the reference.

Differences a caller can observe (all deliberate):
  * `device` must be a ROCm GPU ("cuda" device type); there is no CPU path.
  * State tensors are allocated once in __init__ and refilled by `reset` (stable HBM pointers)
    instead of being re-created every episode; `decks` is int32 also when shuffled on device.
  * `HandRanks.dat` is generated natively on first use instead of downloaded.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np
import torch

from ... import _native, handranks

try:  # gymnasium is optional: the class works as a plain object without it
    import gymnasium as gym
    from gymnasium import spaces
    _EnvBase = gym.Env
except Exception:  # pragma: no cover - gymnasium is absent in the build image
    gym = None
    spaces = None

    class _EnvBase:  # minimal stand-in for gym.Env
        metadata: dict = {}

        def reset(self, seed=None, options=None):
            return None

        def close(self):
            return None

        @property
        def unwrapped(self):
            return self


_I32_SCALARS = ("pots", "stages", "deck_positions", "button", "sb", "bb", "idx", "highest", "agg", "acted",
                "last_raise_size", "prev_stacks", "prev_invested")
_BOOL_SCALARS = ("is_done", "equity_dirty")
_ROWS = ("stacks", "current_round_bet", "total_invested", "status")
_TRACKED = frozenset(_I32_SCALARS + _BOOL_SCALARS + _ROWS + ("hands", "board", "decks", "equities", "obs",
                                                             "w1", "w2", "K", "alpha", "hand_ranks",
                                                             "active_players", "n_players", "n_games", "max_players",
                                                             "use_eval_cache", "obs_staging", "chunked_rollout", "chunk_four_lanes", "paired_launches"))


class PokerGPU(_EnvBase):
    metadata = {'render.modes': ['human']}
    NUM_ACTIONS = 13
    ACTIVE, FOLDED, ALLIN, SITOUT = 0, 1, 2, 3
    STATE_SPACE = 28
    MIN_EQUITY_RANK = 4145.0
    MAX_EQUITY_RANK = 36874.0
    MAX_FLOP_EQUITY = 823779.0
    MIN_FLOP_EQUITY = 74359.0
    MIN_TURN_RIVER_EQUITY = 4109.0
    MAX_TURN_RIVER_EQUITY = 36874.0

    # ------------------------------------------------------------------ construction (PokerGPU.py:20-68)
    def __init__(self, device, agents, n_players=6, max_players=10, n_games=100, starting_bbs=100, max_bbs=1000,
                 w1=.5, w2=.5, K=20, alpha=300, seed=0, table_id0=0):
        super().__init__()
        object.__setattr__(self, "_view_dirty", True)
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("pulselib_amd.PokerGPU runs on an MI355X (torch device type 'cuda' under ROCm); "
                               f"got device '{device}'. There is no CPU fallback.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if not (2 <= n_players <= max_players <= _native.MAX_SEATS):
            raise ValueError(f"need 2 <= n_players <= max_players <= {_native.MAX_SEATS}, got {n_players}, {max_players}")
        self._lib = _native.lib()
        self.device = device
        self.w1 = torch.tensor(w1, device=device, dtype=torch.float32)
        self.w2 = torch.tensor(w2, device=device, dtype=torch.float32)
        self.K = torch.tensor(K, device=device, dtype=torch.int32)
        self.alpha = torch.tensor(alpha, device=device, dtype=torch.int32)
        self.agents = agents
        self.n_players = n_players
        self.n_games = n_games
        self.starting_bbs = starting_bbs
        self.max_bbs = max_bbs
        self.max_players = max_players
        self.active_players = n_players
        self.seed = int(seed)
        self.table_id0 = int(table_id0)
        self.episode = 0
        # kernel variants, switchable per instance (PulsePokerView.flags / NULL cache pointers); the parity suite runs
        # the same roll-outs with each of them off
        self.use_eval_cache = True       # reset fills the evaluation cache, steps read it
        self.obs_staging = True          # observations leave as LDS-staged 16-byte bursts (needs n_games % 16 == 0)
        self.chunked_rollout = True      # rollout(): one launch per chunk of steps instead of one per step
        self.chunk_four_lanes = False    # chunk launches: four lanes per table also where two are the default (<= 10 seats)
        self.paired_launches = True      # rollout_until(): two check intervals per launch under the lag-1 rule
        # opt-in: successive steps write their observation into two alternating buffers, so the tensor a step returned
        # stays intact through the NEXT step (a trainer then needs no copy of the pre-step observation).  Off: the
        # reference's single persistent buffer (PokerGPU.py:633).
        self.double_buffer_obs = False

        self.raise_fractions = torch.tensor([0.25, 0.33, 0.50, 0.75, 1.00, 1.50, 2.00, 3.00, 4.00], device=device)
        self.obs_size = 13 + ((self.max_players - 1) * 3)
        if spaces is not None:
            self.action_space = spaces.Discrete(self.NUM_ACTIONS)
            self.observation_space = spaces.Box(low=0, high=10000, shape=(self.obs_size,), dtype=np.float32)

        self.hand_ranks = handranks.device_table(device)

        N, P = n_games, n_players
        i32 = dict(dtype=torch.int32, device=device)
        for name in _I32_SCALARS:
            setattr(self, name, torch.zeros(N, **i32))
        self.last_raise_size.fill_(1)
        self.is_done = torch.zeros(N, dtype=torch.bool, device=device)
        self._is_done_alt = torch.zeros(N, dtype=torch.bool, device=device)
        self.equity_dirty = torch.ones(N, dtype=torch.bool, device=device)
        self.stacks = torch.full((N, P), starting_bbs, **i32)
        self.current_round_bet = torch.zeros((N, P), **i32)
        self.total_invested = torch.zeros((N, P), **i32)
        self.status = torch.zeros((N, P), **i32)
        self.hands = torch.full((N, P, 2), -1, **i32)
        self.board = torch.full((N, 5), -1, **i32)
        self.decks = torch.zeros((N, 52), **i32)
        self._equities_store = torch.full((N * P,), .5, dtype=torch.float32, device=device)
        self.equities = self._equities_store[:N * P].view(N, P)
        self.obs = torch.zeros((N, self.obs_size), dtype=torch.float32, device=device)
        self._rewards = [torch.zeros(N, dtype=torch.float32, device=device) for _ in range(2)]
        # evaluation cache filled by the reset kernel (include/pulse_env.h: pre_*); internal, not reference state
        self._pre_board = torch.zeros(N, **i32)
        self._pre_hands = torch.zeros((N, P), **i32)
        self._pre_eq = torch.zeros((N, 3, P), dtype=torch.float32, device=device)
        self._pre_rank = torch.zeros((N, P), **i32)

        # constants / scratch names the reference exposes (PokerGPU.py:61-68,138-155)
        self.g = torch.arange(N, device=device)
        self.is_truncated = torch.zeros(N, dtype=torch.bool, device=device)
        self.bb_amounts = torch.ones(N, **i32)
        self.equity_turn_denom = torch.tensor(32765, **i32)
        self.equity_flop_denom = torch.tensor(749420, **i32)
        self.offset_cards = torch.arange(52, **i32)
        self.is_round_over = torch.zeros(N, dtype=torch.bool, device=device)
        self.raise_amounts = torch.zeros(N, **i32)

        self._has_episode = False      # reference: `self.stacks is None` / `hasattr(self, 'button_pos')`
        self._pp = 0
        self._views = [None, None]
        self._types_cache = {}
        self._act_struct = None                      # PulseQNetAct of act_policy_step (filled per call)
        object.__setattr__(self, "_obs_alt", None)
        object.__setattr__(self, "_obs_bufs", None)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in _TRACKED or name == "double_buffer_obs":
            object.__setattr__(self, "_view_dirty", True)

    def set_agents(self, agents):
        self.agents = agents

    # ------------------------------------------------------------------ C-ABI view
    def _as_state(self, name, dtype, shape):
        t = getattr(self, name)
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        if t.dim() == 0 and shape is not None and shape != ():
            t = t.expand(shape).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"PokerGPU.{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        if t is not getattr(self, name):
            object.__setattr__(self, name, t)
        return t

    def _build_views(self):
        N, P, A = self.n_games, self.n_players, int(self.active_players)
        if not (2 <= A <= P):
            raise ValueError(f"active_players must be in [2, {P}], got {A}")
        ptr = {}
        for name in _I32_SCALARS:
            ptr[name] = self._as_state(name, torch.int32, (N,)).data_ptr()
        for name in _ROWS:
            ptr[name] = self._as_state(name, torch.int32, (N, P)).data_ptr()
        ptr["hands"] = self._as_state("hands", torch.int32, (N, P, 2)).data_ptr()
        ptr["board"] = self._as_state("board", torch.int32, (N, 5)).data_ptr()
        ptr["decks"] = self._as_state("decks", torch.int32, (N, 52)).data_ptr()
        ptr["obs"] = self._as_state("obs", torch.float32, (N, self.obs_size)).data_ptr()
        ptr["equity_dirty"] = self._as_state("equity_dirty", torch.bool, (N,)).data_ptr()
        eq = self.equities
        if not (isinstance(eq, torch.Tensor) and eq.dtype == torch.float32 and eq.device == self.device
                and eq.is_contiguous() and tuple(eq.shape) == (N, A)):
            src = torch.as_tensor(eq, dtype=torch.float32, device=self.device)
            new = self._equities_store[:N * A].view(N, A)
            if tuple(src.shape) == (N, A):
                new.copy_(src)
            else:                      # active_players was re-assigned (tests do): keep what overlaps
                new.fill_(.5)
                k = min(A, src.shape[1]) if src.dim() == 2 and src.shape[0] == N else 0
                if k:
                    new[:, :k] = src[:, :k]
            object.__setattr__(self, "equities", new)
        ptr["equities"] = self.equities.data_ptr()
        for name, dt in (("w1", torch.float32), ("w2", torch.float32), ("K", torch.int32), ("alpha", torch.int32)):
            ptr[name] = self._as_state(name, dt, ()).data_ptr()
        use_cache = self.use_eval_cache
        ptr["pre_board"] = self._pre_board.data_ptr() if use_cache else None
        ptr["pre_hands"] = self._pre_hands.data_ptr() if use_cache else None
        ptr["pre_eq"] = self._pre_eq.data_ptr() if use_cache else None
        ptr["pre_rank"] = self._pre_rank.data_ptr() if use_cache else None
        hr = self._as_state("hand_ranks", torch.int32, None)
        cur = self._as_state("is_done", torch.bool, (N,))
        if cur.data_ptr() == self._is_done_alt.data_ptr():
            object.__setattr__(self, "_is_done_alt", torch.zeros_like(cur))
        bufs = (cur, self._is_done_alt)
        cur_obs = self._as_state("obs", torch.float32, (N, self.obs_size))
        if self.double_buffer_obs:
            if self._obs_alt is None or self._obs_alt.shape != cur_obs.shape or self._obs_alt.data_ptr() == cur_obs.data_ptr():
                object.__setattr__(self, "_obs_alt", torch.zeros_like(cur_obs))
            obs_bufs = (cur_obs, self._obs_alt)
        else:
            obs_bufs = (cur_obs, cur_obs)
        views = []
        for src, dst in ((0, 1), (1, 0)):
            v = _native.PokerView()
            v.n_games, v.n_players, v.active_players, v.max_players = N, P, A, self.max_players
            v.obs_size, v.hand_ranks_len = self.obs_size, hr.numel()
            v.flags = ((0 if self.obs_staging else _native.VIEW_NO_OBS_STAGING) | (0 if self.chunked_rollout else _native.VIEW_NO_CHUNK)
                       | (_native.VIEW_FOUR_LANES if self.chunk_four_lanes else 0) | (0 if self.paired_launches else _native.VIEW_NO_PAIRS))
            v.hand_ranks = hr.data_ptr()
            for k, p in ptr.items():
                setattr(v, k, p)
            v.is_done = bufs[src].data_ptr()
            v.is_done_out = bufs[dst].data_ptr()
            v.obs = obs_bufs[dst].data_ptr()
            views.append(v)
        inplace = _native.PokerView.from_buffer_copy(views[0])
        inplace.is_done_out = inplace.is_done
        object.__setattr__(self, "_views", views)
        object.__setattr__(self, "_view_inplace", inplace)
        object.__setattr__(self, "_done_bufs", bufs)
        object.__setattr__(self, "_obs_bufs", obs_bufs)
        object.__setattr__(self, "_pp", 0)
        object.__setattr__(self, "_view_dirty", False)

    def _view(self, inplace=False):
        if self._view_dirty:
            self._build_views()
        if inplace:
            v = self._view_inplace
            v.is_done = v.is_done_out = self.is_done.data_ptr()
            v.obs = self._obs_bufs[self._pp].data_ptr()
            return v
        return self._views[self._pp]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _on_device(self):
        """The C entry points launch on the CURRENT device: make that the device this environment's tensors live on
        (a no-op context in the one-GPU-per-process layout the bench and the trainer use)."""
        if torch.cuda.current_device() == self.device.index:
            return contextlib.nullcontext()
        return torch.cuda.device(self.device)

    def _actions(self, actions):
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int64 and actions.device == self.device
                and actions.is_contiguous()):
            actions = torch.as_tensor(actions, dtype=torch.int64).to(self.device).contiguous()
        if actions.dim() != 1 or actions.shape[0] != self.n_games:
            raise ValueError(f"actions must have shape ({self.n_games},), got {tuple(actions.shape)}")
        return actions

    # ------------------------------------------------------------------ reset (PokerGPU.py:73-157)
    def reset(self, seed=None, options=None, rotation=0):
        super().reset(seed=seed)
        if seed is not None:
            self.seed = int(seed)
        options = options or {}
        ap = options.get('active_players', False)
        if isinstance(ap, (bool, type(None))) or not isinstance(ap, int):
            if ap:      # reference behaviour: sample on the device and sync once per episode (PokerGPU.py:76-77)
                candidate_players = torch.randint(2, self.n_players + 1, (1,), device=self.device).item()
            else:
                candidate_players = self.n_players
        else:           # extension: the caller sampled it (host RNG), no device sync
            if not (2 <= ap <= self.n_players):
                raise ValueError(f"active_players must be in [2, {self.n_players}], got {ap}")
            candidate_players = ap
        q_seat = options.get('q_agent_seat', 0)
        A, N = int(max(candidate_players, q_seat + 1)), self.n_games
        if self._view_dirty or not (2 <= A <= self.n_players):
            self.active_players = A
        else:
            # only the number of seats in the hand changes between episodes: patch the cached C views in place instead
            # of rebuilding (and re-validating) them -- that rebuild was most of the idle time at an episode boundary
            object.__setattr__(self, "active_players", A)
            for v in (*self._views, self._view_inplace):
                v.active_players = A

        prefixed_decks = options.get("prefixed_decks")
        deck_tensor = None
        if prefixed_decks is not None:
            deck_tensor = torch.as_tensor(prefixed_decks, dtype=torch.int32, device=self.device)
            expected_shape = (self.n_games, 52)
            if tuple(deck_tensor.shape) != expected_shape:
                raise ValueError(f"prefixed_decks must have shape {expected_shape}, got {tuple(deck_tensor.shape)}")
            deck_tensor = deck_tensor.contiguous()

        if self._view_dirty:
            self.equities = self._equities_store[:N * A].view(N, A)
        else:       # same storage, same pointer: only the shape the attribute shows changes
            object.__setattr__(self, "equities", self._equities_store[:N * A].view(N, A))
        v = self._view(inplace=True)
        o = self.__dict__.get("_reset_opts")
        if o is None:                                   # one options block per environment, refilled (its construction was a
            o = _native.PokerResetOpts()                # tenth of the host time of a reset)
            object.__setattr__(self, "_reset_opts", o)
        o.stats_rewards = o.stats_out = None
        o.first = 0 if self._has_episode else 1
        o.starting_bbs, o.max_bbs = int(self.starting_bbs), int(self.max_bbs)
        o.rotation = int(options.get('rotation', rotation)) if self._has_episode else 0
        o.seed, o.episode, o.table_id0 = self.seed & (2**64 - 1), self.episode, self.table_id0
        o.prefixed_decks = deck_tensor.data_ptr() if deck_tensor is not None else None
        o.decks_out = self.decks.data_ptr()
        o.shuffle_key_bits = int(getattr(self, "_shuffle_key_bits", 0))      # test hook (pulse_env.h)
        stats = options.get("episode_stats")
        if stats is not None:
            # (rewards fp32[N], acc float64[STATS_SLOTS, STATS_STRIDE] from new_episode_stats()): the sums of the episode that
            # ends with this reset -- acc[:, 0].sum() += sum(rewards), acc[:, 1].sum() += tables done -- taken by the reset
            # launch itself before it clears the flags (no launch of their own)
            rew, out = stats
            if not (rew.dtype == torch.float32 and rew.numel() == self.n_games and rew.is_contiguous() and out.dtype == torch.float64
                    and out.numel() == _native.STATS_SLOTS * _native.STATS_STRIDE and out.is_contiguous()
                    and rew.device == self.device and out.device == self.device):
                raise ValueError("episode_stats must be (fp32[n_games] rewards, the float64 accumulator of new_episode_stats()) on the environment's device")
            o.stats_rewards, o.stats_out = rew.data_ptr(), out.data_ptr()
        with self._on_device():
            _native.check(self._lib.pulse_poker_reset(C.byref(v), C.byref(o), self._stream()), "pulse_poker_reset")
        self.button_pos = self.button[0]
        self._has_episode = True
        self.episode += 1
        return self.obs, self.get_info()

    def new_episode_stats(self):
        """Accumulator for reset(options={"episode_stats": (rewards, acc)}); episode_stats_totals(acc) -> (reward sum, tables done)."""
        return torch.zeros((_native.STATS_SLOTS, _native.STATS_STRIDE), dtype=torch.float64, device=self.device)

    @staticmethod
    def episode_stats_totals(acc):
        return acc[:, :2].sum(dim=0)

    # ------------------------------------------------------------------ step (PokerGPU.py:527-633)
    def step(self, actions):
        actions = self._actions(actions)
        if "calculate_equities" in self.__dict__:
            # a caller wrapped the method (tests/poker/test_poker_gpu_round_progression.py:241-301):
            # keep the reference's host-side `if equity_dirty.any(): self.calculate_equities()`
            if self.equity_dirty.any():
                self.calculate_equities()
        v = self._view()
        rewards = self._rewards[self._pp]
        with self._on_device():
            _native.check(self._lib.pulse_poker_step(C.byref(v), actions.data_ptr(), rewards.data_ptr(), self._stream()),
                          "pulse_poker_step")
        pp = 1 - self._pp
        object.__setattr__(self, "_pp", pp)
        object.__setattr__(self, "is_done", self._done_bufs[pp])
        object.__setattr__(self, "obs", self._obs_bufs[pp])
        return self.obs, rewards, self.is_done, self.is_truncated, self.get_info()

    def policy_step(self, agent_types, actions, step_counter):
        """Scripted-opponent policy (build_actions) fused with step() in one launch.  `agent_types`:
        bytes/uint8 sequence of PULSE_AGENT_* per seat; EXTERNAL seats take their action from `actions`."""
        actions = self._actions(actions)
        key = tuple(int(x) for x in agent_types)
        types = self._types_cache.get(key)
        if types is None:
            if len(key) != self.n_players:
                raise ValueError(f"agent_types must have {self.n_players} entries, got {len(key)}")
            types = self._types_cache[key] = (C.c_uint8 * self.n_players)(*key)
        v = self._view()
        rewards = self._rewards[self._pp]
        with self._on_device():
            _native.check(self._lib.pulse_poker_policy_step(C.byref(v), types, self.seed & (2**64 - 1), int(step_counter),
                                                            self.table_id0, actions.data_ptr(), rewards.data_ptr(),
                                                            self._stream()), "pulse_poker_policy_step")
        pp = 1 - self._pp
        object.__setattr__(self, "_pp", pp)
        object.__setattr__(self, "is_done", self._done_bufs[pp])
        object.__setattr__(self, "obs", self._obs_bufs[pp])
        return self.obs, rewards, self.is_done, self.is_truncated, self.get_info()

    def act_policy_step(self, learner, q_seat, agent_types, actions, step_counter, states, seat_idx, terminated, row_mask_out, stop_rule=None):
        """`learner.act_into(states, seat_idx, q_seat, actions, step_counter=step_counter, terminated=terminated,
        row_mask_out=row_mask_out, select_for_training=True)` followed by `policy_step(agent_types, actions, step_counter)` as
        ONE launch (pulse_poker_act_policy_step: the workgroup that picks the learner's actions of a window of 128 tables steps
        those tables itself) -- the same results word for word.  `states` must not be the buffer this step writes its
        observation into (double_buffer_obs).  `stop_rule`: the done tables after the step are counted for it by the launch
        (one check point, as rollout(..., stop_rule=)).  Falls back to the two calls where the fused launch does not apply
        (a table count that is no multiple of 128, more than ten seats, rows the kernel's 16-byte loads cannot take)."""
        actions = self._actions(actions)
        n = self.n_games
        fits = (n % 128 == 0 and self.max_players <= 10 and not self.chunk_four_lanes and self.double_buffer_obs and states.is_cuda
                and states.dtype == torch.float32 and states.dim() == 2 and states.stride(1) == 1 and states.stride(0) % 4 == 0
                and states.data_ptr() % 16 == 0 and 16 <= learner.state_dim <= 40 and learner.state_dim % 8 == 0
                and seat_idx.dtype == torch.int32 and seat_idx.is_contiguous() and row_mask_out is not None)
        if not fits:
            learner.act_into(states, seat_idx, q_seat, actions, step_counter=step_counter, terminated=terminated, row_mask_out=row_mask_out,
                             select_for_training=True)
            out = self.policy_step(agent_types, actions, step_counter)
            if stop_rule is not None:
                stop_rule.submit(terminated | out[2] if terminated is not None else out[2])
            return out
        key = tuple(int(x) for x in agent_types)
        types = self._types_cache.get(key)
        if types is None:
            if len(key) != self.n_players:
                raise ValueError(f"agent_types must have {self.n_players} entries, got {len(key)}")
            types = self._types_cache[key] = (C.c_uint8 * self.n_players)(*key)
        v = self._view()
        if v.obs == states.data_ptr():
            raise ValueError("act_policy_step: `states` is the buffer this step writes its observation into")
        rewards = self._rewards[self._pp]
        net, scratch = learner.begin_fused_act(states, row_mask_out, n)
        act = self._act_struct
        if act is None:
            act = self._act_struct = _native.QNetAct()
        act.states, act.row_stride, act.seat_idx, act.q_seat = states.data_ptr(), states.stride(0), seat_idx.data_ptr(), int(q_seat)
        act.epsilon, act.seed, act.step, act.table_id0 = float(learner.epsilon), learner.seed & (2**64 - 1), int(step_counter), learner.table_id0
        act.terminated = None if terminated is None else terminated.data_ptr()
        act.row_mask_out, act.select_scratch, act.select_words = row_mask_out.data_ptr(), scratch.data_ptr(), scratch.numel()
        with self._on_device():
            _native.check(self._lib.pulse_poker_act_policy_step(
                C.byref(v), types, self.seed & (2**64 - 1), int(step_counter), self.table_id0, actions.data_ptr(), rewards.data_ptr(),
                C.byref(net), C.byref(act), None if stop_rule is None else stop_rule.handle, self._stream()), "pulse_poker_act_policy_step")
        pp = 1 - self._pp
        object.__setattr__(self, "_pp", pp)
        object.__setattr__(self, "is_done", self._done_bufs[pp])
        object.__setattr__(self, "obs", self._obs_bufs[pp])
        return self.obs, rewards, self.is_done, self.is_truncated, self.get_info()

    def rollout(self, agent_types, actions, n_steps, step_counter0, timer=None, stop_rule=None):
        """`n_steps` fused policy+step transitions in ONE native call and (chunked_rollout) ONE launch: the
        tables' state stays in registers between the steps, every step still stores its observation, reward,
        done flag and action.  Memory afterwards is bit-identical to calling
        policy_step(agent_types, actions, step_counter0 + i) for i in range(n_steps); returns what the last step
        returned.  `stop_rule` (stoprule.LaggedDoneCount): the done-count of the final state is submitted as one
        check point by the same call; `timer` (stoprule.RolloutTimer): the call is bracketed by HIP events."""
        actions = self._actions(actions)
        key = tuple(int(x) for x in agent_types)
        types = self._types_cache.get(key)
        if types is None:
            if len(key) != self.n_players:
                raise ValueError(f"agent_types must have {self.n_players} entries, got {len(key)}")
            types = self._types_cache[key] = (C.c_uint8 * self.n_players)(*key)
        if self._view_dirty:
            self._build_views()
        pp = self._pp
        with self._on_device():
            _native.check(self._lib.pulse_poker_rollout(C.byref(self._views[pp]), C.byref(self._views[1 - pp]), types,
                                                        self.seed & (2**64 - 1), int(step_counter0), self.table_id0,
                                                        actions.data_ptr(), self._rewards[pp].data_ptr(),
                                                        self._rewards[1 - pp].data_ptr(), int(n_steps),
                                                        None if timer is None else timer.handle,
                                                        None if stop_rule is None else stop_rule.handle,
                                                        self._stream()), "pulse_poker_rollout")
        if n_steps <= 0:
            return self.obs, self._rewards[pp], self.is_done, self.is_truncated, self.get_info()
        last = pp if (n_steps - 1) % 2 == 0 else 1 - pp
        new_pp = pp if n_steps % 2 == 0 else 1 - pp
        object.__setattr__(self, "_pp", new_pp)
        object.__setattr__(self, "is_done", self._done_bufs[new_pp])
        object.__setattr__(self, "obs", self._obs_bufs[new_pp])
        return self.obs, self._rewards[last], self.is_done, self.is_truncated, self.get_info()

    def rollout_until(self, agent_types, actions, chunk_steps, max_steps, step_counter0, stop_rule, timer=None, time_every=0):
        """The trainer's inner loop for scripted tables without the interpreter in it (pulse_poker_rollout_until): chunks
        of `chunk_steps` fused policy+step transitions, the stop rule's verdict after each, until it ends the episode
        or `max_steps` steps have run.  Returns (steps_done, over).  `stop_rule` must decide natively
        (stoprule.LaggedDoneCount with exchange "local" or "rccl"); with a host-side exchange call rollout() + over()
        per chunk instead."""
        if stop_rule is None or getattr(stop_rule, "handle", None) is None or stop_rule.exchange == "host":
            raise ValueError("rollout_until needs a stop rule that decides natively (exchange 'local' or 'rccl')")
        actions = self._actions(actions)
        key = tuple(int(x) for x in agent_types)
        types = self._types_cache.get(key)
        if types is None:
            if len(key) != self.n_players:
                raise ValueError(f"agent_types must have {self.n_players} entries, got {len(key)}")
            types = self._types_cache[key] = (C.c_uint8 * self.n_players)(*key)
        if self._view_dirty:
            self._build_views()
        pp = self._pp
        done, over = C.c_int32(0), C.c_int32(0)
        with self._on_device():
            _native.check(self._lib.pulse_poker_rollout_until(
                C.byref(self._views[pp]), C.byref(self._views[1 - pp]), types, self.seed & (2**64 - 1), int(step_counter0), self.table_id0,
                actions.data_ptr(), self._rewards[pp].data_ptr(), self._rewards[1 - pp].data_ptr(), int(chunk_steps), int(max_steps),
                None if timer is None else timer.handle, int(time_every), stop_rule.handle, self._stream(), C.byref(done), C.byref(over)),
                "pulse_poker_rollout_until")
        stop_rule.decisions += -(-done.value // max(int(chunk_steps), 1))
        if done.value % 2:
            new_pp = 1 - pp
            object.__setattr__(self, "_pp", new_pp)
            object.__setattr__(self, "is_done", self._done_bufs[new_pp])
            object.__setattr__(self, "obs", self._obs_bufs[new_pp])
        return done.value, bool(over.value)

    # ------------------------------------------------------------------ white-box methods
    def _phases(self, phases, actions=None, actor_idx=None, rewards=None):
        v = self._view(inplace=True)
        # converted copies must stay referenced until the launch is enqueued (the caching allocator would
        # otherwise hand their memory to the next conversion)
        act_t = self._actions(actions) if actions is not None else None
        idx_t = None
        if actor_idx is not None:
            idx_t = torch.as_tensor(actor_idx).to(device=self.device, dtype=torch.int32).contiguous()
        with self._on_device():
            _native.check(self._lib.pulse_poker_phases(C.byref(v), phases, act_t.data_ptr() if act_t is not None else None,
                                                       idx_t.data_ptr() if idx_t is not None else None,
                                                       rewards.data_ptr() if rewards is not None else None, self._stream()),
                          "pulse_poker_phases")
        del act_t, idx_t

    def get_obs(self):                               # PokerGPU.py:159-179
        self._phases(_native.PH_OBS)
        return self.obs

    def get_info(self):                              # PokerGPU.py:181-186
        return {'active_players': self.active_players, 'stacks': self.stacks, 'seat_idx': self.idx}

    def calculate_equities(self):                    # PokerGPU.py:455-525
        self._phases(_native.PH_EQUITY)

    def execute_actions(self, actions):              # PokerGPU.py:230-303
        self._phases(_native.PH_EXECUTE, actions=actions)

    def resolve_fold_winners(self):                  # PokerGPU.py:331-338
        self._phases(_native.PH_FOLDWIN)

    def resolve_terminated_games(self):              # PokerGPU.py:380-453
        self._phases(_native.PH_SHOWDOWN)

    def poker_reward_gpu(self, actions, actor_idx):  # PokerGPU.py:305-329
        rewards = torch.empty(self.n_games, dtype=torch.float32, device=self.device)
        self._phases(_native.PH_REWARD, actions=actions, actor_idx=actor_idx, rewards=rewards)
        return rewards

    # cold helpers kept for API compatibility; the fused kernels deal and post blinds themselves
    def post_blinds(self):                           # PokerGPU.py:188-199
        g, bb = self.g, self.bb.long()
        self.stacks[g, bb] -= self.bb_amounts
        self.current_round_bet[g, bb] += self.bb_amounts
        self.total_invested[g, bb] += self.bb_amounts
        self.pots += self.bb_amounts
        self.status[g, bb] = torch.where(self.stacks[g, bb] == 0, self.ALLIN, self.ACTIVE).to(torch.int32)

    def deal_players_cards(self, n_cards):           # PokerGPU.py:201-206
        card_idx = self.deck_positions.unsqueeze(1) + self.offset_cards[:n_cards].unsqueeze(0)
        cards = self.decks[self.g.unsqueeze(1), card_idx.long()]
        self.deck_positions += n_cards
        return cards

    def deal_cards(self, g, n_cards):                # PokerGPU.py:208-214
        card_idx = self.deck_positions[g].unsqueeze(1) + self.offset_cards[:n_cards].unsqueeze(0)
        cards = self.decks[g.unsqueeze(1), card_idx.long()]
        self.deck_positions[g] += n_cards
        return cards.to(torch.int32)

    def close(self):
        return None
