"""MI355X drop-in for the reference's batched blackjack, environments/blackjack/blackjack.py
(cited as blackjack.py:line).  reset and step are one HIP launch each (csrc/envs.hip); the dealer's
`while active_dealers.any()` loop -- one host sync per drawn card in the reference -- runs inside
the kernel.  Same attribute names, dtypes and return tuple (truncated is None, blackjack.py:186).
`decks` is int32 (reference: int64 from argsort); pass options={"decks": tensor[B,52]} to inject decks."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _native

try:
    import gymnasium as gym
    from gymnasium import spaces
    _EnvBase = gym.Env
except Exception:  # pragma: no cover
    spaces = None

    class _EnvBase:
        metadata: dict = {}

        def reset(self, seed=None, options=None):
            return None


class BlackJack(_EnvBase):
    metadata = {'render.modes': ['human']}
    NUM_ACTIONS = 2  # hit, stand
    WIN_REWARD, LOSS_REWARD = 1, -1

    def __init__(self, device, batch_size, seed=0):
        super().__init__()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"pulselib_amd.BlackJack runs on an MI355X ('cuda' device); got '{device}'. No CPU fallback.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.lib()
        self.device = device
        self.batch_size = batch_size
        self.obs_size = 3
        if spaces is not None:
            self.action_space = spaces.Discrete(self.NUM_ACTIONS)
            self.observation_space = spaces.Box(low=0, high=10000, shape=(self.obs_size,), dtype=np.float32)
        self.g = torch.arange(batch_size, device=device, dtype=torch.int32)
        self.seed, self.episode = int(seed), 0
        B, i32 = batch_size, dict(dtype=torch.int32, device=device)
        self.decks = torch.zeros((B, 52), **i32)
        for name in ("deck_positions", "players_card_idx", "player_card_sums", "dealer_card_idx", "dealer_upcard",
                     "dealer_card_sums", "rewards"):
            setattr(self, name, torch.zeros(B, **i32))
        self.players_cards = torch.zeros((B, 20), **i32)
        self.dealer_cards = torch.zeros((B, 20), **i32)
        for name in ("terminated", "has_ace", "dealer_has_ace"):
            setattr(self, name, torch.zeros(B, dtype=torch.bool, device=device))
        self.obs = torch.zeros((B, 3), **i32)

    def _view(self):
        v = _native.BlackjackView()
        v.batch_size = self.batch_size
        for name in ("decks", "deck_positions", "players_cards", "players_card_idx", "player_card_sums", "dealer_cards",
                     "dealer_card_idx", "dealer_upcard", "dealer_card_sums", "terminated", "has_ace", "dealer_has_ace",
                     "rewards", "obs"):
            t = getattr(self, name)
            assert t.is_contiguous() and t.device == self.device, name
            setattr(v, name, t.data_ptr())
        return v

    def reset(self, seed=None, options=None):                       # blackjack.py:23-48
        if seed is not None:
            self.seed = int(seed)
        src = None
        if options and options.get("decks") is not None:
            src = torch.as_tensor(options["decks"]).to(device=self.device, dtype=torch.int32).contiguous()
            if tuple(src.shape) != (self.batch_size, 52):
                raise ValueError(f"decks must have shape {(self.batch_size, 52)}, got {tuple(src.shape)}")
        v = self._view()
        _native.check(self._lib.pulse_blackjack_reset(C.byref(v), src.data_ptr() if src is not None else None,
                                                      self.decks.data_ptr(), self.seed, self.episode,
                                                      torch.cuda.current_stream(self.device).cuda_stream), "pulse_blackjack_reset")
        self.episode += 1
        return self.obs, self.get_info()

    def get_obs(self):                                              # blackjack.py:103-108 (kept in sync by the kernels)
        return self.obs

    def get_info(self):
        return {}

    def step(self, actions):                                        # blackjack.py:179-186
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int64 and actions.device == self.device
                and actions.is_contiguous()):
            actions = torch.as_tensor(actions, dtype=torch.int64).to(self.device).contiguous()
        assert actions.shape == (self.batch_size,)
        v = self._view()
        _native.check(self._lib.pulse_blackjack_step(C.byref(v), actions.data_ptr(),
                                                     torch.cuda.current_stream(self.device).cuda_stream), "pulse_blackjack_step")
        return self.obs, self.rewards, self.terminated, None, self.get_info()
