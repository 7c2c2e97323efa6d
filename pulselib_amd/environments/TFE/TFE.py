"""MI355X 2048: the reference's environments/2048/TFE.py (cited as TFE.py:line) steps ONE board per
env object through numba; here `TFEBatch` steps B boards per HIP launch (one lane per board; 4 x 4 boards -- the
size config/tfe.yaml runs -- packed to 64 bits with the move as four row-table lookups, csrc/envs.hip: tfe_step4_kernel;
3 x 3 and 5 x 5 in registers; any other side from 2 to 8 by a plain per-lane loop) and `TFE` keeps the reference's
single-board constructor / step signature on top of it (a batch of one).

Tile spawns use Philox4x32-10(seed, board id, step counter) instead of numba's `random` (TFE.py:17-34):
same distribution (uniform empty cell, 4 with probability 0.1), reproducible across CPU oracle and GPU."""
from __future__ import annotations

import numpy as np
import torch

from ... import _native

try:
    import gymnasium as gym
    _EnvBase = gym.Env
except Exception:  # pragma: no cover
    gym = None

    class _EnvBase:
        def reset(self, seed=None, options=None):
            return None


class TFEBatch:
    """B independent n x n boards (n in 2..8) on the GPU."""

    def __init__(self, device, batch_size, board_size=4, seed=0, board_id0=0):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"pulselib_amd.TFEBatch runs on an MI355X ('cuda' device); got '{device}'. No CPU fallback.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if not 2 <= board_size <= 8:
            raise ValueError("board_size must be 2..8")
        self._lib = _native.lib()
        self.device, self.batch_size, self.n = device, batch_size, board_size
        self.seed, self.board_id0 = int(seed), int(board_id0)
        self.boards = torch.zeros((batch_size, board_size, board_size), dtype=torch.int32, device=device)
        self.total_score = torch.zeros(batch_size, dtype=torch.int64, device=device)
        self.rewards = torch.zeros(batch_size, dtype=torch.int32, device=device)
        self.dones = torch.zeros(batch_size, dtype=torch.bool, device=device)
        self.truncated = torch.zeros(batch_size, dtype=torch.bool, device=device)      # constant: never truncates (TFE.py:189)
        self.step_counter = 0

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, seed=None, options=None):                       # TFE.py:143-149
        if seed is not None:
            self.seed = int(seed)
        self.step_counter = 0
        _native.check(self._lib.pulse_tfe_reset(self.boards.data_ptr(), self.total_score.data_ptr(), self.batch_size, self.n,
                                                self.seed, self.board_id0, self._stream()), "pulse_tfe_reset")
        self.dones.zero_()
        return self.boards, {"score": self.total_score}

    def step(self, actions):                                        # TFE.py:152-189
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int64 and actions.device == self.device
                and actions.is_contiguous()):
            actions = torch.as_tensor(actions, dtype=torch.int64).to(self.device).contiguous()
        assert actions.shape == (self.batch_size,)
        self.step_counter += 1
        _native.check(self._lib.pulse_tfe_step(self.boards.data_ptr(), self.total_score.data_ptr(), actions.data_ptr(),
                                               self.rewards.data_ptr(), self.dones.data_ptr(), self.batch_size, self.n,
                                               self.seed, self.board_id0, self.step_counter, self._stream()), "pulse_tfe_step")
        return self.boards, self.rewards, self.dones, self.truncated, {"score": self.total_score}


class TFE(_EnvBase):
    """Reference-compatible single-board env (TFE.py:112-189) backed by a TFEBatch of one."""

    def __init__(self, board_height, board_width, device="cuda", seed=0):
        if board_height != board_width:
            raise ValueError("only square boards are supported (the reference's rotate buffer is square too, TFE.py:38-44)")
        self.n, self.m = board_height, board_width
        if gym is not None:
            self.action_space = gym.spaces.Discrete(4)
            self.observation_space = gym.spaces.Box(low=0, high=np.inf, shape=(self.n, self.m), dtype=np.int32)
        self._batch = TFEBatch(device, 1, board_height, seed=seed)
        self.board = np.zeros((self.n, self.m), dtype=np.int32)
        self.total_score = 0
        self.render_mode = 'human'

    def get_obs(self):
        return self.board

    def get_info(self):
        return {'score': self.total_score}

    def is_game_over(self):
        return bool(self._batch.dones[0].item())

    def reset(self, seed=None, options=None):
        boards, _ = self._batch.reset(seed=seed)
        self.board = boards[0].cpu().numpy()
        self.total_score = 0
        return self.get_obs(), self.get_info()

    def step(self, action: int):
        boards, rewards, dones, _, info = self._batch.step(torch.tensor([int(action)], dtype=torch.long))
        self.board = boards[0].cpu().numpy()
        self.total_score = int(info["score"][0].item())
        return self.get_obs(), int(rewards[0].item()), bool(dones[0].item()), False, self.get_info()
