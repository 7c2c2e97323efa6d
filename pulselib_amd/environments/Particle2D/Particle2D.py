"""MI355X drop-in for environments/Particle2D/Particle2D.py (cited as Particle2D.py:line): the eight
eager ops of step() are one HIP launch (csrc/envs.hip: particle2d_step_kernel), fp32 with torch's
op order (no FMA contraction)."""
from __future__ import annotations

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


class Particle2D(_EnvBase):
    metadata = {'render.modes': ['human']}

    def __init__(self, device, batch_size, dt=0.1, max_steps=200):
        super().__init__()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"pulselib_amd.Particle2D runs on an MI355X ('cuda' device); got '{device}'. No CPU fallback.")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.lib()
        self.device, self.batch_size, self.dt, self.max_steps = device, batch_size, dt, max_steps
        if spaces is not None:
            self.action_space = spaces.Box(-1, 1, (2,), np.float32)
            self.observation_space = spaces.Box(-np.inf, np.inf, (4,), np.float32)
        self.state = torch.zeros((batch_size, 4), device=device)
        self.terminated = torch.zeros(batch_size, device=device, dtype=torch.bool)
        self.steps = torch.zeros(batch_size, device=device, dtype=torch.int32)
        self._truncated = torch.zeros(batch_size, device=device, dtype=torch.bool)     # constant (Particle2D.py:30)
        # step() returns fresh tensors, as the reference does (state.clone(), Particle2D.py:26-30): a caller may keep a
        # trajectory of them.  `reuse_outputs = True` (opt-in, like PokerGPU.double_buffer_obs) alternates two persistent
        # output sets instead: what a step returned then stays intact through the NEXT step only, no allocation per call.
        self.reuse_outputs = False
        self._out = None
        self._pp = 0

    def _new_outputs(self):
        return (torch.empty((self.batch_size, 4), device=self.device), torch.empty(self.batch_size, device=self.device),
                torch.empty(self.batch_size, device=self.device, dtype=torch.bool))

    def reset(self, seed=None, options=None):                       # Particle2D.py:15-20
        if seed is not None:
            torch.manual_seed(seed)
        if options and options.get("state") is not None:
            self.state = torch.as_tensor(options["state"], dtype=torch.float32).to(self.device).contiguous().clone()
        else:
            self.state = torch.cat([torch.randn(self.batch_size, 2, device=self.device) * 5,
                                    torch.zeros(self.batch_size, 2, device=self.device)], dim=1)
        self.terminated = torch.zeros(self.batch_size, device=self.device, dtype=torch.bool)
        self.steps = torch.zeros(self.batch_size, device=self.device, dtype=torch.int32)
        return self.state.clone(), {}

    def step(self, action):                                         # Particle2D.py:22-30
        if not (isinstance(action, torch.Tensor) and action.dtype == torch.float32 and action.device == self.device and action.is_contiguous()):
            action = torch.as_tensor(action, dtype=torch.float32).to(self.device).contiguous()
        assert action.shape == (self.batch_size, 2)
        if self.reuse_outputs:
            if self._out is None:
                self._out = [self._new_outputs() for _ in range(2)]
            obs, rewards, terminated = self._out[self._pp]
            self._pp ^= 1
        else:
            obs, rewards, terminated = self._new_outputs()          # (torch's caching allocator: no hipMalloc per call)
        _native.check(self._lib.pulse_particle2d_step(self.state.data_ptr(), action.data_ptr(), self.steps.data_ptr(),
                                                      obs.data_ptr(), rewards.data_ptr(), terminated.data_ptr(),
                                                      self.batch_size, float(self.dt), int(self.max_steps),
                                                      torch.cuda.current_stream(self.device).cuda_stream), "pulse_particle2d_step")
        self.terminated = terminated
        return obs, rewards, self.terminated, self._truncated, {}
