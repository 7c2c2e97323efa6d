"""Scripted batched opponents: same class names and `.action(states)` contract as the reference's
environments/Poker/Player.py:79-176.  Each call is one launch of the policy kernel
(csrc/poker.hip: scripted_action) over the whole batch -- fold / call / raise masks are the
reference's integer rules bit for bit; the random raise pick comes from Philox instead of the
torch generator (same distribution)."""
from __future__ import annotations

import random

import torch

from ... import _native
from . import utils as _utils
from .qnetwork import PokerQNetwork  # noqa: F401  (environments/Poker/Player.py:178 lives here in the reference)


class Player:
    """environments/Poker/Player.py:14-41 (bookkeeping fields only)"""

    def __init__(self, stack_size: int, player_id: int):
        self.id = player_id
        self.stack = stack_size
        self.current_round_bet = 0
        self.total_invested = 0
        self.status = 'active'
        self.hand = []

    def action(self, state):
        raise NotImplementedError

    def learn(self, *a, **k):
        pass

    def reset_state(self, new_hand, starting_stack=None):
        self.hand = new_hand
        self.current_round_bet = 0
        self.total_invested = 0
        self.status = 'active'
        if starting_stack is not None:
            self.stack = starting_stack


class RandomPlayer(Player):
    """environments/Poker/Player.py:43-45"""

    def action(self, state, valid_actions=None):
        return random.randint(0, 12)


class _ScriptedGPU(Player):
    NATIVE = _native.AGENT_EXTERNAL

    def __init__(self, starting_stack: int, player_id: int, device):
        super().__init__(starting_stack, player_id)
        self.device = device
        self.raise_distribution = torch.arange(2, 11, device=device)

    def action(self, states):
        n = states.shape[0]
        actions = torch.zeros(n, dtype=torch.long, device=states.device)
        seat = torch.zeros(n, dtype=torch.int32, device=states.device)
        _utils.launch_policy(states, actions, seat, [self.NATIVE])
        return actions


class HeuristicHandsPlayerGPU(_ScriptedGPU):   # Player.py:79-104
    NATIVE = _native.AGENT_HEURISTIC_HANDS


class TightAggressivePlayerGPU(_ScriptedGPU):  # Player.py:106-126
    NATIVE = _native.AGENT_TIGHT_AGGRESSIVE


class LoosePassivePlayerGPU(_ScriptedGPU):     # Player.py:128-151
    NATIVE = _native.AGENT_LOOSE_PASSIVE


class SmallBallPlayerGPU(_ScriptedGPU):        # Player.py:153-176
    NATIVE = _native.AGENT_SMALL_BALL
