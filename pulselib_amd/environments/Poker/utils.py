"""Rollout glue of the Poker GPU trainer: same names and signatures as the reference's
environments/Poker/utils.py (:80-87 PokerAgentType, :108-123 build_actions, :125-157
load_gpu_agents, :173-183 get_rotated_agents), without its eval7 import.

`build_actions` serves every scripted seat type with ONE launch of the policy kernel
(pulse_poker_policy) instead of one boolean-mask round trip per type; a Q-network seat is served by the
agent's fused `act_into` (qnetwork.py) when it has one, else by the reference's masked call into the agent."""
from __future__ import annotations

import ctypes as C
import enum
import itertools

import torch

from ... import _native


class PokerAgentType(enum.Enum):
    QLEARNING = 'qlearning'
    HEURISTIC = "heuristic"
    RANDOM = 'random'
    HEURISTIC_HANDS = 'heuristic_hands'
    TIGHT_AGGRESSIVE = "tight_aggressive"
    LOOSE_PASSIVE = "loose_passive"
    SMALL_BALL = "small_ball"


NATIVE_TYPE = {
    PokerAgentType.QLEARNING: _native.AGENT_EXTERNAL,
    PokerAgentType.RANDOM: _native.AGENT_RANDOM,
    PokerAgentType.HEURISTIC_HANDS: _native.AGENT_HEURISTIC_HANDS,
    PokerAgentType.TIGHT_AGGRESSIVE: _native.AGENT_TIGHT_AGGRESSIVE,
    PokerAgentType.LOOSE_PASSIVE: _native.AGENT_LOOSE_PASSIVE,
    PokerAgentType.SMALL_BALL: _native.AGENT_SMALL_BALL,
}

_policy_seed = 0x5EED
_policy_counter = itertools.count(1)


def set_policy_seed(seed: int) -> None:
    """Seed of the Philox stream behind the scripted opponents' random picks."""
    global _policy_seed, _policy_counter
    _policy_seed = int(seed)
    _policy_counter = itertools.count(1)


def native_types(agent_types) -> list[int]:
    out = []
    for t in agent_types:
        if t not in NATIVE_TYPE:
            raise ValueError(f"agent type {t} has no batched GPU policy")
        out.append(NATIVE_TYPE[t])
    return out


def launch_policy(state, actions, curr_players, native, table_id0=0, step_counter=None):
    if not (state.is_cuda and state.dtype == torch.float32 and state.stride(-1) == 1):
        state = state.to(dtype=torch.float32).contiguous()
    curr = curr_players if (curr_players.dtype == torch.int32 and curr_players.is_contiguous()) else curr_players.to(torch.int32).contiguous()
    if actions.dtype != torch.int64 or not actions.is_contiguous():
        raise ValueError("actions must be a contiguous int64 tensor (it is written in place)")
    n = state.shape[0]
    types = (C.c_uint8 * len(native))(*native)
    counter = next(_policy_counter) if step_counter is None else int(step_counter)
    lib = _native.lib()
    _native.check(lib.pulse_poker_policy(state.data_ptr(), state.stride(0), curr.data_ptr(), n, types, len(native),
                                         _policy_seed, counter, int(table_id0), actions.data_ptr(),
                                         torch.cuda.current_stream(state.device).cuda_stream), "pulse_poker_policy")


def build_actions(state, actions, curr_players, agents, agent_types, device, epsilon=0.1):
    """environments/Poker/utils.py:108-123 -- fills `actions` in place for every table."""
    native = [NATIVE_TYPE.get(t, _native.AGENT_EXTERNAL) for t in agent_types]
    if any(n != _native.AGENT_EXTERNAL for n in native):
        launch_policy(state, actions, curr_players, native)
    # seats without a batched policy (the Q-network) keep the reference's masked call, first agent of the type acts
    grouped = {}
    for agent_idx, agent_type in enumerate(agent_types):
        if NATIVE_TYPE.get(agent_type, _native.AGENT_EXTERNAL) == _native.AGENT_EXTERNAL:
            grouped.setdefault(agent_type, []).append(agent_idx)
    for agent_type, seat_indices in grouped.items():
        agent = agents[seat_indices[0]]
        if agent_type == PokerAgentType.QLEARNING and hasattr(agent, "act_into") and state.is_cuda:
            for seat_idx in seat_indices:          # one fused launch per learner seat: no mask gather, no sync
                agent.act_into(state, curr_players, seat_idx, actions)
            continue
        mask = torch.zeros_like(curr_players, dtype=torch.bool)
        for seat_idx in seat_indices:
            mask |= curr_players == seat_idx
        agent = agents[seat_indices[0]]
        if agent_type == PokerAgentType.QLEARNING:
            actions[mask] = agent.get_actions(state[mask])
        else:
            actions[mask] = agent.action(state[mask])


def load_gpu_agents(device, num_players: int, agent_types: list, starting_stack: int, action_space_n: int):
    """environments/Poker/utils.py:125-157"""
    from .Player import (HeuristicHandsPlayerGPU, LoosePassivePlayerGPU, RandomPlayer, SmallBallPlayerGPU,
                         TightAggressivePlayerGPU)
    players, types = [], []
    assert len(agent_types) == num_players
    table = {
        'random': (RandomPlayer, PokerAgentType.RANDOM),
        'heuristic_hands': (HeuristicHandsPlayerGPU, PokerAgentType.HEURISTIC_HANDS),
        'tight_aggressive': (TightAggressivePlayerGPU, PokerAgentType.TIGHT_AGGRESSIVE),
        'loose_passive': (LoosePassivePlayerGPU, PokerAgentType.LOOSE_PASSIVE),
        'small_ball': (SmallBallPlayerGPU, PokerAgentType.SMALL_BALL),
    }
    for i, a_type in enumerate(agent_types):
        if a_type not in table:
            raise ValueError(f"Unknown agent type: {a_type}")
        cls, agent_type = table[a_type]
        players.append(cls(starting_stack, i) if cls is RandomPlayer else cls(starting_stack, i, device))
        types.append(agent_type)
    return players, types


def get_rotated_agents(agents, agent_types, episode_idx=None, q_agent_idx=None):
    """environments/Poker/utils.py:173-183"""
    n = len(agents)
    q_idx = q_agent_idx if q_agent_idx is not None else agent_types.index(PokerAgentType.QLEARNING)
    target_seat = (episode_idx % n) if episode_idx is not None else 0
    rotation = (target_seat - q_idx) % n
    rotated_agents = agents[-rotation:] + agents[:-rotation]
    rotated_types = agent_types[-rotation:] + agent_types[:-rotation]
    return rotated_agents, rotated_types, target_seat, rotation
