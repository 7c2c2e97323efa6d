from .PokerGPU import PokerGPU
from .Player import (HeuristicHandsPlayerGPU, LoosePassivePlayerGPU, Player, PokerQNetwork, RandomPlayer,
                     SmallBallPlayerGPU, TightAggressivePlayerGPU)
from .utils import PokerAgentType, build_actions, get_rotated_agents, load_gpu_agents

__all__ = ["PokerGPU", "Player", "RandomPlayer", "HeuristicHandsPlayerGPU", "TightAggressivePlayerGPU",
           "LoosePassivePlayerGPU", "SmallBallPlayerGPU", "PokerQNetwork", "PokerAgentType", "build_actions", "get_rotated_agents",
           "load_gpu_agents"]
