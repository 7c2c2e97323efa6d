from .qlearning import QLearningBatch

__all__ = ["QLearningBatch"]
