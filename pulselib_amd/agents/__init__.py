from .first_visit_mc import FirstVisitMonteCarlo
from .qlearning import QLearningBatch

__all__ = ["FirstVisitMonteCarlo", "QLearningBatch"]
