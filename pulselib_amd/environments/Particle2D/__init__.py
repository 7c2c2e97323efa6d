from .Particle2D import Particle2D

__all__ = ["Particle2D"]
