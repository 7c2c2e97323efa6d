from .blackjack import BlackJack

__all__ = ["BlackJack"]
