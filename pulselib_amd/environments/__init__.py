"""Environment registry: same ids as the reference's environments/__init__.py:3-31 when gymnasium
is installed; importable (and usable through the classes directly) when it is not."""
try:
    from gymnasium.envs.registration import register

    _IDS = {
        'Pulse-Poker-GPU-v1': ('pulselib_amd.environments.Poker:PokerGPU', 200000),
        'Pulse-Blackjack-Standard': ('pulselib_amd.environments.blackjack.blackjack:BlackJack', 100000),
        'Pulse-Particle-2d': ('pulselib_amd.environments.Particle2D.Particle2D:Particle2D', 100000),
        'Pulse-2048-v2': ('pulselib_amd.environments.TFE:TFE', 200000),
    }
    for _id, (_entry, _steps) in _IDS.items():
        try:
            register(id=_id, entry_point=_entry, max_episode_steps=_steps)
        except Exception:   # already registered (e.g. by the reference package)
            pass
except ImportError:   # gymnasium absent: classes are still importable directly
    pass
