"""Time pulse_poker_reset variants on the GPU (events over repeated launches): device shuffle vs prefixed
decks, evaluation cache on/off.  Usage: python tools/time_reset.py [n_tables]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    from pulselib_amd.environments.Poker import PokerGPU
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dev = torch.device("cuda:0")
    for cache in ((True,) if "cache-on-only" in sys.argv else (True, False)):
        env = PokerGPU(device=dev, agents=[], n_players=10, max_players=10, n_games=N, seed=3)
        env.use_eval_cache = cache
        env.reset(options={"active_players": 6})
        decks = env.decks.clone()
        for A in (10, 6, 2):
            t_sh = timed(lambda: env.reset(options={"active_players": A}))
            t_pf = timed(lambda: env.reset(options={"active_players": A, "prefixed_decks": decks}))
            wrote = N * (208 + 4 * 4 * 10 + 80 + 4 * A + 20 + 18 * 4 - 6 + 160 + (40 + 4 + 120 + 40 if cache else 0))     # bytes the launch stores
            print(f"N={N} cache={int(cache)} A={A:2d}  shuffle {t_sh:7.1f} us ({wrote / t_sh / 1e6:5.2f} TB/s written)   prefixed {t_pf:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
