"""Known-answer scenarios restated (as data) from the reference's own unit tests
(/root/reference/tests/poker/*.py, file:test cited per scenario).  The expected values were produced by
the reference's authors with the real HandRanks.dat, so they pin our regenerated table as well as the
showdown / side-pot / run-out logic.  Cards are encoded as the reference tests do:
rank + 13*suit + 1 with ranks 2..A = 0..12 and suits c,d,h,s = 0..3
(tests/poker/test_poker_gpu_showdown.py:7-9) -- NOT the table's own numbering, on purpose."""

ACTIVE, FOLDED, ALLIN, SITOUT = 0, 1, 2, 3


def enc(card: str) -> int:
    return "23456789TJQKA".index(card[0]) + 13 * "cdhs".index(card[1]) + 1


def cards(*cs):
    return [enc(c) for c in cs]


def ordered_deck(*cs):
    used = cards(*cs)
    return used + [c for c in range(1, 53) if c not in used]


_RUNOUT_DECK = ordered_deck("Ah", "Ad", "Kc", "Kd", "2s", "2c", "7d", "9h", "3s", "Js", "4c", "Qd")

SCENARIOS = [
    dict(name="showdown_pays_strongest_active_hand",            # test_poker_gpu_showdown.py:24
         n_players=2, poke=[("board", 0, cards("2c", "7d", "9h", "Js", "Kd")), ("hands", (0, 0), cards("Ah", "Qh")),
                            ("hands", (0, 1), cards("3c", "4d")), ("status", 0, [ACTIVE, ACTIVE]), ("stacks", 0, [50, 50]),
                            ("total_invested", 0, [20, 20]), ("pots", 0, 40), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [90, 50]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="showdown_excludes_folded_players",               # test_poker_gpu_showdown.py:43
         n_players=3, poke=[("board", 0, cards("As", "Ks", "Qs", "Js", "2d")), ("hands", (0, 0), cards("Ts", "3c")),
                            ("hands", (0, 1), cards("9h", "9d")), ("hands", (0, 2), cards("4c", "4d")),
                            ("status", 0, [ACTIVE, ACTIVE, FOLDED]), ("stacks", 0, [100, 100, 100]),
                            ("total_invested", 0, [30, 30, 30]), ("pots", 0, 90), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [190, 100, 100]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="showdown_splits_tied_even_pot",                  # test_poker_gpu_showdown.py:63
         n_players=2, poke=[("board", 0, cards("Ah", "Kd", "Qc", "Js", "9d")), ("hands", (0, 0), cards("2c", "3d")),
                            ("hands", (0, 1), cards("2d", "3c")), ("status", 0, [ACTIVE, ACTIVE]), ("stacks", 0, [10, 20]),
                            ("total_invested", 0, [12, 12]), ("pots", 0, 24), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [22, 32]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="side_pot_by_commitment",                         # test_poker_gpu_side_pot_showdown.py:28
         n_players=3, poke=[("board", 0, cards("As", "Ks", "Qs", "Js", "2d")), ("hands", (0, 0), cards("Ts", "3c")),
                            ("hands", (0, 1), cards("9h", "9d")), ("hands", (0, 2), cards("4c", "4d")),
                            ("status", 0, [ALLIN, ALLIN, ACTIVE]), ("stacks", 0, [0, 0, 100]),
                            ("total_invested", 0, [10, 50, 50]), ("pots", 0, 110), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [30, 80, 100]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="side_pot_splits_main_pot_first",                 # test_poker_gpu_side_pot_showdown.py:48
         n_players=3, poke=[("board", 0, cards("Ah", "Kd", "Qc", "Js", "3d")), ("hands", (0, 0), cards("2c", "4d")),
                            ("hands", (0, 1), cards("2d", "4c")), ("hands", (0, 2), cards("9h", "9c")),
                            ("status", 0, [ALLIN, ACTIVE, ACTIVE]), ("stacks", 0, [0, 70, 70]),
                            ("total_invested", 0, [10, 30, 30]), ("pots", 0, 70), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [15, 125, 70]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="side_pot_keeps_folded_chips",                    # test_poker_gpu_side_pot_showdown.py:68
         n_players=3, poke=[("board", 0, cards("As", "Ks", "Qs", "Js", "2d")), ("hands", (0, 0), cards("Ts", "3c")),
                            ("hands", (0, 1), cards("9h", "9d")), ("hands", (0, 2), cards("4c", "4d")),
                            ("status", 0, [ALLIN, ACTIVE, FOLDED]), ("stacks", 0, [0, 100, 100]),
                            ("total_invested", 0, [10, 50, 50]), ("pots", 0, 110), ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [30, 180, 100]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="side_pot_multiple_layers",                       # test_poker_gpu_side_pot_showdown.py:89
         n_players=4, poke=[("board", 0, cards("As", "Ks", "Qs", "Js", "2d")), ("hands", (0, 0), cards("Ts", "3c")),
                            ("hands", (0, 1), cards("9h", "9d")), ("hands", (0, 2), cards("Kc", "Kd")),
                            ("hands", (0, 3), cards("4c", "4d")), ("status", 0, [ALLIN, ALLIN, ALLIN, ACTIVE]),
                            ("stacks", 0, [0, 0, 0, 50]), ("total_invested", 0, [10, 30, 50, 50]), ("pots", 0, 140),
                            ("stages", 0, 4), ("is_done", 0, True)],
         call="resolve_terminated_games", expect=[("stacks", 0, [40, 60, 40, 50]), ("pots", 0, 0), ("stages", 0, 5)]),
    dict(name="side_pot_batched_mixed_shapes",                  # test_poker_gpu_side_pot_showdown.py:110
         n_players=3, n_games=2,
         poke=[("board", 0, cards("As", "Ks", "Qs", "Js", "2d")), ("hands", (0, 0), cards("Ts", "3c")),
               ("hands", (0, 1), cards("9h", "9d")), ("hands", (0, 2), cards("4c", "4d")), ("status", 0, [ALLIN, ALLIN, ACTIVE]),
               ("stacks", 0, [0, 0, 100]), ("total_invested", 0, [10, 50, 50]), ("pots", 0, 110), ("stages", 0, 4),
               ("is_done", 0, True),
               ("board", 1, cards("2c", "7d", "9h", "Js", "Kd")), ("hands", (1, 0), cards("Ah", "Qh")),
               ("hands", (1, 1), cards("3c", "4d")), ("hands", (1, 2), cards("5c", "6d")), ("status", 1, [ACTIVE, ACTIVE, FOLDED]),
               ("stacks", 1, [50, 50, 50]), ("total_invested", 1, [20, 20, 20]), ("pots", 1, 60), ("stages", 1, 4),
               ("is_done", 1, True)],
         call="resolve_terminated_games",
         expect=[("stacks", 0, [30, 80, 100]), ("stacks", 1, [110, 50, 50]), ("pots", None, [0, 0]), ("stages", None, [5, 5])]),
    dict(name="preflop_allin_runs_out_full_board",              # test_poker_gpu_preflop_allin_resolver.py:34,66
         n_players=2, poke=[("hands", (0, 0), cards("Ah", "Ad")), ("hands", (0, 1), cards("Kc", "Kd")), ("decks", 0, _RUNOUT_DECK),
                            ("deck_positions", 0, 4), ("board", 0, [-1] * 5), ("status", 0, [ALLIN, ALLIN]), ("stacks", 0, [0, 0]),
                            ("total_invested", 0, [50, 50]), ("pots", 0, 100), ("stages", 0, 0), ("is_done", 0, True)],
         call="resolve_terminated_games",
         expect=[("board", 0, cards("2c", "7d", "9h", "Js", "Qd")), ("deck_positions", 0, 12), ("pots", 0, 0), ("stages", 0, 5),
                 ("stacks", 0, [100, 0])]),
    dict(name="preflop_allin_splits_tied_runout",               # test_poker_gpu_preflop_allin_resolver.py:98
         n_players=2, poke=[("hands", (0, 0), cards("Ac", "Kd")), ("hands", (0, 1), cards("Ad", "Kc")),
                            ("decks", 0, ordered_deck("Ac", "Kd", "Ad", "Kc", "2s", "Qh", "Jh", "Td", "3s", "2c", "4c", "7d")),
                            ("deck_positions", 0, 4), ("board", 0, [-1] * 5), ("status", 0, [ALLIN, ALLIN]), ("stacks", 0, [10, 20]),
                            ("total_invested", 0, [12, 12]), ("pots", 0, 24), ("stages", 0, 0), ("is_done", 0, True)],
         call="resolve_terminated_games",
         expect=[("board", 0, cards("Qh", "Jh", "Td", "2c", "7d")), ("stacks", 0, [22, 32])]),
    dict(name="single_survivor_is_not_run_out",                 # test_poker_gpu_preflop_allin_resolver.py:182
         n_players=2, poke=[("decks", 0, _RUNOUT_DECK), ("deck_positions", 0, 4), ("board", 0, [-1] * 5),
                            ("status", 0, [ACTIVE, FOLDED]), ("stacks", 0, [0, 50]), ("pots", 0, 100), ("stages", 0, 0),
                            ("is_done", 0, True)],
         call="resolve_terminated_games",
         expect=[("board", 0, [-1] * 5), ("deck_positions", 0, 4), ("pots", 0, 100), ("stages", 0, 0)]),
    dict(name="river_auto_runout_resolves_showdown",            # test_poker_gpu_no_actor_rewards.py:118
         n_players=3, poke=[("status", 0, [ALLIN, ALLIN, FOLDED]), ("stacks", 0, [90, 90, 100]), ("idx", 0, 0), ("agg", 0, 0),
                            ("acted", 0, 0), ("highest", 0, 10), ("current_round_bet", 0, [10, 10, 0]),
                            ("total_invested", 0, [10, 10, 0]), ("pots", 0, 20), ("stages", 0, 3), ("is_done", 0, False),
                            ("board", 0, cards("2c", "7d", "9h", "Js", "Kd")), ("hands", (0, 0), cards("Ah", "Qh")),
                            ("hands", (0, 1), cards("3c", "4d")), ("hands", (0, 2), cards("5c", "5d"))],
         call=("step", [12]),
         expect=[("rewards", 0, 0.0), ("dones", 0, True), ("pots", 0, 0), ("stages", 0, 5), ("stacks", 0, [110, 90, 100])]),
    dict(name="preflop_auto_runout_advances_one_street",        # test_poker_gpu_no_actor_rewards.py:68
         n_players=3, poke=[("status", 0, [ALLIN, ALLIN, FOLDED]), ("stacks", 0, [90, 90, 100]), ("idx", 0, 0), ("agg", 0, 0),
                            ("acted", 0, 0), ("highest", 0, 10), ("current_round_bet", 0, [10, 10, 0]),
                            ("total_invested", 0, [10, 10, 0]), ("pots", 0, 20), ("stages", 0, 0), ("is_done", 0, False)],
         call=("step", [1]),
         expect=[("rewards", 0, 0.0), ("dones", 0, False), ("stages", 0, 1), ("pots", 0, 20)]),
    dict(name="batched_rewards_zero_only_without_legal_actor",  # test_poker_gpu_no_actor_rewards.py:135
         n_players=3, n_games=2,
         poke=[("status", 0, [ALLIN, ALLIN, FOLDED]), ("stacks", 0, [90, 90, 100]), ("idx", 0, 0), ("agg", 0, 0), ("acted", 0, 0),
               ("highest", 0, 10), ("current_round_bet", 0, [10, 10, 0]), ("total_invested", 0, [10, 10, 0]), ("pots", 0, 20),
               ("stages", 0, 0), ("is_done", 0, False),
               ("status", 1, [ACTIVE, ACTIVE, FOLDED]), ("stacks", 1, [100, 100, 100]), ("idx", 1, 0), ("agg", 1, 1), ("acted", 1, 0),
               ("highest", 1, 10), ("current_round_bet", 1, [0, 10, 0]), ("total_invested", 1, [0, 10, 0]), ("pots", 1, 20),
               ("stages", 1, 0), ("is_done", 1, False)],
         call=("step", [12, 1]),
         expect=[("rewards", 0, 0.0), ("rewards_nonzero", 1, True), ("dones", None, [False, False]), ("stages", None, [1, 0]),
                 ("current_round_bet", 1, [10, 10, 0]), ("idx", 1, 1)]),
    dict(name="step_does_not_mutate_terminal_games",            # test_poker_gpu_showdown.py:83
         n_players=3, poke=[("is_done", 0, True)], call=("step", [12]), expect=[("unchanged", None, ("status", "stacks", "idx", "pots", "stages"))]),
]
