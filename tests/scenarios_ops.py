"""The reference's white-box PokerGPU tests restated AS DATA (no reference code): each scenario is a list of ops --
poke a state tensor, call a method, step, expect -- run by tests/scenario_runner.py on the oracle (CPU) and on the HIP
path (GPU).  Source of every scenario: /root/reference/tests/poker/<file>:<line>; the acceptance list they come from
is scripts/Poker/test_poker_gpu_logic_runner.py:810-841.  tests/scenarios.py holds the showdown / side-pot /
run-out known answers in the older single-call format.

Ops:
  ("reset", options[, force_players])   env.reset(options); force_players = what the reference's patched torch.randint returns
  ("poke", name, index, value)          state[name][index] = value          (index None = whole tensor)
  ("active", n)                         env.active_players = n
  ("scalars", {...})                    reward constants w1 / w2 / K / alpha re-assigned
  ("call", method, *args)               execute_actions / calculate_equities / resolve_* / get_obs / post_blinds
  ("step", actions)                     env.step; keeps rewards / dones / seat_idx (info) / obs of this step
  ("snapshot", names)                   remember tensors for ("same", ...)
  ("expect", name, index, want)         exact (ints, bools); names also: dones rewards seat_idx obs active_players obs_shape
  ("approx", name, index, want, tol)    floats
  ("positive", name, index)             every entry > 0 (dealt cards)
  ("between", name, index, lo, hi)      every entry in [lo, hi]
  ("same", name, index)                 equals the snapshot
  ("obs_hand", game, seat)              obs[game, 5:7] == hands[game, seat]
  ("decks_are_permutations",)           every deck a permutation of 1..52, dealt hole cards distinct per game
Cards in pokes use the reference tests' encoding rank + 13*suit + 1 (tests/scenarios.py: enc)."""
import math

from tests.scenarios import cards

ACTIVE, FOLDED, ALLIN, SITOUT = 0, 1, 2, 3
FRESH = {"active_players": False, "q_agent_seat": 0, "rotation": 0}       # what every reference `_build_env` resets with


def S(name, src, n_players, ops, n_games=1, max_players=None, fresh=True, set_active=False):
    """fresh: the builder resets once with FRESH (as the reference's _build_env); set_active: `env.active_players = n_players` after it."""
    pre = ([("reset", FRESH)] if fresh else []) + ([("active", n_players)] if set_active else [])
    return dict(name=name, src=src, n_players=n_players, max_players=max_players or n_players, n_games=n_games, ops=pre + list(ops))


def pokes(game, **kw):
    return [("poke", k, game, v) for k, v in kw.items()]


SCENARIOS = []
add = SCENARIOS.append
LM = "test_poker_gpu_environment_logic_matrix.py"

# ---- actions/pot-fraction-* (logic_matrix:24-34,126-147): action ids 3..11 from a 100-chip pot
for action_id, bet in ((3, 25), (4, 33), (5, 50), (6, 75), (7, 100), (8, 150), (9, 200), (10, 300), (11, 400)):
    add(S(f"actions/pot-fraction-action-{action_id}-bets-{bet}", f"{LM}:24-34,126", 2,
          pokes(0, idx=0, agg=1, acted=0, highest=0) + [("poke", "current_round_bet", None, 0), ("poke", "total_invested", None, 0)] +
          pokes(0, stacks=[500, 500], status=[ACTIVE, ACTIVE], pots=100, last_raise_size=1) +
          [("call", "execute_actions", [action_id]),
           ("expect", "current_round_bet", 0, [bet, 0]), ("expect", "total_invested", 0, [bet, 0]), ("expect", "pots", 0, 100 + bet),
           ("expect", "stacks", 0, [500 - bet, 500]), ("expect", "highest", 0, bet), ("expect", "last_raise_size", 0, bet),
           ("expect", "agg", 0, 0), ("expect", "acted", 0, 1)]))

# ---- observation/* (logic_matrix:36-55,150-187)
for idx, button, rel in ((0, 1, 3), (1, 1, 0), (2, 1, 1), (3, 1, 2)):
    add(S(f"observation/relative-position-seat-{idx}-from-button-{button}", f"{LM}:36-41,150", 4,
          pokes(0, idx=idx, button=button) + [("call", "get_obs"), ("expect", "obs", (0, 8), rel)]))
for acting_bet, highest, call in ((0, 10, 10), (3, 10, 7), (9, 10, 1), (10, 10, 0)):
    add(S(f"observation/call-amount-bet-{acting_bet}-highest-{highest}", f"{LM}:43-48,160", 4,
          pokes(0, idx=0, highest=highest, current_round_bet=[acting_bet, 0, 0, 0]) + [("call", "get_obs"), ("expect", "obs", (0, 10), call)]))
for idx, flat in ((0, [102, 1, 12, 103, 2, 13, 104, 0, 14]), (1, [103, 2, 13, 104, 0, 14, 101, 0, 11]),
                  (2, [104, 0, 14, 101, 0, 11, 102, 1, 12]), (3, [101, 0, 11, 102, 1, 12, 103, 2, 13])):
    add(S(f"observation/opponents-wrap-from-seat-{idx}", f"{LM}:50-55,171", 4,
          pokes(0, idx=idx, stacks=[101, 102, 103, 104], status=[ACTIVE, FOLDED, ALLIN, ACTIVE], current_round_bet=[11, 12, 13, 14]) +
          [("call", "get_obs"), ("expect", "obs", (0, slice(13, 22)), flat)]))

# ---- reset/* (logic_matrix:57-76,184-246)
for cand in (2, 3, 4, 5, 6):
    add(S(f"reset/deck-position-after-{cand}-player-deal", f"{LM}:57-63,184", 6,
          [("reset", {"active_players": True, "q_agent_seat": 0, "rotation": 0}, cand), ("expect", "active_players", None, cand),
           ("expect", "deck_positions", None, [cand * 2]), ("positive", "hands", (0, slice(0, cand)))], fresh=False))
for cand, q_seat, want in ((2, 0, 2), (2, 2, 3), (2, 4, 5), (5, 1, 5)):
    add(S(f"reset/candidate-{cand}-q-seat-{q_seat}-gives-{want}-active", f"{LM}:65-70,204", 6,
          [("reset", {"active_players": True, "q_agent_seat": q_seat, "rotation": 0}, cand), ("expect", "active_players", None, want),
           ("expect", "status", (0, slice(want, 6)), [SITOUT] * (6 - want)), ("expect", "hands", (0, slice(want, 6)), [[-1, -1]] * (6 - want))],
          fresh=False))
for n, count, button, sb, bb, idx in ((2, 3, 0, 0, 1, 0), (4, 3, 2, 3, 0, 1), (3, 2, 1, 2, 0, 1)):
    add(S(f"reset/{n}-players-{count}-resets-button-{button}", f"{LM}:72-76,224", n,
          [("reset", FRESH)] * count + [("expect", "button", None, [button]), ("expect", "sb", None, [sb]), ("expect", "bb", None, [bb]),
                                        ("expect", "idx", None, [idx])], fresh=False))

# ---- termination/fold-resolution-* (logic_matrix:78-82,249-274)
for mode, done, status, stacks, pot in (("done_single", True, [FOLDED, ACTIVE, FOLDED], [10, 35, 30], 0),
                                        ("live_single", False, [FOLDED, ACTIVE, FOLDED], [10, 20, 30], 15),
                                        ("done_multi", True, [ACTIVE, ACTIVE, FOLDED], [10, 20, 30], 15)):
    add(S(f"termination/fold-resolution-{mode}", f"{LM}:78-82,249", 3,
          pokes(0, pots=15, stacks=[10, 20, 30], is_done=done, status=status) +
          [("call", "resolve_fold_winners"), ("expect", "stacks", 0, stacks), ("expect", "pots", 0, pot)]))

# ---- heads-up/check-around (logic_matrix:84-88,277-304)
HEADSUP_POSTFLOP = [("active", 2)] + pokes(0, status=[ACTIVE, ACTIVE], button=0, idx=1, agg=0, acted=0, highest=0, current_round_bet=[0, 0],
                                           total_invested=[10, 10], pots=20, is_done=False, equity_dirty=False, equities=[0.6, 0.4])
for stage, want_stage, want_done in ((1, 2, False), (2, 3, False), (3, 5, True)):
    add(S(f"heads-up/check-check-from-stage-{stage}", f"{LM}:84-88,277", 2,
          HEADSUP_POSTFLOP + pokes(0, stages=stage) + [("step", [1]), ("step", [1]), ("expect", "stages", 0, want_stage),
                                                       ("expect", "dones", 0, want_done)] +
          ([("expect", "pots", 0, 0)] if want_done else [("expect", "seat_idx", 0, 1)])))

# ---- no-actor/auto-runout (logic_matrix:90-94,307-329)
NO_ACTOR = pokes(0, status=[ALLIN, ALLIN, FOLDED], stacks=[90, 90, 100], idx=0, agg=0, acted=0, highest=10, current_round_bet=[10, 10, 0],
                 total_invested=[10, 10, 0], pots=20, is_done=False)
for stage, want_stage, want_done in ((1, 2, False), (2, 3, False), (3, 5, True)):
    add(S(f"no-actor/auto-runout-from-stage-{stage}", f"{LM}:90-94,307", 3,
          NO_ACTOR + pokes(0, stages=stage) + [("step", [12]), ("approx", "rewards", 0, 0.0, 1e-7), ("expect", "stages", 0, want_stage),
                                               ("expect", "dones", 0, want_done)]))

# ---- equity/preflop baseline (logic_matrix:96-101,332-340)
for n in (2, 3, 4, 6):
    add(S(f"equity/preflop-{n}-player-dirty-rows-become-half", f"{LM}:96-101,332", n,
          [("poke", "equities", None, 0.17)] + pokes(0, stages=0, equity_dirty=True) +
          [("call", "calculate_equities"), ("approx", "equities", 0, [0.5] * n, 1e-7), ("expect", "equity_dirty", 0, False)]))

add(S("setup/post-blinds-keeps-blind-seat-active-when-stack-remains", f"{LM}:349-361", 2,
      [("poke", "status", None, ACTIVE)] + pokes(0, stacks=[5, 5]) + [("poke", "current_round_bet", None, 0), ("poke", "total_invested", None, 0),
                                                                      ("poke", "pots", None, 0)] + pokes(0, bb=1) +
      [("call", "post_blinds"), ("expect", "status", (0, 1), ACTIVE), ("expect", "stacks", (0, 1), 4), ("expect", "pots", 0, 1)]))

# ---- actions/min-raise (logic_matrix:103-106,364-392)
MIN_RAISE = pokes(0, status=[ACTIVE, ACTIVE], idx=0, stacks=[50, 50], agg=1, acted=0)
add(S("actions/min-raise-opens-to-one-chip-from-unopened-pot", f"{LM}:103-106,372", 2,
      MIN_RAISE + pokes(0, highest=0) + [("poke", "current_round_bet", None, 0), ("poke", "total_invested", None, 0)] + pokes(0, pots=0, last_raise_size=1) +
      [("call", "execute_actions", [2]), ("expect", "current_round_bet", (0, 0), 1), ("expect", "highest", 0, 1), ("expect", "last_raise_size", 0, 1)]))
add(S("actions/min-raise-adds-last-raise-size-on-top-of-call-cost", f"{LM}:103-106,382", 2,
      MIN_RAISE + pokes(0, highest=10, current_round_bet=[6, 10], total_invested=[6, 10], pots=16, last_raise_size=4) +
      [("call", "execute_actions", [2]), ("expect", "current_round_bet", (0, 0), 14), ("expect", "highest", 0, 14), ("expect", "last_raise_size", 0, 4)]))

# ---- actions/exact-stack (logic_matrix:108-112,395-438)
EXACT = pokes(0, status=[ACTIVE, ACTIVE], idx=0, agg=1, acted=0, highest=10, current_round_bet=[4, 10], total_invested=[4, 10], pots=14)
add(S("actions/call-with-exact-stack-matches-bet-and-goes-allin", f"{LM}:108-112,402", 2,
      EXACT + pokes(0, stacks=[6, 50]) + [("call", "execute_actions", [1]), ("expect", "status", (0, 0), ALLIN),
                                         ("expect", "current_round_bet", (0, 0), 10), ("expect", "agg", 0, 1)]))
add(S("actions/allin-with-exact-stack-matching-call-does-not-reopen", f"{LM}:108-112,413", 2,
      EXACT + pokes(0, stacks=[6, 50], last_raise_size=4) + [("call", "execute_actions", [12]), ("expect", "status", (0, 0), ALLIN),
                                                             ("expect", "current_round_bet", (0, 0), 10), ("expect", "agg", 0, 1),
                                                             ("expect", "last_raise_size", 0, 4)]))
add(S("actions/allin-with-exact-stack-for-full-raise-reopens", f"{LM}:108-112,426", 2,
      EXACT + pokes(0, stacks=[10, 50], last_raise_size=4) + [("call", "execute_actions", [12]), ("expect", "status", (0, 0), ALLIN),
                                                              ("expect", "current_round_bet", (0, 0), 14), ("expect", "agg", 0, 0),
                                                              ("expect", "last_raise_size", 0, 4)]))

# ---- equity/mixed-dirty-mask (logic_matrix:114-119,441-466): four tables on four streets, all dirty
add(S("equity/mixed-dirty-mask-recalculates-every-street", f"{LM}:114-119,441", 2,
      [("poke", "equity_dirty", None, True), ("poke", "equities", None, 0.17), ("poke", "stages", None, [0, 1, 2, 3]),
       ("poke", "board", 1, [1, 2, 3, -1, -1]), ("poke", "board", 2, [1, 2, 3, 4, -1]), ("poke", "board", 3, [1, 2, 3, 4, 5])] +
      [("poke", "hands", (g, s), h) for g in (1, 2, 3) for s, h in ((0, [6, 7]), (1, [8, 9]))] +
      [("call", "calculate_equities"), ("approx", "equities", 0, [0.5, 0.5], 1e-7)] + [("between", "equities", g, 0.0, 1.0) for g in (1, 2, 3)] +
      [("expect", "equity_dirty", None, [False] * 4)], n_games=4))

add(S("round-progression/closes-on-current-actor-when-aggressor-checks-last", f"{LM}:469-485", 3,
      pokes(0, status=[ACTIVE, ACTIVE, ACTIVE], idx=1, agg=1, acted=2, highest=0, current_round_bet=[0, 0, 0], total_invested=[0, 0, 0], pots=0,
            stages=0, is_done=False) + [("step", [1]), ("expect", "stages", 0, 1), ("expect", "idx", 0, 1)]))

# ---- test_poker_gpu_round_progression.py
RP = "test_poker_gpu_round_progression.py"
add(S("round/skips-folded-and-allin-seats-when-selecting-next-actor", f"{RP}:46", 4,
      pokes(0, stacks=[100, 100, 0, 100], status=[ACTIVE, FOLDED, ALLIN, ACTIVE], idx=0, agg=0, acted=0, highest=0, current_round_bet=[0] * 4,
            total_invested=[0] * 4, is_done=False) +
      [("step", [1]), ("expect", "dones", 0, False), ("expect", "seat_idx", 0, 3), ("expect", "idx", 0, 3), ("expect", "stages", 0, 0)], set_active=True))
add(S("round/marks-round-over-and-transitions-to-next-street", f"{RP}:66", 4,
      pokes(0, status=[ACTIVE, FOLDED, ACTIVE, ACTIVE], idx=0, agg=2, acted=2, highest=0, current_round_bet=[0] * 4, total_invested=[0] * 4,
            is_done=False, stages=0) +
      [("step", [1]), ("expect", "dones", 0, False), ("expect", "stages", 0, 1), ("expect", "highest", 0, 0),
       ("expect", "current_round_bet", 0, [0] * 4), ("expect", "agg", 0, 1), ("positive", "board", (0, slice(0, 3)))], set_active=True))
add(S("round/fold-leaves-single-survivor-and-ends-hand", f"{RP}:88", 2,
      pokes(0, stacks=[50, 60], status=[ACTIVE, ACTIVE], idx=0, agg=1, acted=0, highest=0, current_round_bet=[0, 0], total_invested=[0, 0], pots=10,
            is_done=False) +
      [("step", [0]), ("expect", "dones", 0, True), ("expect", "status", 0, [FOLDED, ACTIVE]), ("expect", "pots", 0, 0),
       ("expect", "stacks", 0, [50, 70])], set_active=True))
add(S("round/multiway-preflop-closes-after-big-blind-checks-option", f"{RP}:109", 4,
      [op for _ in range(4) for op in (("step", [1]), ("expect", "dones", 0, False))] +
      [("expect", "stages", 0, 1), ("expect", "idx", 0, 1), ("expect", "seat_idx", 0, 1), ("expect", "current_round_bet", 0, [0] * 4),
       ("positive", "board", (0, slice(0, 3)))], set_active=True))
add(S("round/multiway-postflop-checkaround-advances-to-turn", f"{RP}:125", 4,
      [op for _ in range(4) for op in (("step", [1]), ("expect", "dones", 0, False))] + [("expect", "stages", 0, 1), ("expect", "idx", 0, 1)] +
      [op for _ in range(4) for op in (("step", [1]), ("expect", "dones", 0, False))] +
      [("expect", "stages", 0, 2), ("expect", "idx", 0, 1), ("expect", "seat_idx", 0, 1), ("expect", "current_round_bet", 0, [0] * 4),
       ("positive", "board", (0, 3))], set_active=True))
add(S("round/heads-up-postflop-opener-is-first-active-left-of-button", f"{RP}:146", 2,
      [op for _ in range(2) for op in (("step", [1]), ("expect", "dones", 0, False))] +
      [("expect", "stages", 0, 1), ("expect", "button", 0, 0), ("expect", "idx", 0, 1), ("expect", "seat_idx", 0, 1)], set_active=True))


def _raise_state(acting_stack):        # round_progression.py:19-43 _configure_raise_state(acting 0, aggressor 2, highest 20, bet 15, lrs 10, acted 2)
    return pokes(0, idx=0, agg=2, acted=2, highest=20, last_raise_size=10, current_round_bet=[15, 0, 20], total_invested=[15, 0, 20],
                 stacks=[acting_stack, 100, 100], status=[ACTIVE] * 3, pots=35, is_done=False)


add(S("actions/short-all-in-does-not-reopen-or-shrink-min-raise", f"{RP}:159", 3,
      _raise_state(8) + [("call", "execute_actions", [12]), ("expect", "current_round_bet", (0, 0), 23), ("expect", "highest", 0, 23),
                         ("expect", "agg", 0, 2), ("expect", "acted", 0, 3), ("expect", "last_raise_size", 0, 10), ("expect", "status", (0, 0), ALLIN)],
      set_active=True))
add(S("actions/full-all-in-reopens-and-updates-min-raise", f"{RP}:183", 3,
      _raise_state(20) + [("call", "execute_actions", [12]), ("expect", "current_round_bet", (0, 0), 35), ("expect", "highest", 0, 35),
                          ("expect", "agg", 0, 0), ("expect", "acted", 0, 1), ("expect", "last_raise_size", 0, 15), ("expect", "status", (0, 0), ALLIN)],
      set_active=True))
# the reference patches calculate_equities to return [0.8, 0.2, 0.4]; the same state as data: those equities, dirty flag cleared
add(S("reward/uses-acting-seat-equity-before-the-turn-advances", f"{RP}:207", 3,
      [("scalars", dict(w1=1.0, w2=0.0, K=100, alpha=1))] +
      pokes(0, status=[ACTIVE] * 3, idx=0, agg=2, acted=0, highest=1, current_round_bet=[0, 0, 1], total_invested=[0, 0, 1], pots=1, is_done=False,
            stages=0, equity_dirty=False, equities=[0.8, 0.2, 0.4]) +
      [("step", [1]), ("expect", "seat_idx", 0, 1), ("approx", "rewards", 0, math.tanh((0.8 * 2.0) / 100.0), 1e-6)], set_active=True))
add(S("equity/recomputed-after-street-transition", f"{RP}:279", 4,
      pokes(0, status=[ACTIVE, FOLDED, ACTIVE, ACTIVE], idx=0, agg=2, acted=2, highest=0, current_round_bet=[0] * 4, total_invested=[0] * 4,
            is_done=False, stages=0, equity_dirty=True) +
      [("step", [1]), ("expect", "stages", 0, 1), ("expect", "equity_dirty", 0, True), ("step", [1]), ("expect", "equity_dirty", 0, False)],
      set_active=True))

# ---- test_poker_gpu_headsup_opening_contracts.py
HU = "test_poker_gpu_headsup_opening_contracts.py"
add(S("heads-up/reset-button-is-small-blind-other-seat-big-blind", f"{HU}:19", 2,
      [("expect", "button", None, [0]), ("expect", "sb", None, [0]), ("expect", "bb", None, [1]), ("expect", "idx", None, [0]),
       ("expect", "current_round_bet", 0, [0, 1]), ("expect", "total_invested", 0, [0, 1]), ("expect", "pots", None, [1])]))
add(S("heads-up/second-hand-rotates-button-and-keeps-opening-order", f"{HU}:31", 2,
      [("reset", FRESH), ("expect", "button", None, [1]), ("expect", "sb", None, [1]), ("expect", "bb", None, [0]), ("expect", "idx", None, [1]),
       ("expect", "current_round_bet", 0, [1, 0]), ("expect", "total_invested", 0, [1, 0]), ("expect", "pots", None, [1])]))
add(S("heads-up/preflop-call-then-check-advances-to-flop-big-blind-first", f"{HU}:45", 2,
      [("step", [1]), ("expect", "dones", 0, False), ("expect", "stages", 0, 0), ("expect", "idx", 0, 1), ("expect", "seat_idx", 0, 1),
       ("expect", "current_round_bet", 0, [1, 1]), ("step", [1]), ("expect", "dones", 0, False), ("expect", "stages", 0, 1), ("expect", "idx", 0, 1),
       ("expect", "seat_idx", 0, 1), ("expect", "current_round_bet", 0, [0, 0]), ("positive", "board", (0, slice(0, 3)))]))
add(S("heads-up/button-fold-gives-big-blind-the-opening-pot", f"{HU}:65", 2,
      pokes(0, stacks=[100, 99], current_round_bet=[0, 1], total_invested=[0, 1], pots=1, idx=0, agg=1, highest=1, status=[ACTIVE, ACTIVE], acted=0,
            is_done=False) +
      [("step", [0]), ("expect", "dones", 0, True), ("expect", "status", 0, [FOLDED, ACTIVE]), ("expect", "stacks", 0, [100, 100]), ("expect", "pots", 0, 0)]))

# ---- test_poker_gpu_reset_rotation.py: persistent stacks, button forced back to 0 before the second reset
RR = "test_poker_gpu_reset_rotation.py"


def _rotated(stacks, rotation, starting=100, max_bbs=1000):       # reset_rotation.py:28-42 with the big blind (seat 0 after button 0 -> +1... see file) posting 1
    fixed = [starting if (s == 0 or s > max_bbs) else s for s in stacks]
    n = len(fixed)
    rolled = [fixed[(i - rotation) % n] for i in range(n)]
    rolled[0] -= 1
    return rolled


for label, stacks, rotation, want, kw in (("rolls-persistent-stacks-from-options-rotation", [10, 20, 30], 1, [29, 10, 20], None),
                                           ("options-rotation-equals-explicit-rotation-argument", [10, 20, 30], 1, [29, 10, 20], 1),
                                           ("zero-rotation-keeps-stack-order-before-blind-post", [10, 20, 30], 0, [9, 20, 30], None),
                                           ("wraps-rotation-larger-than-table-size", [10, 20, 30], 4, _rotated([10, 20, 30], 4), None),
                                           ("restores-invalid-persistent-stacks-before-rotating", [0, 1001, 25], 1, _rotated([0, 1001, 25], 1), None)):
    opts = {"active_players": False, "q_agent_seat": 0, "rotation": rotation}
    add(S(f"reset-rotation/{label}", f"{RR}:45-91", 3,
          pokes(0, stacks=stacks) + [("poke", "button", None, [0]), ("reset", dict(opts, **({"_rotation_kwarg": kw} if kw is not None else {}))),
                                     ("expect", "stacks", 0, want)], set_active=True))
assert _rotated([10, 20, 30], 1) == [29, 10, 20] and _rotated([10, 20, 30], 0) == [9, 20, 30]

# ---- test_poker_gpu_street_actor_reset.py
SA = "test_poker_gpu_street_actor_reset.py"


def _round_end(game, status, idx, agg, acted, stage, n):          # street_actor_reset.py:31-51
    return pokes(game, status=status, idx=idx, agg=agg, acted=acted, highest=0, stages=stage, current_round_bet=[0] * n, total_invested=[0] * n,
                 is_done=False, pots=0)


add(S("street/flop-transition-restarts-from-first-seat-left-of-button", f"{SA}:54", 4,
      _round_end(0, [ACTIVE] * 4, 0, 1, 3, 0, 4) + [("step", [1]), ("expect", "dones", 0, False), ("expect", "stages", 0, 1), ("expect", "idx", 0, 1),
                                                    ("expect", "seat_idx", 0, 1)], set_active=True))
add(S("street/turn-transition-skips-folded-seats-left-of-button", f"{SA}:74", 4,
      _round_end(0, [ACTIVE, FOLDED, ACTIVE, ACTIVE], 0, 2, 2, 1, 4) + [("step", [1]), ("expect", "stages", 0, 2), ("expect", "idx", 0, 2),
                                                                        ("expect", "agg", 0, 1)], set_active=True))
add(S("street/river-transition-skips-all-in-seats-left-of-button", f"{SA}:93", 4,
      _round_end(0, [ACTIVE, ALLIN, ACTIVE, ACTIVE], 0, 2, 2, 2, 4) + [("step", [1]), ("expect", "stages", 0, 3), ("expect", "idx", 0, 2),
                                                                       ("positive", "board", (0, 4))], set_active=True))
add(S("street/heads-up-transition-uses-the-other-active-seat", f"{SA}:112", 2,
      _round_end(0, [ACTIVE, ACTIVE], 0, 1, 1, 1, 2) + [("step", [1]), ("expect", "dones", 0, False), ("expect", "stages", 0, 2), ("expect", "idx", 0, 1),
                                                        ("expect", "seat_idx", 0, 1)], set_active=True))
add(S("street/batched-transition-resets-only-finished-rounds", f"{SA}:132", 4,
      _round_end(0, [ACTIVE, FOLDED, ACTIVE, ACTIVE], 0, 2, 2, 0, 4) + _round_end(1, [ACTIVE, FOLDED, ALLIN, ACTIVE], 0, 0, 0, 0, 4) +
      [("step", [1, 1]), ("expect", "dones", None, [False, False]), ("expect", "stages", None, [1, 0]), ("expect", "idx", None, [2, 3]),
       ("expect", "seat_idx", None, [2, 3]), ("obs_hand", 0, 2), ("obs_hand", 1, 3)], n_games=2, set_active=True))

# ---- test_poker_gpu_reset_batch_contracts.py
RB = "test_poker_gpu_reset_batch_contracts.py"
add(S("reset/none-options-initialise-a-full-ring", f"{RB}:26", 4,
      [("reset", None), ("expect", "obs_shape", None, [1, 22]), ("expect", "active_players", None, 4), ("expect", "deck_positions", None, [8]),
       ("expect", "button", None, [0]), ("expect", "sb", None, [1]), ("expect", "bb", None, [2]), ("expect", "idx", None, [3]),
       ("expect", "status", 0, [ACTIVE] * 4)], fresh=False))
add(S("reset/partial-options-without-active-players-key", f"{RB}:42", 4,
      [("reset", {"rotation": 0}), ("expect", "obs_shape", None, [1, 22]), ("expect", "active_players", None, 4), ("expect", "status", 0, [ACTIVE] * 4)],
      fresh=False))
add(S("reset/random-active-players-respects-q-seat-floor", f"{RB}:53", 6,
      [("reset", {"active_players": True, "q_agent_seat": 4, "rotation": 0}, 2), ("expect", "active_players", None, 5),
       ("expect", "status", 0, [ACTIVE] * 5 + [SITOUT]), ("expect", "hands", (0, 5), [-1, -1])], fresh=False))
add(S("reset/inactive-seats-sit-out-with-negative-hands", f"{RB}:67", 6,
      [("reset", {"active_players": True, "q_agent_seat": 0, "rotation": 0}, 4), ("expect", "active_players", None, 4),
       ("expect", "status", 0, [ACTIVE] * 4 + [SITOUT] * 2), ("expect", "hands", (0, slice(4, 6)), [[-1, -1], [-1, -1]])], fresh=False))
add(S("reset/decks-are-permutations-and-hole-cards-unique", f"{RB}:81", 6, [("decks_are_permutations",)], n_games=3))
add(S("reset/second-reset-advances-button-blinds-and-first-actor", f"{RB}:93", 4,
      [("reset", FRESH), ("expect", "button", None, [1]), ("expect", "sb", None, [2]), ("expect", "bb", None, [3]), ("expect", "idx", None, [0])]))
add(S("showdown/odd-chip-goes-to-the-only-eligible-contributor-of-a-tied-main-pot", f"{RB}:105", 2,
      pokes(0, board=cards("Ah", "Kd", "Qc", "Js", "9d")) + [("poke", "hands", (0, 0), cards("2c", "3d")), ("poke", "hands", (0, 1), cards("2d", "3c"))] +
      pokes(0, status=[ACTIVE, ACTIVE], stacks=[10, 20], total_invested=[12, 13], pots=25, stages=4, is_done=True) +
      [("call", "resolve_terminated_games"), ("expect", "stacks", 0, [22, 33]), ("expect", "pots", 0, 0), ("expect", "stages", 0, 5)]))
add(S("showdown/done-single-survivor-row-is-left-for-fold-resolution", f"{RB}:124", 2,
      pokes(0, status=[ACTIVE, FOLDED], stacks=[50, 60], total_invested=[10, 10], pots=20, stages=2, is_done=True) +
      [("call", "resolve_terminated_games"), ("expect", "stacks", 0, [50, 60]), ("expect", "pots", 0, 20), ("expect", "stages", 0, 2)]))
add(S("step/mixed-batch-keeps-each-row-isolated", f"{RB}:141", 4,
      pokes(0, status=[ACTIVE, ACTIVE, FOLDED, FOLDED], stacks=[45, 40, 100, 100], idx=0, agg=1, acted=0, highest=10, current_round_bet=[5, 10, 0, 0],
            total_invested=[5, 10, 0, 0], pots=15, is_done=False) +
      pokes(1, status=[ACTIVE, FOLDED, ACTIVE, ACTIVE], idx=0, agg=2, acted=2, highest=0, current_round_bet=[0] * 4, total_invested=[0] * 4, pots=0,
            stages=0, is_done=False) +
      pokes(2, status=[ALLIN, ALLIN, FOLDED, FOLDED], stacks=[90, 90, 100, 100], idx=0, agg=0, acted=0, highest=10, current_round_bet=[10, 10, 0, 0],
            total_invested=[10, 10, 0, 0], pots=20, stages=0, is_done=False) +
      pokes(3, status=[ACTIVE] * 4, stacks=[70, 80, 90, 100], idx=2, agg=1, acted=1, highest=5, current_round_bet=[5] * 4, total_invested=[5] * 4,
            pots=20, stages=2, is_done=True) +
      [("snapshot", ("status", "stacks", "idx", "pots", "stages")), ("step", [0, 1, 12, 12]),
       ("expect", "dones", 0, True), ("expect", "stacks", 0, [45, 55, 100, 100]), ("expect", "pots", 0, 0),
       ("expect", "dones", 1, False), ("expect", "stages", 1, 1), ("expect", "idx", 1, 2), ("positive", "board", (1, slice(0, 3))),
       ("expect", "dones", 2, False), ("expect", "stages", 2, 1), ("approx", "rewards", 2, 0.0, 1e-7), ("positive", "board", (2, slice(0, 3))),
       ("expect", "dones", 3, True)] + [("same", n, 3) for n in ("status", "stacks", "idx", "pots", "stages")] +
      [("expect", "seat_idx", 1, 2), ("expect", "seat_idx", 3, 2)], n_games=4))

# ---- test_poker_gpu_action_terminal_contracts.py
AT = "test_poker_gpu_action_terminal_contracts.py"
add(S("actions/call-uses-remaining-stack-and-marks-allin", f"{AT}:26", 2,
      pokes(0, idx=0, highest=10, current_round_bet=[4, 10], total_invested=[4, 10], stacks=[3, 50], pots=14, status=[ACTIVE, ACTIVE], acted=0) +
      [("call", "execute_actions", [1]), ("expect", "stacks", 0, [0, 50]), ("expect", "current_round_bet", 0, [7, 10]),
       ("expect", "total_invested", 0, [7, 10]), ("expect", "pots", 0, 17), ("expect", "status", (0, 0), ALLIN), ("expect", "acted", 0, 1)],
      set_active=True))
add(S("actions/min-raise-reopens-action-and-updates-raise-size", f"{AT}:47", 3,
      pokes(0, idx=0, agg=2, acted=2, highest=10, last_raise_size=4, current_round_bet=[6, 0, 10], total_invested=[6, 0, 10], stacks=[50, 50, 50],
            status=[ACTIVE] * 3, pots=16) +
      [("call", "execute_actions", [2]), ("expect", "current_round_bet", 0, [14, 0, 10]), ("expect", "total_invested", 0, [14, 0, 10]),
       ("expect", "pots", 0, 24), ("expect", "stacks", 0, [42, 50, 50]), ("expect", "highest", 0, 14), ("expect", "agg", 0, 0),
       ("expect", "last_raise_size", 0, 4), ("expect", "acted", 0, 1)], set_active=True))
add(S("termination/fold-winner-payout-is-idempotent", f"{AT}:72", 3,
      pokes(0, status=[FOLDED, ACTIVE, FOLDED], stacks=[10, 20, 30], pots=15, is_done=True) +
      [("call", "resolve_fold_winners"), ("expect", "stacks", 0, [10, 35, 30]), ("call", "resolve_fold_winners"), ("expect", "stacks", 0, [10, 35, 30]),
       ("expect", "pots", 0, 0)], set_active=True))
add(S("showdown/noop-when-no-done-rows-need-resolution", f"{AT}:88", 2,
      [("snapshot", ("board", "stacks", "pots", "stages", "deck_positions")), ("call", "resolve_terminated_games")] +
      [("same", n, None) for n in ("board", "stacks", "pots", "stages", "deck_positions")], set_active=True))
ORDERED = list(range(1, 53))
add(S("showdown/turn-runout-keeps-board-and-burns-once", f"{AT}:105", 2,
      pokes(0, decks=ORDERED, deck_positions=10, board=[11, 22, 33, 44, -1]) + [("poke", "hands", (0, 0), [1, 2]), ("poke", "hands", (0, 1), [3, 4])] +
      pokes(0, status=[ACTIVE, ACTIVE], stacks=[100, 100], total_invested=[10, 10], pots=20, stages=2, is_done=True) +
      [("call", "resolve_terminated_games"), ("expect", "board", 0, [11, 22, 33, 44, 12]), ("expect", "deck_positions", 0, 12), ("expect", "pots", 0, 0),
       ("expect", "stages", 0, 5), ("expect", "stacks_sum", 0, 220)], set_active=True))
add(S("showdown/flop-runout-keeps-flop-deals-turn-then-river", f"{AT}:129", 2,
      pokes(0, decks=ORDERED, deck_positions=10, board=[11, 22, 33, -1, -1]) + [("poke", "hands", (0, 0), [1, 2]), ("poke", "hands", (0, 1), [3, 4])] +
      pokes(0, status=[ACTIVE, ACTIVE], stacks=[100, 100], total_invested=[10, 10], pots=20, stages=1, is_done=True) +
      [("call", "resolve_terminated_games"), ("expect", "board", 0, [11, 22, 33, 12, 14]), ("expect", "deck_positions", 0, 14), ("expect", "pots", 0, 0),
       ("expect", "stages", 0, 5), ("expect", "stacks_sum", 0, 220)], set_active=True))
add(S("equity/clean-rows-are-left-untouched", f"{AT}:152", 2,
      [("poke", "equities", 0, [0.2, 0.8]), ("poke", "equities", 1, [0.7, 0.3]), ("poke", "equity_dirty", None, [False, True]),
       ("poke", "stages", None, [0, 0]), ("call", "calculate_equities"), ("approx", "equities", 0, [0.2, 0.8], 1e-7),
       ("approx", "equities", 1, [0.5, 0.5], 1e-7), ("expect", "equity_dirty", None, [False, False])], n_games=2, set_active=True))
add(S("step/clears-round-state-after-fold-ends-hand", f"{AT}:166", 2,
      pokes(0, status=[ACTIVE, ACTIVE], idx=0, agg=1, acted=0, highest=10, current_round_bet=[5, 10], total_invested=[5, 10], pots=15, stacks=[45, 40],
            is_done=False) +
      [("step", [0]), ("expect", "dones", 0, True), ("expect", "pots", 0, 0), ("expect", "current_round_bet", 0, [0, 0]),
       ("expect", "total_invested", 0, [0, 0]), ("expect", "highest", 0, 0), ("expect", "stacks", 0, [45, 55])], set_active=True))
add(S("step/clears-round-state-after-river-showdown", f"{AT}:188", 2,
      pokes(0, board=cards("2c", "7d", "9h", "Js", "Kd")) + [("poke", "hands", (0, 0), cards("Ah", "Qh")), ("poke", "hands", (0, 1), cards("3c", "4d"))] +
      pokes(0, status=[ACTIVE, ACTIVE], idx=0, agg=1, acted=1, highest=0, current_round_bet=[0, 0], total_invested=[10, 10], pots=20, stacks=[50, 50],
            stages=3, is_done=False, equity_dirty=False, equities=[0.8, 0.2]) +
      [("step", [1]), ("expect", "dones", 0, True), ("expect", "pots", 0, 0), ("expect", "stages", 0, 5), ("expect", "current_round_bet", 0, [0, 0]),
       ("expect", "total_invested", 0, [0, 0]), ("expect", "highest", 0, 0), ("expect", "stacks", 0, [70, 50])], set_active=True))
