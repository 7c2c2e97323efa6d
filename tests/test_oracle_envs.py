"""Pin oracle/envs_oracle.c (Blackjack, 2048, Particle2D restatements) against fixtures recorded from the
reference (tests/golden/make_golden.py), plus the tabular-Q helpers of utils/numba.py against their
closed forms."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

BJ_NAMES = ("deck_positions", "players_card_idx", "player_card_sums", "dealer_card_idx", "dealer_upcard", "dealer_card_sums",
            "players_cards", "dealer_cards", "terminated", "has_ace", "dealer_has_ace", "obs")


@pytest.mark.parametrize("ep", range(3))
def test_blackjack_oracle_matches_reference(golden_dir, ep):
    fx = np.load(golden_dir / "blackjack.npz")
    decks = fx[f"e{ep}/decks"].astype(np.int32)
    env = orc.OracleBlackjack(decks.shape[0])
    env.reset(decks)
    for n in BJ_NAMES:
        np.testing.assert_array_equal(getattr(env, n).astype(np.int32), fx[f"e{ep}/reset/{n}"], err_msg=f"reset {n}")
    acts = fx[f"e{ep}/actions"].astype(np.int64)
    for s in range(acts.shape[0]):
        obs, rew, term = env.step(acts[s])
        for n in BJ_NAMES:
            np.testing.assert_array_equal(getattr(env, n).astype(np.int32), fx[f"e{ep}/steps/{n}"][s].astype(np.int32), err_msg=f"step {s} {n}")
        np.testing.assert_array_equal(rew, fx[f"e{ep}/steps/rewards"][s].astype(np.int32), err_msg=f"step {s} rewards")
    assert env.terminated.mean() > 0.9


def test_tfe_oracle_matches_reference(golden_dir):
    fx = np.load(golden_dir / "tfe.npz")
    seed = int(fx["seed"])
    want = fx["boards"].astype(np.int32)
    steps, B = fx["actions"].shape
    n = want.shape[-1]
    boards = np.zeros((B, n, n), dtype=np.int32)
    score = np.zeros(B, dtype=np.int64)
    orc.tfe_reset(boards, score, n, seed)
    np.testing.assert_array_equal(boards, want[0])
    rewards = np.zeros(B, dtype=np.int32)
    dones = np.zeros(B, dtype=np.uint8)
    for s in range(steps):
        orc.tfe_step(boards, score, fx["actions"][s].astype(np.int64), rewards, dones, n, seed, s + 1)
        np.testing.assert_array_equal(boards, want[s + 1], err_msg=f"step {s}")
        np.testing.assert_array_equal(rewards, fx["rewards"][s].astype(np.int32), err_msg=f"step {s}")
        np.testing.assert_array_equal(dones, fx["dones"][s], err_msg=f"step {s}")
        np.testing.assert_array_equal(score, fx["scores"][s].astype(np.int64), err_msg=f"step {s}")
    assert want.max() >= 64 and fx["rewards"].max() >= 5      # the fixture exercises real merges


def test_particle2d_oracle_matches_reference(golden_dir):
    fx = np.load(golden_dir / "particle2d.npz")
    state = fx["state0"].astype(np.float32).copy()
    steps = np.zeros(state.shape[0], dtype=np.int32)
    for s in range(fx["actions"].shape[0]):
        obs, rew, term = orc.particle2d_step(state, fx["actions"][s], steps, 0.1, 20)
        # fp32, tolerance stated by SURVEY.md C.3: torch's norm kernel may order x*x+y*y differently
        np.testing.assert_allclose(obs, fx["obs"][s], rtol=1e-6, atol=1e-7, err_msg=f"step {s}")
        np.testing.assert_allclose(rew, fx["rewards"][s], rtol=1e-6, atol=1e-6, err_msg=f"step {s}")
        np.testing.assert_array_equal(term.astype(np.uint8), fx["terminated"][s], err_msg=f"step {s}")
    np.testing.assert_array_equal(steps, fx["steps_final"])


def test_q_helpers_follow_utils_numba():
    lib = orc.lib()
    q = np.array([0.1, 0.7, 0.7, -1.0], dtype=np.float64)
    # greedy branch: first maximal index (utils/numba.py:13-19); exploring branch: uniform index
    assert lib.oracle_select_action_epsilon_greedy(q.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(0.5), C.c_uint32(0)) == 1
    assert lib.oracle_select_action_epsilon_greedy(q.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(0.05), C.c_uint32(0xC0000000)) == 3
    cur = np.array([1.0, 2.0, 3.0, 4.0]); nxt = np.array([0.5, 9.0, -1.0, 2.0])
    lib.oracle_update_q_entry(cur.ctypes.data_as(C.c_void_p), 2, nxt.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(1.0),
                              C.c_double(0.99), 0)
    assert cur[2] == 3.0 + 0.1 * (1.0 + 0.99 * 9.0 - 3.0)          # utils/numba.py:33-39
    lib.oracle_update_q_entry(cur.ctypes.data_as(C.c_void_p), 0, nxt.ctypes.data_as(C.c_void_p), 4, C.c_double(0.5), C.c_double(-2.0),
                              C.c_double(0.99), 1)
    assert cur[0] == 1.0 + 0.5 * (-2.0 - 1.0)                       # terminal: target = reward


def test_oracle_shuffle_is_stable_argsort_of_philox_keys():
    """oracle_shuffle_decks (the definition the HIP reset kernels are held to) against numpy: deck = stable
    argsort of the 52 Philox key words + 1, also with keys cut to few bits (ties everywhere)."""
    from oracle import oracle as orc
    n, seed, id0, ep = 48, 31337, 1 << 33, 7
    keys = np.array([[orc.philox4x32(seed, id0 + t, ep * 16 + s) for s in range(13)] for t in range(n)],
                    dtype=np.uint32).reshape(n, 52)
    for bits in (0, 32, 6, 1):
        k = keys >> (32 - bits) if 0 < bits < 32 else keys
        want = np.argsort(k, axis=1, kind="stable").astype(np.int32) + 1
        np.testing.assert_array_equal(orc.shuffle_decks(seed, id0, ep, n, bits), want)
