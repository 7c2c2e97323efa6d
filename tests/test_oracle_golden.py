"""Pin the oracle (oracle/poker_oracle.c) against fixtures produced by running the reference
(/root/reference/environments/Poker/PokerGPU.py) -- see tests/golden/make_golden.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import DYN_BOOL, DYN_I32, DYN_ROWS, INT_KEYS, assert_rewards_close, assert_state_equal, reward_tol


def _env_snapshot(env):
    d = {n: getattr(env, n) for n in INT_KEYS}
    d["equities"] = env.equities
    d["obs"] = env.obs
    return d


@pytest.fixture(scope="module")
def rollouts(golden_dir):
    return np.load(golden_dir / "poker_rollouts.npz")


@pytest.fixture(scope="module")
def methods(golden_dir):
    return np.load(golden_dir / "poker_methods.npz")


def _cases(golden_dir):
    return [str(c) for c in np.load(golden_dir / "poker_rollouts.npz")["cases"]]


def test_rollout_cases_present(rollouts):
    assert len(rollouts["cases"]) == 6


@pytest.mark.parametrize("case", ["p10_uniform", "p10_callish", "p10_allin", "p6_allin", "p2_headsup", "p4_wild"])
def test_oracle_matches_reference_rollout(rollouts, oracle_table, case):
    P, MP, N, episodes, steps = [int(x) for x in rollouts[f"{case}/meta"]]
    env = orc.OraclePokerEnv(n_players=P, max_players=MP, n_games=N, starting_bbs=100, max_bbs=1000,
                             w1=.5, w2=.3, K=100, alpha=50, hand_ranks_table=oracle_table)
    for e in range(episodes):
        A = int(rollouts[f"{case}/A"][e])
        decks = rollouts[f"{case}/e{e}/decks"].astype(np.int32)
        env.reset(options={"active_players": A, "q_agent_seat": int(rollouts[f"{case}/q_seat"][e]),
                           "rotation": int(rollouts[f"{case}/rotation"][e]), "prefixed_decks": decks})
        want = {k: rollouts[f"{case}/e{e}/reset/{k}"] for k in INT_KEYS + ("equities", "obs")}
        assert_state_equal(_env_snapshot(env), want, ctx=f"{case} e{e} reset")
        np.testing.assert_array_equal(env.obs, want["obs"])
        np.testing.assert_array_equal(env.equities, want["equities"])
        acts = rollouts[f"{case}/e{e}/actions"].astype(np.int64)
        for s in range(steps):
            obs, rew, dones, _, _ = env.step(acts[s])
            want = {k: rollouts[f"{case}/e{e}/steps/{k}"][s] for k in INT_KEYS + ("equities", "obs")}
            ctx = f"{case} e{e} step {s}"
            assert_state_equal(_env_snapshot(env), want, ctx=ctx)
            np.testing.assert_array_equal(obs, want["obs"], err_msg=ctx)
            np.testing.assert_array_equal(env.equities, want["equities"], err_msg=ctx)
            np.testing.assert_array_equal(dones.astype(np.uint8), rollouts[f"{case}/e{e}/steps/dones"][s], err_msg=ctx)
            assert_rewards_close(rew, rollouts[f"{case}/e{e}/steps/rewards"][s], 50, ctx)


def _poked_env(methods, key, oracle_table):
    P, A, N = [int(x) for x in methods[f"{key}/meta"]]
    env = orc.OraclePokerEnv(n_players=P, max_players=10, n_games=N, w1=.5, w2=.3, K=100, alpha=50,
                             hand_ranks_table=oracle_table)
    env.reset(options={"active_players": A, "prefixed_decks": methods[f"{key}/decks"].astype(np.int32)})
    for k in DYN_I32 + DYN_ROWS:
        getattr(env, k)[...] = methods[f"{key}/pre/{k}"]
    for k in DYN_BOOL:
        getattr(env, k)[...] = methods[f"{key}/pre/{k}"]
    env.equities[...] = methods[f"{key}/pre/equities"]
    return env


@pytest.mark.parametrize("ci", range(5))
def test_oracle_methods_match_reference(methods, oracle_table, ci):
    key = f"c{ci}"
    env = _poked_env(methods, key, oracle_table)
    np.testing.assert_array_equal(env.get_obs(), methods[f"{key}/get_obs/obs"])

    env = _poked_env(methods, key, oracle_table)
    env.calculate_equities()
    np.testing.assert_array_equal(env.equities, methods[f"{key}/calculate_equities/equities"])
    np.testing.assert_array_equal(env.equity_dirty, methods[f"{key}/calculate_equities/equity_dirty"])

    env = _poked_env(methods, key, oracle_table)
    env.execute_actions(methods[f"{key}/execute_actions/actions"])
    want = {k: methods[f"{key}/execute_actions/post/{k}"] for k in INT_KEYS}
    assert_state_equal(_env_snapshot(env), want, ctx=f"{key} execute_actions")

    env = _poked_env(methods, key, oracle_table)
    env.is_done[...] = methods[f"{key}/resolve/is_done"]
    env.resolve_fold_winners()
    env.resolve_terminated_games()
    want = {k: methods[f"{key}/resolve/post/{k}"] for k in INT_KEYS}
    assert_state_equal(_env_snapshot(env), want, ctx=f"{key} resolve")

    env = _poked_env(methods, key, oracle_table)
    env.is_done[...] = methods[f"{key}/step/is_done_pre"]
    obs, rew, dones, _, _ = env.step(methods[f"{key}/step/actions"])
    want = {k: methods[f"{key}/step/post/{k}"] for k in INT_KEYS}
    assert_state_equal(_env_snapshot(env), want, ctx=f"{key} step")
    np.testing.assert_array_equal(obs, methods[f"{key}/step/post/obs"])
    np.testing.assert_array_equal(env.equities, methods[f"{key}/step/post/equities"])
    np.testing.assert_array_equal(dones.astype(np.uint8), methods[f"{key}/step/dones"])
    assert_rewards_close(rew, methods[f"{key}/step/rewards"], 50, f"{key} step")
