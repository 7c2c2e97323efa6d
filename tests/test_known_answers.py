"""Known answers from the reference's own unit tests (tests/scenarios.py), checked on the oracle here
(CPU, -m "not gpu") and on the HIP path (tests/test_poker_gpu_parity.py imports the same runner)."""
import numpy as np
import pytest

from tests.scenarios import SCENARIOS


def run_scenario(sc, make_env, poke, read, step):
    env = make_env(sc["n_players"], sc.get("n_games", 1))
    before = {n: np.array(read(env, n)).copy() for n in ("status", "stacks", "idx", "pots", "stages")}
    for name, index, value in sc["poke"]:
        poke(env, name, index, value)
    for name, index, value in sc["poke"]:          # "unchanged" scenarios compare against the poked state
        if name in before:
            before[name] = np.array(read(env, name)).copy()
    extra = {}
    call = sc["call"]
    if isinstance(call, tuple):
        rewards, dones = step(env, call[1])
        extra["rewards"], extra["dones"] = np.asarray(rewards), np.asarray(dones)
        extra["rewards_nonzero"] = np.asarray(rewards) != 0
    else:
        getattr(env, call)()
    for name, index, want in sc["expect"]:
        if name == "unchanged":
            for n in want:
                assert np.array_equal(np.array(read(env, n)), before[n]), f"{sc['name']}: {n} changed"
            continue
        got = extra[name] if name in extra else np.array(read(env, name))
        got = got if index is None else got[index]
        if name == "rewards":
            assert abs(float(got) - want) < 1e-6, f"{sc['name']}: {name}[{index}] = {got}, want {want}"
        else:
            assert np.array_equal(np.asarray(got).astype(np.int64), np.asarray(want).astype(np.int64)), \
                f"{sc['name']}: {name}[{index}] = {np.asarray(got).tolist()}, want {want}"


def _oracle_env(oracle_table):
    from oracle import oracle as orc

    def make(n_players, n_games):
        env = orc.OraclePokerEnv(n_players=n_players, max_players=n_players, n_games=n_games, hand_ranks_table=oracle_table)
        rng = np.random.default_rng(0)
        decks = np.stack([rng.permutation(52) + 1 for _ in range(n_games)]).astype(np.int32)
        env.reset(options={"prefixed_decks": decks})
        return env

    def poke(env, name, index, value):
        getattr(env, name)[index] = value

    def read(env, name):
        return getattr(env, name)

    def step(env, actions):
        _, rew, dones, _, _ = env.step(np.asarray(actions, dtype=np.int64))
        return rew.copy(), dones.copy()

    return make, poke, read, step


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_oracle_reproduces_reference_known_answers(sc, oracle_table):
    run_scenario(sc, *_oracle_env(oracle_table))


# ---- the rest of the reference's white-box scenarios, as op lists (tests/scenarios_ops.py) ---------------------------
from tests.scenario_runner import OracleAdapter, run_ops  # noqa: E402
from tests.scenarios_ops import SCENARIOS as OP_SCENARIOS  # noqa: E402


def test_op_scenarios_cover_the_reference_acceptance_list():
    """scripts/Poker/test_poker_gpu_logic_runner.py:810-841 runs the logic matrix + the contract files; every family of
    that list is restated (names are ours, sources are cited per scenario)."""
    names = [s["name"] for s in OP_SCENARIOS]
    assert len(names) == len(set(names)) and len(names) >= 90
    families = {n.split("/")[0] for n in names}
    assert {"actions", "observation", "reset", "termination", "heads-up", "no-actor", "equity", "round", "street", "showdown",
            "step", "reward", "reset-rotation", "setup", "round-progression"} <= families
    assert all(s["src"].startswith("test_poker_gpu_") for s in OP_SCENARIOS)


@pytest.mark.parametrize("sc", OP_SCENARIOS, ids=[s["name"] for s in OP_SCENARIOS])
def test_oracle_reproduces_reference_op_scenarios(sc, oracle_table):
    run_ops(sc, OracleAdapter(oracle_table))
