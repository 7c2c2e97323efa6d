"""GPU parity for the Blackjack, 2048 and Particle2D step kernels (csrc/envs.hip) through the C ABI:
against fixtures recorded from the reference, and against the oracle at BASELINE.json's sizes."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

BJ_NAMES = ("deck_positions", "players_card_idx", "player_card_sums", "dealer_card_idx", "dealer_upcard", "dealer_card_sums",
            "players_cards", "dealer_cards", "terminated", "has_ace", "dealer_has_ace", "obs")


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("ep", range(3))
def test_blackjack_hip_matches_reference(golden_dir, ep):
    from pulselib_amd.environments.blackjack import BlackJack
    fx = np.load(golden_dir / "blackjack.npz")
    decks = fx[f"e{ep}/decks"].astype(np.int32)
    env = BlackJack(torch.device(DEV), decks.shape[0])
    obs, info = env.reset(options={"decks": torch.from_numpy(decks)})
    for n in BJ_NAMES:
        np.testing.assert_array_equal(_np(getattr(env, n)).astype(np.int32), fx[f"e{ep}/reset/{n}"], err_msg=f"reset {n}")
    acts = fx[f"e{ep}/actions"].astype(np.int64)
    for s in range(acts.shape[0]):
        obs, rew, term, trunc, _ = env.step(torch.from_numpy(acts[s]))
        assert trunc is None                                            # blackjack.py:186
        for n in BJ_NAMES:
            np.testing.assert_array_equal(_np(getattr(env, n)).astype(np.int32), fx[f"e{ep}/steps/{n}"][s].astype(np.int32),
                                          err_msg=f"step {s} {n}")
        np.testing.assert_array_equal(_np(rew), fx[f"e{ep}/steps/rewards"][s].astype(np.int32))
        assert obs.dtype == torch.int32 and rew.dtype == torch.int32 and term.dtype == torch.bool


def test_blackjack_hip_matches_oracle_with_device_shuffle():
    """Free-running mode: decks shuffled on device (Philox Fisher-Yates); the oracle replays the same decks."""
    from pulselib_amd.environments.blackjack import BlackJack
    B = 100000
    env = BlackJack(torch.device(DEV), B, seed=9)
    ref = orc.OracleBlackjack(B)
    rng = np.random.default_rng(1)
    for ep in range(2):
        env.reset()
        decks = _np(env.decks)
        assert np.array_equal(np.sort(decks, axis=1), np.tile(np.arange(52, dtype=np.int32), (B, 1)))
        ref.reset(decks)
        np.testing.assert_array_equal(_np(env.obs), ref.obs)
        for s in range(10):
            a = rng.integers(0, 2, B).astype(np.int64)
            obs, rew, term, _, _ = env.step(torch.from_numpy(a))
            robs, rrew, rterm = ref.step(a)
            np.testing.assert_array_equal(_np(obs), robs)
            np.testing.assert_array_equal(_np(rew), rrew)
            np.testing.assert_array_equal(_np(term), rterm)
            for n in ("player_card_sums", "dealer_card_sums", "deck_positions", "players_cards", "dealer_cards"):
                np.testing.assert_array_equal(_np(getattr(env, n)), getattr(ref, n), err_msg=n)
        assert rterm.all()
    top = np.bincount(decks[:, 0], minlength=52)
    assert top.min() > 1600 and top.max() < 2250          # mean 1923: every card about equally likely on top


def test_tfe_hip_matches_reference(golden_dir):
    from pulselib_amd.environments.TFE import TFEBatch
    fx = np.load(golden_dir / "tfe.npz")
    want = fx["boards"].astype(np.int32)
    steps, B = fx["actions"].shape
    env = TFEBatch(torch.device(DEV), B, want.shape[-1], seed=int(fx["seed"]))
    boards, info = env.reset()
    np.testing.assert_array_equal(_np(boards), want[0])
    for s in range(steps):
        boards, rew, dones, trunc, info = env.step(torch.from_numpy(fx["actions"][s].astype(np.int64)))
        np.testing.assert_array_equal(_np(boards), want[s + 1], err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(rew), fx["rewards"][s].astype(np.int32), err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(dones).astype(np.uint8), fx["dones"][s], err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(info["score"]), fx["scores"][s].astype(np.int64), err_msg=f"step {s}")


@pytest.mark.parametrize("n,B", [(4, 262144), (3, 1000), (5, 777), (2, 500), (6, 300), (8, 129)])
def test_tfe_hip_matches_oracle_at_scale(n, B):
    """BASELINE.json config 3 size (262,144 boards of 4x4: the packed-board kernel) plus the other board sides (3, 5: boards
    in registers; 2, 6, 8: the any-side kernel -- the reference takes any board_height, TFE.py:112-131)."""
    from pulselib_amd.environments.TFE import TFEBatch
    env = TFEBatch(torch.device(DEV), B, n, seed=31, board_id0=1000)
    boards = np.zeros((B, n, n), dtype=np.int32)
    score = np.zeros(B, dtype=np.int64)
    rewards = np.zeros(B, dtype=np.int32)
    dones = np.zeros(B, dtype=np.uint8)
    env.reset()
    orc.tfe_reset(boards, score, n, 31, board_id0=1000)
    np.testing.assert_array_equal(_np(env.boards), boards)
    rng = np.random.default_rng(2)
    for s in range(60 if B < 10000 else 25):
        a = rng.integers(0, 4, B).astype(np.int64)
        b, r, d, _, info = env.step(torch.from_numpy(a))
        orc.tfe_step(boards, score, a, rewards, dones, n, 31, s + 1, board_id0=1000)
        np.testing.assert_array_equal(_np(b), boards, err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(r), rewards)
        np.testing.assert_array_equal(_np(d).astype(np.uint8), dones)
        np.testing.assert_array_equal(_np(info["score"]), score)
    # size-independent invariant: tile mass only grows by the spawned tile (2 or 4) per step
    assert (_np(env.boards).reshape(B, -1).sum(1) >= 4).all()


def test_tfe_packed_kernel_steps_unusual_boards_cell_by_cell():
    """The 4 x 4 kernel runs on the board packed to 4-bit log2 tiles; boards that form cannot hold -- a 1, a tile that is no
    power of two, a negative, 32,768 and above -- must come out as the reference's cell-by-cell rule gives them (oracle),
    lane by lane inside wavefronts whose other boards are ordinary ones, and 16,384 + 16,384 must merge to 32,768."""
    from pulselib_amd.environments.TFE import TFEBatch
    B, n = 4096, 4
    env = TFEBatch(torch.device(DEV), B, n, seed=9, board_id0=77)
    env.reset()
    rng = np.random.default_rng(4)
    boards = (2 ** rng.integers(1, 12, (B, n, n))).astype(np.int32) * (rng.random((B, n, n)) < 0.7)
    odd = rng.choice(B, 600, replace=False)
    for i, bd in enumerate(odd):                      # one unusual cell each, all kinds, spread over the wavefronts
        r, c = rng.integers(0, 4, 2)
        boards[bd, r, c] = (1, 3, 6, -2, 32768, 65536, 2 ** 20, 12, 16384, 5)[i % 10]
    boards[5] = np.array([[16384, 16384, 16384, 16384], [2, 2, 4, 4], [0, 0, 0, 0], [8192, 8192, 0, 4096]])
    boards = boards.astype(np.int32)
    env.boards.copy_(torch.from_numpy(boards))
    score = np.zeros(B, dtype=np.int64); rewards = np.zeros(B, dtype=np.int32); dones = np.zeros(B, dtype=np.uint8)
    for s in range(12):
        a = rng.integers(0, 4, B).astype(np.int64)
        if s == 0:
            a[5] = 0
        b, r, d, _, info = env.step(torch.from_numpy(a))
        orc.tfe_step(boards, score, a, rewards, dones, n, 9, s + 1, board_id0=77)
        np.testing.assert_array_equal(_np(b), boards, err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(r), rewards, err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(d).astype(np.uint8), dones, err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(info["score"]), score, err_msg=f"step {s}")
        if s == 0:
            assert boards[5, 0].tolist()[:2] == [32768, 32768] and score[5] >= 65536 + 4 + 8 + 16384


def test_tfe_single_board_wrapper_keeps_reference_signature():
    from pulselib_amd.environments.TFE import TFE
    env = TFE(4, 4, device=DEV, seed=5)
    obs, info = env.reset()
    assert obs.shape == (4, 4) and obs.dtype == np.int32 and (obs > 0).sum() == 2 and info == {"score": 0}
    obs, reward, done, truncated, info = env.step(2)
    assert isinstance(reward, int) and isinstance(done, bool) and truncated is False and "score" in info


def test_particle2d_hip_matches_reference(golden_dir):
    from pulselib_amd.environments.Particle2D import Particle2D
    fx = np.load(golden_dir / "particle2d.npz")
    B = fx["state0"].shape[0]
    env = Particle2D(torch.device(DEV), B, dt=0.1, max_steps=20)
    env.reset(options={"state": torch.from_numpy(fx["state0"])})
    for s in range(fx["actions"].shape[0]):
        obs, rew, term, trunc, _ = env.step(torch.from_numpy(fx["actions"][s]))
        # fp32 tolerance (rtol 1e-6); integers exact
        np.testing.assert_allclose(_np(obs), fx["obs"][s], rtol=1e-6, atol=1e-7, err_msg=f"step {s}")
        np.testing.assert_allclose(_np(rew), fx["rewards"][s], rtol=1e-6, atol=1e-6, err_msg=f"step {s}")
        np.testing.assert_array_equal(_np(term).astype(np.uint8), fx["terminated"][s], err_msg=f"step {s}")
        assert not trunc.any()
    np.testing.assert_array_equal(_np(env.steps), fx["steps_final"])


def test_particle2d_hip_matches_oracle_one_million():
    """BASELINE.json config 5: 1M particles; HIP and oracle use the same op order, so results are bit-equal."""
    from pulselib_amd.environments.Particle2D import Particle2D
    B = 1 << 20
    rng = np.random.default_rng(0)
    state0 = np.concatenate([5 * rng.standard_normal((B, 2)), np.zeros((B, 2))], axis=1).astype(np.float32)
    env = Particle2D(torch.device(DEV), B)
    env.reset(options={"state": torch.from_numpy(state0)})
    state, steps = state0.copy(), np.zeros(B, dtype=np.int32)
    for s in range(5):
        a = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
        obs, rew, term, _, _ = env.step(torch.from_numpy(a))
        robs, rrew, rterm = orc.particle2d_step(state, a, steps, 0.1, 200)
        np.testing.assert_array_equal(_np(obs), robs)
        np.testing.assert_array_equal(_np(rew), rrew)
        np.testing.assert_array_equal(_np(term), rterm)


def test_particle2d_step_outputs_are_fresh_unless_reuse_is_asked_for():
    """The reference returns new tensors from every step (state.clone(), Particle2D.py:26-30): a caller may keep them.
    Default: no output of an earlier step is ever rewritten.  reuse_outputs (opt-in): two persistent sets alternate, so
    an output survives exactly one further step."""
    from pulselib_amd.environments.Particle2D import Particle2D
    B = 4096
    env = Particle2D(torch.device(DEV), B)
    env.reset(seed=3)
    a = torch.rand((B, 2), device=DEV) * 2 - 1
    kept = [env.step(a)[:3] for _ in range(4)]
    copies = [[x.clone() for x in k] for k in kept]
    env.step(a)
    assert len({k[0].data_ptr() for k in kept}) == 4
    for k, c in zip(kept, copies):
        assert all(torch.equal(x, y) for x, y in zip(k, c))
    env.reuse_outputs = True
    o1 = env.step(a)[0]; c1 = o1.clone()
    o2 = env.step(a)[0]
    assert torch.equal(o1, c1) and o2.data_ptr() != o1.data_ptr()
    o3 = env.step(a)[0]
    assert o3.data_ptr() == o1.data_ptr() and not torch.equal(o3, c1)


def _pack_board(b):
    key = 0
    for i, v in enumerate(b.reshape(-1)):
        e = min(int(v).bit_length() - 1, 15) if v > 0 else 0
        key |= e << (4 * i)
    return key


def test_batched_q_learning_matches_reference_semantics_per_board():
    """BASELINE.json config 3 (2048 + tabular Q-learning): with private tables every board is an independent
    copy of the reference agent (QLearningNumba.py + utils/numba.py); the oracle replays it with a Python dict
    per board and the oracle's scalar helpers, same Philox draws -> actions, boards and Q tables bit-equal."""
    import ctypes as C
    from pulselib_amd.agents import QLearningBatch
    from pulselib_amd.environments.TFE import TFEBatch
    B, n, steps = 384, 4, 60
    cfg = {"ALPHA": 0.1, "GAMMA": 0.99, "EPSILON": 0.1}
    env = TFEBatch(torch.device(DEV), B, n, seed=77)
    agent = QLearningBatch(torch.device(DEV), B, n, config=cfg, private_tables=True, slots=256, seed=991)
    boards_ref = np.zeros((B, n, n), dtype=np.int32)
    score = np.zeros(B, dtype=np.int64)
    rew_ref = np.zeros(B, dtype=np.int32)
    done_ref = np.zeros(B, dtype=np.uint8)
    env.reset()
    orc.tfe_reset(boards_ref, score, n, 77)
    tables = [dict() for _ in range(B)]
    lib = orc.lib()

    def q_of(g, key):
        return tables[g].setdefault(key, np.zeros(4, dtype=np.float64))

    for s in range(steps):
        acts = agent.get_actions(env.boards, s)
        # oracle: epsilon-greedy with the same Philox words (p from words 0,1; randint from word 2)
        a_ref = np.zeros(B, dtype=np.int64)
        keys_s = [None] * B
        for g in range(B):
            w = orc.philox4x32(991, g, s)
            p = float(((int(w[0]) << 21) ^ (int(w[1]) >> 11)) * (1.0 / 9007199254740992.0))
            keys_s[g] = _pack_board(boards_ref[g])
            q = q_of(g, keys_s[g])
            a_ref[g] = lib.oracle_select_action_epsilon_greedy(q.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(p),
                                                               C.c_uint32(int(w[2])))
        np.testing.assert_array_equal(_np(acts), a_ref, err_msg=f"step {s} actions")
        boards, rew, dones, _, _ = env.step(acts)
        orc.tfe_step(boards_ref, score, a_ref, rew_ref, done_ref, n, 77, s + 1)
        np.testing.assert_array_equal(_np(boards), boards_ref, err_msg=f"step {s} boards")
        agent.update(boards, rew, dones)
        for g in range(B):
            cur, nxt = q_of(g, keys_s[g]), q_of(g, _pack_board(boards_ref[g]))
            lib.oracle_update_q_entry(cur.ctypes.data_as(C.c_void_p), int(a_ref[g]), nxt.ctypes.data_as(C.c_void_p), 4,
                                      C.c_double(0.1), C.c_double(float(rew_ref[g])), C.c_double(0.99), int(done_ref[g]))
    for g in (0, 1, 17, B - 1):
        got = agent.table(g)
        assert set(got) == set(tables[g]), f"board {g}: state sets differ"
        for k, v in tables[g].items():
            np.testing.assert_array_equal(got[k], v, err_msg=f"board {g} state {k:#x}")
    assert sum(len(t) for t in tables) > B * 20 and max(float(np.abs(v).max()) for t in tables for v in t.values()) > 0


@pytest.mark.parametrize("private", [True, False], ids=["private-tables", "shared-table"])
def test_fused_rollout_step_equals_select_step_update(private):
    """pulse_qtable_rollout_step (select + 2048 move + update in one launch, the found s' carried to the next step) against the
    three separate calls: same actions, boards, rewards, done flags, scores, and the same table -- bit for bit with private
    tables (no races); with a shared table at a size where boards never meet in a state, too."""
    from pulselib_amd.agents import QLearningBatch
    from pulselib_amd.environments.TFE import TFEBatch
    B, n, steps = 2048, 4, 50
    cfg = {"ALPHA": 0.1, "GAMMA": 0.99, "EPSILON": 0.1}
    dev = torch.device(DEV)
    kw = dict(config=cfg, private_tables=private, slots=128 if private else 1 << 20, seed=991)
    e1, e2 = TFEBatch(dev, B, n, seed=77), TFEBatch(dev, B, n, seed=77)
    a1, a2 = QLearningBatch(dev, B, n, **kw), QLearningBatch(dev, B, n, **kw)
    e1.reset(); e2.reset()
    if not private:
        # give every board a start state of its own (distinct tiles in the last row), so that no two boards ever update one
        # entry in the same launch: the shared table is then race-free and must match exactly
        for e in (e1, e2):
            e.boards[:, 3, :] = 0
            ids = torch.arange(B, device=dev)
            for c in range(4):
                e.boards[:, 3, c] = (2 ** (1 + ((ids >> (3 * c)) & 7))).to(torch.int32)
    for s in range(steps):
        acts = a1.get_actions(e1.boards, s).clone()
        nb, r, d, _, _ = e1.step(acts)
        a1.update(nb, r, d)
        nb2, r2, d2, _, _ = a2.rollout_step(e2, s)
        np.testing.assert_array_equal(_np(a2.actions), _np(acts), err_msg=f"step {s} actions")
        np.testing.assert_array_equal(_np(nb2), _np(nb), err_msg=f"step {s} boards")
        np.testing.assert_array_equal(_np(r2), _np(r), err_msg=f"step {s} rewards")
        np.testing.assert_array_equal(_np(d2), _np(d), err_msg=f"step {s} dones")
        np.testing.assert_array_equal(_np(e2.total_score), _np(e1.total_score))
        if s == 20:                                   # boards changed behind the agent's back: the carried entries must be dropped
            for e in (e1, e2):
                e.boards[::2] = e.boards[::2].flip(1).contiguous()
            a2.forget_states()
    if private:
        np.testing.assert_array_equal(_np(a2.keys), _np(a1.keys))
        np.testing.assert_array_equal(_np(a2.values), _np(a1.values))
    else:
        # Boards start from distinct states, but merges can still lead two of them through one state in the same launch: there
        # the reads of q[s'] race with another board's update of that very entry (in both forms).  All but a handful of the
        # ~100,000 entries must agree exactly.
        t1, t2 = a1.table(), a2.table()
        assert set(t1) == set(t2) and len(t1) > B * 10
        differing = sum(1 for k, v in t1.items() if not np.array_equal(t2[k], v))
        assert differing <= len(t1) // 1000, f"{differing} of {len(t1)} entries differ"
    assert float(_np(a1.values).max()) > 0


def test_fused_rollout_step_at_config3_size():
    """BASELINE config 3's size, the launch bench.py times (`tfe_qlearning_rollout_step`), held to its definition directly:
    (i) private tables (64 slots per board: 20 steps visit at most 41 states), 262,144 boards: the ONE-launch step equals
    get_actions + env.step + update call for call for 20 steps -- actions, boards, rewards, done flags, scores -- and leaves
    bit-identical tables (utils/numba.py:5-39; TFE.py:152-189);
    (ii) the shared table at the step right after a reset, where all 262,144 boards sit in a few hundred two-tile states and
    ~1,000 of them update one entry in the same launch (gamma = 0 and a greedy policy on an empty table: every board of a
    state takes action 0 and has the same target, the move's reward): whatever mix of compare-and-swap winners and combined
    losers an entry saw, its value must be the k updates one after another, t (1 - (1 - alpha)^k) -- no update lost, none
    applied twice (QLearningNumba.py:28-37 per board)."""
    from pulselib_amd.agents import QLearningBatch
    from pulselib_amd.environments.TFE import TFEBatch
    B, n = 262144, 4
    dev = torch.device(DEV)
    cfg = {"ALPHA": 0.1, "GAMMA": 0.99, "EPSILON": 0.1}
    kw = dict(config=cfg, private_tables=True, slots=64, seed=17)
    e1, e2 = TFEBatch(dev, B, n, seed=5), TFEBatch(dev, B, n, seed=5)
    a1, a2 = QLearningBatch(dev, B, n, **kw), QLearningBatch(dev, B, n, **kw)
    e1.reset(); e2.reset()
    for s in range(20):
        acts = a1.get_actions(e1.boards, s).clone()
        nb, r, d, _, _ = e1.step(acts)
        a1.update(nb, r, d)
        nb2, r2, d2, _, _ = a2.rollout_step(e2, s)
        assert torch.equal(a2.actions, acts) and torch.equal(nb2, nb) and torch.equal(r2, r) and torch.equal(d2, d), f"step {s}"
        assert torch.equal(e2.total_score, e1.total_score)
    assert torch.equal(a2.keys, a1.keys) and torch.equal(a2.values.view(torch.int64), a1.values.view(torch.int64))
    assert int((a1.keys != 0).sum()) > 10 * B
    del a1, a2, e1
    torch.cuda.empty_cache()
    # (ii)
    alpha = 0.25
    env = e2
    env.reset()
    agent = QLearningBatch(dev, B, n, config={"ALPHA": alpha, "GAMMA": 0.0, "EPSILON": 0.0}, slots=1 << 22, seed=3)
    start = _np(env.boards).reshape(B, 16).astype(np.int64)
    logs = np.where(start > 0, np.log2(np.maximum(start, 1)).astype(np.int64), 0)
    keys = (logs << (4 * np.arange(16, dtype=np.int64))).sum(axis=1)
    _, rew, _, _, _ = agent.rollout_step(env, 0)
    assert int(agent.actions.abs().max()) == 0                              # greedy on an empty table: the first maximum
    rew = _np(rew).astype(np.float64)
    uniq, inverse, counts = np.unique(keys, return_inverse=True, return_counts=True)
    assert len(uniq) <= 480 and counts.max() > 500
    target = np.zeros(len(uniq)); target[inverse] = rew                     # the same for every board of a state
    assert np.array_equal(target[inverse], rew)
    table = agent.table()
    checked = 0
    for key, k, t in zip(uniq.tolist(), counts.tolist(), target.tolist()):
        want = t * (1.0 - (1.0 - alpha) ** k)
        got = table[key & (2**64 - 1)]
        assert abs(got[0] - want) <= 1e-12 * max(1.0, abs(want)) and got[1] == got[2] == got[3] == 0.0, (hex(key), k, t, got)
        checked += t > 0
    assert checked > 20                                                     # states whose move to the left merges two tiles
    assert int(agent._scratch_tensors["acc_key"].ne(0).sum()) == 0          # every accumulator was applied and freed


def test_shared_table_combines_simultaneous_updates_of_one_entry():
    """Many boards in the SAME state taking the SAME action in one launch (what happens right after reset): one update
    goes through alone, the others are combined per cell -- with equal targets the result is the k + 1 updates applied one
    after another, q0 + (1 - (1 - alpha)^(k+1)) (t - q0), whatever the order (csrc/qtable.hip)."""
    from pulselib_amd.agents import QLearningBatch
    B, n = 4096, 4
    dev = torch.device(DEV)
    agent = QLearningBatch(dev, B, n, config={"ALPHA": 0.25, "GAMMA": 0.0, "EPSILON": 0.0}, slots=1 << 12, seed=1)
    board = torch.tensor([[2, 4, 8, 16], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 2, 0]], dtype=torch.int32, device=dev)
    boards = board.repeat(B, 1, 1).contiguous()
    nxt = boards.clone(); nxt[:, 1, 1] = 2
    rewards = torch.full((B,), 3, dtype=torch.int32, device=dev)
    term = torch.zeros(B, dtype=torch.bool, device=dev)
    q = 0.0
    for rnd in range(3):
        acts = agent.get_actions(boards, rnd)
        assert int(acts.min()) == int(acts.max())                      # greedy on equal rows: everybody takes the first maximum
        a = int(acts[0])
        agent.update(nxt, rewards, term)
        q = q + (1.0 - 0.75 ** B) * (3.0 - q)                          # B transitions with target 3 (gamma = 0)
        tab = agent.table()
        key = [k for k, v in tab.items() if v[a] != 0.0]
        assert len(key) == 1
        assert abs(tab[key[0]][a] - q) < 1e-12 and all(tab[key[0]][i] == 0.0 for i in range(4) if i != a)
    assert int(agent._scratch_tensors["acc_key"].ne(0).sum()) == 0 and int(agent._scratch_tensors["acc_cnt"].sum()) == 0      # scratch left clean


def test_a_called_off_follow_up_launch_drops_its_updates_as_a_whole_and_fails_the_next_call():
    """The follow-up launch that combines a LONG list of deferred updates spreads it over 64 workgroups that meet at a
    counter (csrc/qtable.hip).  With the wait cut to 30 ms and one arrival more expected than the grid has (test hook)
    the launch returns, NO combined update was applied (only the one compare-and-swap winner's own), the next call on
    the table fails with PULSE_EINTERNAL once and clears the accumulators, and the table works again afterwards."""
    import time
    from pulselib_amd.agents import QLearningBatch
    B, n = 4096, 4                                                 # 4,095 deferred transitions > 1,024: the spread form
    dev = torch.device(DEV)
    agent = QLearningBatch(dev, B, n, config={"ALPHA": 0.25, "GAMMA": 0.0, "EPSILON": 0.0}, slots=1 << 12, seed=1)
    board = torch.tensor([[2, 4, 8, 16], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 2, 0]], dtype=torch.int32, device=dev)
    boards = board.repeat(B, 1, 1).contiguous()
    nxt = boards.clone(); nxt[:, 1, 1] = 2
    rewards = torch.full((B,), 3, dtype=torch.int32, device=dev)
    term = torch.zeros(B, dtype=torch.bool, device=dev)
    agent._scratch.wait_ticks, agent._scratch.debug_meet_extra = 3_000_000, 1
    acts = agent.get_actions(boards, 0)
    a = int(acts[0])
    t0 = time.perf_counter()
    agent.update(nxt, rewards, term)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 2.0                          # 30 ms, not the default 3 s
    tab = agent.table()
    key = [k for k, v in tab.items() if v[a] != 0.0]
    W = B // 256                                                   # launch 0: the segment lengths of parity 0 (pulse_env.h: PulseQTableScratch.count)
    deferred = int(agent._scratch_tensors["count"][64:64 + W].sum())   # transitions that lost their compare-and-swap (> 1,024: the spread form)
    assert 1024 < deferred < B
    q0 = 3.0 * (1.0 - 0.75 ** (B - deferred))                      # the winners' updates, one after another -- and nothing half-combined
    assert len(key) == 1 and abs(tab[key[0]][a] - q0) < 1e-12
    assert int(agent._scratch_tensors["acc_key"].ne(0).sum()) > 0              # the accumulators stayed claimed ...
    agent._scratch.debug_meet_extra = 0
    with pytest.raises(RuntimeError, match="could not gather"):
        agent.get_actions(boards, 1); agent.update(nxt, rewards, term)
    torch.cuda.synchronize()
    assert int(agent._scratch_tensors["acc_key"].ne(0).sum()) == 0 and int(agent._scratch_tensors["acc_cnt"].sum()) == 0   # ... and are clean now
    agent.get_actions(boards, 2)
    agent.update(nxt, rewards, term)                               # an ordinary launch again: all B transitions land
    q1 = q0 + (1.0 - 0.75 ** B) * (3.0 - q0)
    assert abs(agent.table()[key[0]][a] - q1) < 1e-12


def test_shared_q_table_learns_and_loses_no_update():
    """Shared-table mode at config-3 size: all 262,144 boards start from states with two tiles, so thousands of
    boards update the same entries concurrently; the CAS loop must apply every one of them."""
    from pulselib_amd.agents import QLearningBatch
    from pulselib_amd.environments.TFE import TFEBatch
    B = 262144
    env = TFEBatch(torch.device(DEV), B, 4, seed=5)
    agent = QLearningBatch(torch.device(DEV), B, 4, config={"ALPHA": 0.5, "GAMMA": 0.0, "EPSILON": 1.0}, slots=1 << 22, seed=3)
    env.reset()
    for s in range(8):
        acts = agent.get_actions(env.boards, s)
        boards, rew, dones, _, _ = env.step(acts)
        agent.update(boards, rew, dones)
    torch.cuda.synchronize()
    table = agent.table()
    assert len(table) > 1000
    vals = np.stack(list(table.values()))
    assert np.isfinite(vals).all() and vals.min() >= 0.0 and vals.max() > 0.5      # rewards are >= 0, gamma = 0


def test_blackjack_first_visit_mc_plumbing():
    """BASELINE.json config 1: GPU blackjack batches -> CPU first-visit MC.  Sanity of the estimates under the
    'hit below 17' policy: standing on 20/21 is clearly good, every value lies in [-1, 1]."""
    from pulselib_amd.scripts.blackjack_fvmc import run
    agent, n = run(torch.device(DEV), batches=3, batch_size=1000, gamma=0.9, seed=7)
    assert n == 3000 and len(agent.values) > 50
    assert all(-1.0 <= v <= 1.0 for v in agent.values.values())
    strong = [v for (s, ace, up), v in agent.values.items() if s in (20, 21)]
    weak = [v for (s, ace, up), v in agent.values.items() if s in (14, 15, 16) and not ace]
    assert sum(strong) / len(strong) > 0.5 > sum(weak) / len(weak)
