"""Pins against fixtures RECORDED FROM THE REFERENCE's own code (tests/golden/make_golden.py: make_scripted,
make_qhelpers, make_performance) for the three pieces whose oracles were restatements only:
  a13  scripted opponents + build_actions  (environments/Poker/Player.py:79-176, environments/Poker/utils.py:108-123)
  a18  tabular-Q helpers                   (utils/numba.py:5-39, agents/TemperalDifference/QLearningNumba.py:10-37)
  f2   BB/100 metrics                      (utils/performance.py:104-349)
CPU only: the oracle (C restatement) and the host-side reduction are held to the fixtures bit for bit (fp32-reduced
reference values: to fp32 rounding); the GPU suites hold the HIP kernels to the oracle and to the same fixtures."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

NATIVE = {"random": 1, "heuristic_hands": 2, "tight_aggressive": 3, "loose_passive": 4, "small_ball": 5}
CORNERS = ((0, 0), (0xFFFFFFFF, 0), (0, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF))     # (pick, coin) = fixture's corner order


@pytest.fixture(scope="module")
def scripted(golden_dir):
    return np.load(golden_dir / "scripted.npz")


def test_fixture_corner_order(scripted):
    assert [str(c) for c in scripted["corners"]] == ["lo_coin0", "hi_coin0", "lo_coin1", "hi_coin1"]


@pytest.mark.parametrize("name", ["heuristic_hands", "tight_aggressive", "loose_passive", "small_ball"])
def test_oracle_scripted_policies_equal_the_reference_at_every_forced_draw(scripted, name):
    """With the draw forced to the lowest / highest value of its range and rand() to 0 / 1, the reference's policy is a
    function of (hole cards, pot) only: all 53 x 53 card pairs (incl. -1) x six pots + 3,000 random rows, bit for bit."""
    rows = scripted["rows"].astype(np.int32)
    types = np.full(rows.shape[0], NATIVE[name], dtype=np.uint8)
    for k, (pick, coin) in enumerate(CORNERS):
        got = orc.scripted_actions_rows(types, rows[:, 0], rows[:, 1], rows[:, 2], pick, coin)
        np.testing.assert_array_equal(got, scripted[f"actions/{name}"][k].astype(np.int64), err_msg=f"{name} corner {k}")
    a = scripted[f"actions/{name}"].astype(np.int64)
    assert (a.min(axis=0) != a.max(axis=0)).any() and (a.min(axis=0) == a.max(axis=0)).any()     # both kinds of rows are in the fixture


def test_oracle_build_actions_equals_the_reference(scripted):
    """build_actions: every seat of a type is served by that type's policy, `random` draws randint(0, 13) inline."""
    rows = scripted["rows"].astype(np.int32)
    names = [str(x) for x in scripted["build/type_names"]]
    seat = scripted["build/seat_idx"]
    types = np.array([NATIVE[names[s]] for s in seat], dtype=np.uint8)
    for k, (pick, coin) in enumerate(CORNERS):
        got = orc.scripted_actions_rows(types, rows[:, 0], rows[:, 1], rows[:, 2], pick, coin)
        np.testing.assert_array_equal(got, scripted["build/actions"][k].astype(np.int64), err_msg=f"corner {k}")


def test_oracle_policy_draws_come_from_the_documented_stream():
    """oracle_policy_draw: one Philox call per two steps, (x, y) for the even step and (z, w) for the odd one."""
    lib = orc.lib()
    out = (C.c_uint32 * 2)()
    for step in (0, 1, 6, 7, 2**33 + 5):
        call = orc.philox4x32(99, 1234, step >> 1)
        lib.oracle_policy_draw(C.c_uint64(99), C.c_uint64(1234), C.c_uint64(step), out)
        assert (out[0], out[1]) == ((int(call[2]), int(call[3])) if step & 1 else (int(call[0]), int(call[1])))


# ---------------------------------------------------------------------------------------------- a18
@pytest.fixture(scope="module")
def qh(golden_dir):
    return np.load(golden_dir / "qhelpers.npz")


@pytest.mark.parametrize("n", [4, 13])
def test_oracle_q_helpers_equal_utils_numba(qh, n):
    lib = orc.lib()
    q, eps, p, word = qh[f"n{n}/q"], qh[f"n{n}/eps"], qh[f"n{n}/p"], qh[f"n{n}/word"]
    got = np.array([lib.oracle_select_action_epsilon_greedy(np.ascontiguousarray(q[i]).ctypes.data_as(C.c_void_p), n, C.c_double(eps[i]),
                                                            C.c_double(p[i]), C.c_uint32(int(word[i]))) for i in range(q.shape[0])])
    np.testing.assert_array_equal(got, qh[f"n{n}/selected"])
    cur = qh[f"n{n}/cur"].copy()
    for i in range(cur.shape[0]):
        lib.oracle_update_q_entry(cur[i].ctypes.data_as(C.c_void_p), int(qh[f"n{n}/action"][i]),
                                  np.ascontiguousarray(qh[f"n{n}/nxt"][i]).ctypes.data_as(C.c_void_p), n, C.c_double(qh[f"n{n}/alpha"][i]),
                                  C.c_double(qh[f"n{n}/reward"][i]), C.c_double(qh[f"n{n}/gamma"][i]), int(qh[f"n{n}/terminal"][i]))
    np.testing.assert_array_equal(cur, qh[f"n{n}/after"])            # float64, same operation order: bit-equal


def test_q_agent_sequence_equals_qlearning_numba(qh):
    """The dict-of-rows plumbing of QLearningNumba.py:10-37 replayed with the oracle's two helpers: same actions, same table."""
    lib = orc.lib()
    table = {}
    row = lambda s: table.setdefault(int(s), np.zeros(4, dtype=np.float64))
    states = qh["agent/states"]
    for t in range(len(qh["agent/actions"])):
        cur = row(states[t])
        a = lib.oracle_select_action_epsilon_greedy(cur.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1), C.c_double(qh["agent/p"][t]),
                                                    C.c_uint32(int(qh["agent/word"][t])))
        assert a == qh["agent/actions"][t], t
        nxt = row(states[t + 1])
        lib.oracle_update_q_entry(cur.ctypes.data_as(C.c_void_p), int(a), nxt.ctypes.data_as(C.c_void_p), 4, C.c_double(0.1),
                                  C.c_double(float(qh["agent/rewards"][t])), C.c_double(0.99), int(qh["agent/terminal"][t]))
    assert sorted(table) == qh["agent/keys"].tolist()
    np.testing.assert_array_equal(np.stack([table[int(k)] for k in qh["agent/keys"]]), qh["agent/table"])


# ---------------------------------------------------------------------------------------------- f2
def test_hand_metric_summaries_equal_utils_performance(golden_dir):
    """The per-(position, street, player count, mix) sums kept on the device reduce to the reference's numbers:
    utils/performance.py run on the per-hand lists (fp32 tensors there; exact integers here -> agreement to fp32 rounding)."""
    from pulselib_amd.utils.performance import accumulate_hands, summarize_totals
    fx = np.load(golden_dir / "performance.npz")
    totals = accumulate_hands(fx["delta"], fx["stage"], fx["position"], fx["count"], fx["mix"])
    s = summarize_totals(totals)
    ref = dict(zip([str(n) for n in fx["final/scalar_names"]], fx["final/scalars"]))
    rel = lambda got, want: abs(got - want) <= 2e-6 * max(1.0, abs(want))
    assert s["total_hands"] == int(ref["total_hands"]) == fx["delta"].size
    assert s["total_bb_won"] == float(fx["delta"].sum()) and rel(s["total_bb_won"], ref["total_bb_won"])
    for k in ("field_bb_per_100", "lcb95_bb_per_100", "seat_balanced_bb_per_100", "overall_hand_win_rate"):
        assert rel(s[k], ref[k]), (k, s[k], ref[k])
    for name, want in zip(fx["final/street_names"], fx["final/street_win"]):
        assert rel(s["street_win_percentages"][str(name)], want), name
    assert [str(p) for p in fx["final/position_names"]] == list(s["position_win_rates"])
    for name, (hands, rate) in zip(fx["final/position_names"], fx["final/position_rows"]):
        r = s["position_win_rates"][str(name)]
        assert r["hands"] == int(hands) and rel(r["win_rate"], rate)
    for fam in ("opponent_mix", "seat", "player_count", "street_depth"):
        names = [str(x) for x in fx[f"final/slice/{fam}/names"]]
        assert names == list(s["slices"][fam]), fam
        for name, want in zip(names, fx[f"final/slice/{fam}/values"]):
            assert rel(s["slices"][fam][name], want), (fam, name)
    assert rel(s["worst_slice"]["bb_per_100"], ref["worst_slice_bb_per_100"])
    assert (s["worst_slice"]["family"], s["worst_slice"]["slice"]) == (str(fx["final/worst_family"]), str(fx["final/worst_slice"]))
    # per-episode summaries (summarize_episode_performance_metrics :138-167)
    for ep, want in enumerate(fx["episode_summaries"]):
        m = fx["episode"] == ep
        d = fx["delta"][m]
        got = [d.mean(), (d > 0).mean(), d.size, 100 * d.mean()] if d.size else [0, 0, 0, 0]
        assert all(rel(g, w) for g, w in zip(got, want)), ep


def test_rolling_window_averages_equal_utils_performance(golden_dir):
    """f2, the last output of utils/performance.py HandMetrics did not reproduce: the rolling-window mean
    BB delta (utils/performance.py:128-135, reported at :452-457), on the reference's own per-hand order; fixture values are
    the reference's (make_golden.py: make_performance, window 50)."""
    import torch
    from pulselib_amd.utils.performance import calculate_rolling_window_averages, rolling_bb_window_summary
    fx = np.load(golden_dir / "performance.npz")
    deltas = torch.from_numpy(fx["delta"].astype(np.float32))
    W = int(fx["final/rolling/window_size"])
    per_episode = [deltas[torch.from_numpy(fx["episode"] == e)] for e in np.unique(fx["episode"])]     # a list of batches, as the trainer keeps them
    for form in (deltas, per_episode):
        got = rolling_bb_window_summary(form, W)
        assert got["num_windows"] == int(fx["final/rolling/num_windows"]) == deltas.numel() - W + 1
        np.testing.assert_allclose(got["values"], fx["final/rolling/values"], rtol=1e-6, atol=1e-6)
        assert abs(got["last_average"] - float(fx["final/rolling/last_average"])) < 1e-5
        assert abs(got["best_average"] - float(fx["final/rolling/best_average"])) < 1e-5
    assert calculate_rolling_window_averages(deltas[:W - 1], window_size=W).numel() == 0          # fewer hands than the window: []
    assert calculate_rolling_window_averages(deltas, window_size=0).numel() == 0


def test_prefixed_deck_batch_is_the_reference_construction(golden_dir):
    """utils/performance.py:62-67 (seed 20260401 + episode, trainGPU_performance.py:52): the decks of the recorded
    roll-outs were built by the generator script with the same construction from the same torch generator."""
    import torch
    from pulselib_amd.utils.performance import build_prefixed_deck_batch
    d = build_prefixed_deck_batch(n_games=64, seed=20260401, device=torch.device("cpu"))
    assert d.dtype == torch.int32 and tuple(d.shape) == (64, 52)
    assert bool((d.sort(dim=1).values == torch.arange(1, 53, dtype=torch.int32)).all())
    g = torch.Generator(device="cpu"); g.manual_seed(20260401)
    assert torch.equal(d, (torch.rand((64, 52), generator=g).argsort(dim=1) + 1).to(torch.int32))
