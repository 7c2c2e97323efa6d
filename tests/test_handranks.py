"""The 2+2 table: product generator (csrc/handranks_gen.cpp, closed-form evaluator) vs the oracle's
independent restatement (oracle/handranks_oracle.c, 21-subset brute force), the constants the reference
hard-codes (PokerGPU.py:13-18), the golden lookup vectors, and a from-scratch combinatorial evaluator."""
import hashlib
import itertools

import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(scope="module")
def product_table():
    from pulselib_amd import handranks
    return handranks.generate()


def test_product_generator_equals_oracle_table_bit_for_bit(product_table, oracle_table):
    assert product_table.shape == oracle_table.shape == (32487834,)       # PokerGPU.py:51-57: 129,951,336 bytes / 4
    assert np.array_equal(product_table, oracle_table)


def test_table_digest_matches_golden(product_table, golden_dir):
    vec = np.load(golden_dir / "handranks_vectors.npz")
    assert hashlib.sha256(product_table.tobytes()).hexdigest() == str(vec["sha256"])


def _card(s):
    return "23456789TJQKA".index(s[0]) * 4 + "cdhs".index(s[1]) + 1


def _walk(hr, cards):
    p = 53
    for c in cards:
        p = hr[p + c]
    return int(p)


def test_reference_constants(oracle_table):
    hr = oracle_table
    assert _walk(hr, [_card(c) for c in ("As", "Ks", "Qs", "Js", "Ts", "2c", "3d")]) == 36874      # MAX_EQUITY_RANK
    assert _walk(hr, [_card(c) for c in ("9c", "8d", "7h", "5s", "4c", "3d", "2h")]) == 4145       # MIN_EQUITY_RANK
    assert hr[_walk(hr, [_card(c) for c in ("8d", "7h", "5s", "4c", "3d", "2h")])] == 4109         # MIN_TURN_RIVER_EQUITY
    assert hr[_walk(hr, [_card(c) for c in ("7h", "5s", "4c", "3d", "2h")])] == 4097
    # flop "equity" HR[HR[p5]] over every 5-card state (PokerGPU.py:521): max 823779, min over real
    # successor offsets 74359 (entries 0 / 53 are the impossible-hand markers)
    starts = np.arange(53, hr.size, 53)
    slot0 = hr[starts]
    nz = starts[slot0 != 0]
    succ_max = np.max(np.stack([hr[nz + 1 + k] for k in range(52)]), axis=0)
    five = nz[succ_max > 40000]
    assert five.size == 152607
    dbl = hr[hr[five]]
    assert dbl.max() == 823779                                                                    # MAX_FLOP_EQUITY
    assert dbl[dbl > 53].min() == 74359                                                           # MIN_FLOP_EQUITY


def test_lookup_vectors(oracle_table, golden_dir):
    vec = np.load(golden_dir / "handranks_vectors.npz")
    hands = vec["hands"].astype(np.int32)
    np.testing.assert_array_equal(orc.eval_hands(oracle_table, hands), vec["rank7"])
    np.testing.assert_array_equal(orc.eval_hands(oracle_table, hands[:, :6].copy()), vec["rank6"])
    np.testing.assert_array_equal(orc.eval_hands(oracle_table, hands[:, :5].copy()), vec["rank5"])


def _best5_value(cards):
    """Independent evaluator: category << 12 | index-in-category computed by enumerating every 5-card
    hand class in strength order (no table, no Cactus-Kev numbers)."""
    best = None
    for combo in itertools.combinations(cards, 5):
        ranks = sorted(((c - 1) // 4 for c in combo), reverse=True)
        suits = [(c - 1) % 4 for c in combo]
        flush = len(set(suits)) == 1
        cnt = sorted(((ranks.count(r), r) for r in set(ranks)), reverse=True)
        uniq = sorted(set(ranks), reverse=True)
        straight = None
        if len(uniq) == 5:
            if uniq[0] - uniq[4] == 4:
                straight = uniq[0]
            elif uniq == [12, 3, 2, 1, 0]:
                straight = 3
        if straight is not None and flush: key = (9, straight)
        elif cnt[0][0] == 4: key = (8, cnt[0][1], cnt[1][1])
        elif cnt[0][0] == 3 and cnt[1][0] == 2: key = (7, cnt[0][1], cnt[1][1])
        elif flush: key = (6, *ranks)
        elif straight is not None: key = (5, straight)
        elif cnt[0][0] == 3: key = (4, cnt[0][1], *sorted((r for c, r in cnt[1:]), reverse=True))
        elif cnt[0][0] == 2 and cnt[1][0] == 2: key = (3, max(cnt[0][1], cnt[1][1]), min(cnt[0][1], cnt[1][1]), cnt[2][1])
        elif cnt[0][0] == 2: key = (2, cnt[0][1], *sorted((r for c, r in cnt[1:]), reverse=True))
        else: key = (1, *ranks)
        if best is None or key > best:
            best = key
    return best


def test_table_orders_hands_like_an_independent_evaluator(oracle_table):
    """Random 7-card hands: the table's category equals the combinatorial category, and the table value is
    a strictly monotone function of true hand strength (ties <-> equal values)."""
    rng = np.random.default_rng(123)
    hands = [list(rng.permutation(52)[:7] + 1) for _ in range(1500)]
    keys = [_best5_value(h) for h in hands]
    vals = [_walk(oracle_table, h) for h in hands]
    for k, v in zip(keys, vals):
        assert v >> 12 == k[0]
    order = sorted(range(len(hands)), key=lambda i: keys[i])
    for a, b in zip(order, order[1:]):
        if keys[a] == keys[b]:
            assert vals[a] == vals[b]
        else:
            assert vals[a] < vals[b]
