/*
 * oracle/poker_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Scalar, one-table-at-a-time CPU restatement of the reference's batched hold'em state machine,
 * /root/reference/environments/Poker/PokerGPU.py.  Each function cites the reference lines it
 * follows.  It is (1) the checker the GPU parity tests compare the HIP path against on the GPU
 * box, where the Python reference cannot travel, and (2) the "port" CPU baseline timed by
 * bench.py.  It is pinned against the real reference by tests/golden/ (fixtures produced by
 * importing PokerGPU.py in the build container, see tests/golden/make_golden.py).
 *
 * All integer state is bit-exact.  fp32 arithmetic follows torch eager op order (one rounding
 * per op, no FMA contraction: build with -ffp-contract=off); tanh is evaluated in double and
 * rounded once to float (|diff| to torch's fp32 tanh <= 1 ulp).
 *
 * Layout = the reference's own SoA tensors (row-major, int32 unless noted), host pointers.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <limits.h>

enum { ACTIVE = 0, FOLDED = 1, ALLIN = 2, SITOUT = 3 };   /* PokerGPU.py:11 */

typedef struct {
    int32_t n_games, n_players, active_players, obs_size;
    int64_t hr_len;
    const int32_t* hand_ranks;
    /* [N] */
    int32_t *pots, *stages, *deck_positions, *button, *sb, *bb, *idx, *highest, *agg, *acted,
            *last_raise_size, *prev_stacks, *prev_invested, *raise_amounts;
    uint8_t *is_done, *equity_dirty, *is_round_over;
    /* [N,P] */
    int32_t *stacks, *current_round_bet, *total_invested, *status;
    int32_t *hands;      /* [N,P,2] */
    int32_t *board;      /* [N,5]   */
    int32_t *decks;      /* [N,52]  */
    float   *equities;   /* [N,A]   */
    float   *obs;        /* [N,obs_size] */
    float w1, w2; int32_t K, alpha;
} OraclePoker;

static inline int pymod(int x, int m) { int r = x % m; return r < 0 ? r + m : r; }

static inline int32_t hr_at(const OraclePoker* v, int64_t i) {
    if (i < 0) i += v->hr_len;                 /* torch negative-index wrap */
    if (i < 0 || i >= v->hr_len) return 0;     /* out of range: the reference would raise */
    return v->hand_ranks[i];
}

/* PokerGPU.py:437-444 / :477-478 / :497-498 / :518-519 : p = HR[p + card], start 53 */
static int32_t walk(const OraclePoker* v, const int32_t* cards, int n) {
    int32_t p = 53;
    for (int i = 0; i < n; i++) p = hr_at(v, (int64_t)p + cards[i]);
    return p;
}

/* PokerGPU.py:159-179 get_obs */
void oracle_get_obs(const OraclePoker* v, int t) {
    const int P = v->n_players, A = v->active_players;
    float* o = v->obs + (size_t)t * v->obs_size;
    const int idx = v->idx[t];
    for (int j = 0; j < 5; j++) o[j] = (float)v->board[t * 5 + j];
    o[5] = (float)v->hands[((size_t)t * P + idx) * 2 + 0];
    o[6] = (float)v->hands[((size_t)t * P + idx) * 2 + 1];
    o[7] = (float)v->stages[t];
    o[8] = (float)pymod(idx - v->button[t], A);
    o[9] = (float)v->pots[t];
    o[10] = (float)(v->highest[t] - v->current_round_bet[(size_t)t * P + idx]);
    o[11] = (float)v->stacks[(size_t)t * P + idx];
    o[12] = (float)v->status[(size_t)t * P + idx];
    for (int j = 13; j < v->obs_size; j++) o[j] = 0.0f;
    for (int k = 0; k < A - 1; k++) {
        int seat = pymod(idx + 1 + k, A);
        o[13 + 3 * k + 0] = (float)v->stacks[(size_t)t * P + seat];
        o[13 + 3 * k + 1] = (float)v->status[(size_t)t * P + seat];
        o[13 + 3 * k + 2] = (float)v->current_round_bet[(size_t)t * P + seat];
    }
}

/* PokerGPU.py:455-525 calculate_equities, one dirty table */
void oracle_calculate_equities(const OraclePoker* v, int t) {
    if (!v->equity_dirty[t]) return;
    const int P = v->n_players, A = v->active_players;
    const int stage = v->stages[t];
    const int32_t* b = v->board + (size_t)t * 5;
    for (int s = 0; s < A; s++) {
        float e = 0.5f;                                                    /* :461 */
        int32_t c[7];
        c[0] = v->hands[((size_t)t * P + s) * 2]; c[1] = v->hands[((size_t)t * P + s) * 2 + 1];
        if (stage == 3) {                                                  /* :463-481 */
            for (int j = 0; j < 5; j++) c[2 + j] = b[j];
            float r = (float)walk(v, c, 7);
            e = (r - 4109.0f) / (float)32765;
            e = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
        } else if (stage == 2) {                                           /* :483-502 */
            for (int j = 0; j < 4; j++) c[2 + j] = b[j];
            float r = (float)hr_at(v, walk(v, c, 6));
            e = (r - 4109.0f) / (float)32765;
            e = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
        } else if (stage == 1) {                                           /* :504-523 */
            for (int j = 0; j < 3; j++) c[2 + j] = b[j];
            float r = (float)hr_at(v, hr_at(v, walk(v, c, 5)));
            e = (r - 74359.0f) / (float)749420;
            e = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
        }
        v->equities[(size_t)t * A + s] = e;
    }
    v->equity_dirty[t] = 0;                                                /* :525 */
}

/* PokerGPU.py:230-303 execute_actions, one table */
void oracle_execute_actions(const OraclePoker* v, int t, int64_t action) {
    static const float fr[9] = {0.25f, 0.33f, 0.50f, 0.75f, 1.00f, 1.50f, 2.00f, 3.00f, 4.00f}; /* :39 */
    const int P = v->n_players;
    const int idx = v->idx[t];
    int32_t* stack = &v->stacks[(size_t)t * P + idx];
    int32_t* bet   = &v->current_round_bet[(size_t)t * P + idx];
    int32_t* inv   = &v->total_invested[(size_t)t * P + idx];
    int32_t* st    = &v->status[(size_t)t * P + idx];
    const int32_t call_cost = v->highest[t] - *bet;                                          /* :232 */
    const int active = (*st != FOLDED) && (*st != ALLIN) && (*st != SITOUT) && !v->is_done[t]; /* :233 */
    v->raise_amounts[t] = 0;                                                                  /* :256 */
    if (!active) return;
    if (action == 0) {                                                                        /* :236-238 */
        *st = FOLDED; v->acted[t] += 1;
    } else if (action == 1) {                                                                 /* :241-252 */
        int32_t amt = call_cost < *stack ? call_cost : *stack;
        *stack -= amt; *bet += amt; *inv += amt; v->pots[t] += amt;
        if (*stack == 0) *st = ALLIN;
        v->acted[t] += 1;
    } else if (action >= 2) {                                                                 /* :255-303 */
        int32_t raise_amt = 0;
        if (action == 2) raise_amt = v->last_raise_size[t];                                   /* :258-259 */
        if (action == 12) raise_amt = *stack;                                                 /* :262-264 */
        if (action >= 3 && action <= 11) {                                                    /* :267-270 */
            volatile float prod = (float)v->pots[t] * fr[action - 3];
            raise_amt = (int32_t)prod;
        }
        v->raise_amounts[t] = raise_amt;
        int32_t total = call_cost + raise_amt;                                                /* :272 */
        int32_t actual = total < *stack ? total : *stack;                                     /* :273 */
        int is_raise = !(actual <= call_cost);                                                /* :274-275 */
        *stack -= actual; *bet += actual; *inv += actual; v->pots[t] += actual;               /* :277-280 */
        if (*stack == 0) *st = ALLIN;                                                         /* :282-286 */
        if (is_raise) {                                                                       /* :288-301 */
            int32_t new_bet = *bet;
            int32_t raise_size = new_bet - v->highest[t];
            v->highest[t] = new_bet;
            if (raise_size >= v->last_raise_size[t]) {
                v->agg[t] = idx; v->acted[t] = 0; v->last_raise_size[t] = raise_size;
            }
        }
        v->acted[t] += 1;                                                                     /* :303 */
    }
}

static int count_contenders(const OraclePoker* v, int t) {
    int n = 0;
    for (int s = 0; s < v->n_players; s++) {
        int st = v->status[(size_t)t * v->n_players + s];
        n += (st == ACTIVE) || (st == ALLIN);
    }
    return n;
}

/* PokerGPU.py:305-329 poker_reward_gpu, one table (uses post-step state) */
float oracle_reward(const OraclePoker* v, int t, int64_t action, int actor_idx) {
    const int A = v->active_players;
    float cnt = (float)count_contenders(v, t);                                               /* :309 */
    float fair = 1.0f / (cnt < 1.0f ? 1.0f : cnt);                                           /* :310 */
    int32_t cc = v->highest[t] - v->prev_invested[t]; if (cc < 0) cc = 0;                     /* :311 */
    float e = v->equities[(size_t)t * A + actor_idx];                                         /* :314 */
    float potf = (float)v->pots[t];
    volatile float m = e * potf;                                                              /* :315 */
    volatile float den = (float)(v->pots[t] + cc) + 1e-6f;                                    /* :316 */
    volatile float o = (float)cc / den;
    volatile float s = 0.0f;                                                                  /* :306 */
    if (action == 1)      { volatile float d = e - o;    s = d * potf; }                      /* :326 */
    else if (action == 0) { volatile float d = o - e;    s = d * potf; }                      /* :327 */
    else if (action >= 2) { volatile float d = e - fair; s = d * potf; }                      /* :328 */
    volatile float a1 = v->w1 * m;
    volatile float a2 = v->w2 * s;
    volatile float sum = a1 + a2;
    volatile float x = sum / (float)v->K;
    float th = (float)tanh((double)x);
    return (float)v->alpha * th;                                                              /* :329 */
}

/* PokerGPU.py:215-228 _set_first_active_street_actor, one table */
static void set_first_active_street_actor(const OraclePoker* v, int t) {
    const int P = v->n_players, A = v->active_players;
    for (int k = 1; k <= A; k++) {
        int seat = pymod(v->button[t] + k, A);
        if (v->status[(size_t)t * P + seat] == ACTIVE) { v->idx[t] = seat; return; }
    }
}

/* PokerGPU.py:208-214 deal_cards(g, n) into dst */
static void deal_cards(const OraclePoker* v, int t, int n, int32_t* dst) {
    for (int i = 0; i < n; i++) {
        int pos = v->deck_positions[t] + i;
        dst[i] = (pos >= 0 && pos < 52) ? v->decks[(size_t)t * 52 + pos] : 0;
    }
    v->deck_positions[t] += n;
}

/* PokerGPU.py:331-338 resolve_fold_winners, one table; `ended` = newly done this step */
void oracle_resolve_fold_winner(const OraclePoker* v, int t, int ended) {
    if (!ended) return;
    const int P = v->n_players;
    if (count_contenders(v, t) != 1) return;
    for (int s = 0; s < P; s++) {
        int st = v->status[(size_t)t * P + s];
        if (st == ACTIVE || st == ALLIN) { v->stacks[(size_t)t * P + s] += v->pots[t]; break; }
    }
    v->pots[t] = 0;
}

/* PokerGPU.py:340-378 _award_showdown_side_pots, one table */
static void award_side_pots(const OraclePoker* v, int t, const int32_t* ranks, const int* eligible) {
    const int P = v->n_players, A = v->active_players;
    int32_t inv[16], sorted[16], payout[16];
    for (int s = 0; s < A; s++) { inv[s] = v->total_invested[(size_t)t * P + s]; sorted[s] = inv[s]; payout[s] = 0; }
    for (int i = 1; i < A; i++) { int32_t x = sorted[i]; int j = i - 1; while (j >= 0 && sorted[j] > x) { sorted[j + 1] = sorted[j]; j--; } sorted[j + 1] = x; }
    for (int l = 0; l < A; l++) {
        int32_t level = sorted[l];
        int32_t layer_size = level - (l ? sorted[l - 1] : 0);                                 /* :348-352 */
        int contributors = 0; int32_t best = INT32_MIN;
        for (int s = 0; s < A; s++) {
            int contrib = inv[s] >= level;                                                    /* :353 */
            contributors += contrib;
            if (contrib && eligible[s] && ranks[s] > best) best = ranks[s];                   /* :354-357 */
        }
        int winners = 0, first = -1;
        for (int s = 0; s < A; s++) {
            int w = (inv[s] >= level) && eligible[s] && ranks[s] == best;                     /* :358 */
            if (w) { winners++; if (first < 0) first = s; }
        }
        int32_t layer_pot = layer_size * contributors;                                        /* :360 */
        if (!(layer_size > 0 && winners > 0)) continue;                                       /* :362 */
        int32_t share = layer_pot / winners, rem = layer_pot % winners;                       /* :364-373 (non-negative) */
        for (int s = 0; s < A; s++)
            if ((inv[s] >= level) && eligible[s] && ranks[s] == best) payout[s] += share;     /* :374 */
        payout[first] += rem;                                                                 /* :375-376 */
    }
    for (int s = 0; s < A; s++) v->stacks[(size_t)t * P + s] += payout[s];                    /* :377-378 */
}

/* PokerGPU.py:380-453 resolve_terminated_games, one table; uses v->is_done[t] as "newly done" */
void oracle_resolve_terminated(const OraclePoker* v, int t, int newly_done) {
    const int P = v->n_players, A = v->active_players;
    if (!(v->stages[t] < 5 && newly_done)) return;                                            /* :386 */
    if (!(count_contenders(v, t) > 1)) return;                                                /* :390 */
    int32_t* b = v->board + (size_t)t * 5;
    if (v->stages[t] == 0) {                                                                  /* :394-405 */
        v->deck_positions[t] += 1; deal_cards(v, t, 3, b);
        v->deck_positions[t] += 1; deal_cards(v, t, 1, b + 3);
        v->deck_positions[t] += 1; deal_cards(v, t, 1, b + 4);
    } else if (v->stages[t] == 1) {                                                           /* :407-414 */
        v->deck_positions[t] += 1; deal_cards(v, t, 1, b + 3);
        v->deck_positions[t] += 1; deal_cards(v, t, 1, b + 4);
    } else if (v->stages[t] == 2) {                                                           /* :416-419 */
        v->deck_positions[t] += 1; deal_cards(v, t, 1, b + 4);
    }
    int32_t ranks[16]; int eligible[16];
    for (int s = 0; s < A; s++) {                                                             /* :427-450 */
        int32_t c[7];
        c[0] = v->hands[((size_t)t * P + s) * 2]; c[1] = v->hands[((size_t)t * P + s) * 2 + 1];
        for (int j = 0; j < 5; j++) c[2 + j] = b[j];
        int st = v->status[(size_t)t * P + s];
        eligible[s] = (st == ACTIVE) || (st == ALLIN);
        ranks[s] = eligible[s] ? walk(v, c, 7) : INT32_MIN;
    }
    award_side_pots(v, t, ranks, eligible);
    v->pots[t] = 0; v->stages[t] = 5;                                                         /* :452-453 */
}

/* PokerGPU.py:527-633 step, one table.  Returns the reward; done flag is v->is_done[t]. */
float oracle_step_table(const OraclePoker* v, int t, int64_t action) {
    const int P = v->n_players, A = v->active_players;
    const int prev_done = v->is_done[t];                                                      /* :530 */
    const int actor_idx = v->idx[t];                                                          /* :531 */
    const int ast = v->status[(size_t)t * P + actor_idx];
    const int has_legal_actor = (ast != FOLDED) && (ast != ALLIN) && (ast != SITOUT) && !prev_done; /* :532-537 */
    v->prev_stacks[t] = v->stacks[(size_t)t * P + actor_idx];                                 /* :538 */
    v->prev_invested[t] = v->current_round_bet[(size_t)t * P + actor_idx];                    /* :539 */

    oracle_calculate_equities(v, t);                                                          /* :542-543 */
    oracle_execute_actions(v, t, action);                                                     /* :546 */

    int truly_active = 0;                                                                     /* :547 */
    for (int s = 0; s < P; s++) truly_active += v->status[(size_t)t * P + s] == ACTIVE;
    const int all_allin_or_folded = truly_active == 0;                                        /* :548 */
    const int all_acted = v->acted[t] >= truly_active;                                        /* :549 */

    int round_over = v->is_done[t] || all_allin_or_folded;                                    /* :552-553 */
    int has_next = 0, next_seat = 0;                                                          /* :554-561 */
    for (int k = 1; k <= A; k++) {
        int seat = pymod(v->idx[t] + k, A);
        if (v->status[(size_t)t * P + seat] == ACTIVE) { has_next = 1; next_seat = seat; break; }
    }
    if (!has_next) next_seat = pymod(v->idx[t] + 1, A);   /* argmax of all-zero row -> offset 1 */
    const int unresolved = !round_over;                                                       /* :562 */
    const int closes_cur = all_acted && (v->idx[t] == v->agg[t]);                             /* :563 */
    const int closes_next = all_acted && has_next && (next_seat == v->agg[t]);                /* :564 */
    round_over |= unresolved && (!has_next || closes_cur || closes_next);                     /* :565-569 */
    v->is_round_over[t] = (uint8_t)round_over;
    if (!round_over && has_next) v->idx[t] = next_seat;                                       /* :571-573 */

    const int contenders = count_contenders(v, t);                                            /* :576 */
    const int early_term = (contenders <= 1) && round_over;                                   /* :577 */
    if (early_term) v->is_done[t] = 1;                                                        /* :578 */

    const int transition = round_over && !early_term && !v->is_done[t];                       /* :580 */
    if (transition) {
        v->last_raise_size[t] = 1;                                                            /* :581 */
        v->stages[t] += 1;                                                                    /* :584 */
        v->highest[t] = 0;                                                                    /* :585 */
        v->agg[t] = pymod(v->button[t] + 1, A);                                               /* :586 */
        v->acted[t] = 0;                                                                      /* :587 */
        for (int s = 0; s < P; s++) v->current_round_bet[(size_t)t * P + s] = 0;              /* :588 */
        set_first_active_street_actor(v, t);                                                  /* :589 */
        const int st = v->stages[t];
        if (st > 3) { v->is_done[t] = 1; v->stages[t] = 4; }                                  /* :595-598 */
        if (st == 1) { v->deck_positions[t] += 1; deal_cards(v, t, 3, v->board + (size_t)t * 5); v->equity_dirty[t] = 1; }      /* :601-604 */
        if (st == 2) { v->deck_positions[t] += 1; deal_cards(v, t, 1, v->board + (size_t)t * 5 + 3); v->equity_dirty[t] = 1; }  /* :607-610 */
        if (st == 3) { v->deck_positions[t] += 1; deal_cards(v, t, 1, v->board + (size_t)t * 5 + 4); v->equity_dirty[t] = 1; }  /* :613-616 */
    }

    const int all_done = v->is_done[t];                                                       /* :619 */
    const int newly_done = all_done && !prev_done;                                            /* :620 */
    oracle_resolve_fold_winner(v, t, newly_done);                                             /* :621 */
    oracle_resolve_terminated(v, t, newly_done);                                              /* :622 */

    if (all_done) {                                                                           /* :625-628 */
        for (int s = 0; s < P; s++) { v->current_round_bet[(size_t)t * P + s] = 0; v->total_invested[(size_t)t * P + s] = 0; }
        v->highest[t] = 0;
    }
    float r = oracle_reward(v, t, action, actor_idx);                                         /* :631 */
    if (!has_legal_actor || prev_done) r = 0.0f;                                              /* :632 */
    oracle_get_obs(v, t);                                                                     /* :633 */
    return r;
}

/* Whole batch, optionally over OpenMP threads (bench.py cpu_baseline; threads reported there). */
void oracle_step(const OraclePoker* v, const int64_t* actions, float* rewards, int n_threads) {
    const int N = v->n_games;
    #pragma omp parallel for num_threads(n_threads) schedule(static) if (n_threads > 1)
    for (int t = 0; t < N; t++) rewards[t] = oracle_step_table(v, t, actions[t]);
}

/* PokerGPU.py:188-199 post_blinds, one table */
void oracle_post_blinds(const OraclePoker* v, int t) {
    const int P = v->n_players;
    const int bb = v->bb[t];
    v->stacks[(size_t)t * P + bb] -= 1;
    v->current_round_bet[(size_t)t * P + bb] += 1;
    v->total_invested[(size_t)t * P + bb] += 1;
    v->pots[t] += 1;
    v->status[(size_t)t * P + bb] = (v->stacks[(size_t)t * P + bb] == 0) ? ALLIN : ACTIVE;
}

/*
 * PokerGPU.py:73-157 reset, one table, after the host has fixed `active_players`, filled
 * v->decks (prefixed or random) and computed the new button value.  `first` = no previous stacks.
 */
void oracle_reset_table(const OraclePoker* v, int t, int first, int starting_bbs, int max_bbs,
                        int rotation, int button_value) {
    const int P = v->n_players, A = v->active_players;
    v->last_raise_size[t] = 1;                                                                /* :81 */
    v->deck_positions[t] = 0;                                                                 /* :93 */
    for (int j = 0; j < 5; j++) v->board[(size_t)t * 5 + j] = -1;                             /* :95 */
    v->pots[t] = 0; v->stages[t] = 0;                                                         /* :96-97 */
    int32_t* st = v->stacks + (size_t)t * P;
    if (first) { for (int s = 0; s < P; s++) st[s] = starting_bbs; }                          /* :101-102 */
    else {                                                                                    /* :104-110 */
        int32_t tmp[16];
        for (int s = 0; s < P; s++) { int32_t x = st[s]; if (x == 0 || x > max_bbs) x = starting_bbs; tmp[s] = x; }
        for (int s = 0; s < P; s++) st[pymod(s + rotation, P)] = tmp[s];                      /* torch.roll */
    }
    for (int s = 0; s < P; s++) {                                                             /* :112-119 */
        int32_t* h = v->hands + ((size_t)t * P + s) * 2;
        if (s < A) { h[0] = v->decks[(size_t)t * 52 + 2 * s]; h[1] = v->decks[(size_t)t * 52 + 2 * s + 1]; }
        else { h[0] = -1; h[1] = -1; }
        v->current_round_bet[(size_t)t * P + s] = 0;
        v->total_invested[(size_t)t * P + s] = 0;
        v->status[(size_t)t * P + s] = s < A ? ACTIVE : SITOUT;
    }
    v->deck_positions[t] += 2 * A;                                                            /* :205 */
    v->button[t] = button_value;                                                              /* :121 */
    if (A == 2) { v->sb[t] = v->button[t]; v->bb[t] = pymod(v->button[t] + 1, A); }            /* :123-125 */
    else { v->sb[t] = pymod(v->button[t] + 1, A); v->bb[t] = pymod(v->button[t] + 2, A); }     /* :127-128 */
    oracle_post_blinds(v, t);                                                                 /* :129 */
    v->idx[t] = (A == 2) ? v->button[t] : pymod(v->bb[t] + 1, A);                              /* :130-133 */
    v->highest[t] = 1; v->agg[t] = v->bb[t]; v->acted[t] = 0; v->is_done[t] = 0;             /* :134-137 */
    v->raise_amounts[t] = 0; v->is_round_over[t] = 0;                                         /* :139,142 */
    for (int s = 0; s < A; s++) v->equities[(size_t)t * A + s] = 0.5f;                        /* :144 */
    v->equity_dirty[t] = 1;                                                                   /* :145 */
    v->prev_stacks[t] = 0; v->prev_invested[t] = 0;                                           /* :150-151 */
    oracle_get_obs(v, t);                                                                     /* :157 */
}

void oracle_reset(const OraclePoker* v, int first, int starting_bbs, int max_bbs, int rotation,
                  const int32_t* button_values, int n_threads) {
    const int N = v->n_games;
    #pragma omp parallel for num_threads(n_threads) schedule(static) if (n_threads > 1)
    for (int t = 0; t < N; t++)
        oracle_reset_table(v, t, first, starting_bbs, max_bbs, rotation, button_values[t]);
}

/* Standalone 5/6/7-card lookups for table checks (PokerGPU.py:437-444, :500, :521). */
void oracle_eval_hands(const int32_t* hr, int64_t hr_len, const int32_t* cards, int n_hands,
                       int n_cards, int32_t* out) {
    OraclePoker v; memset(&v, 0, sizeof v); v.hand_ranks = hr; v.hr_len = hr_len;
    for (int i = 0; i < n_hands; i++) {
        int32_t p = walk(&v, cards + (size_t)i * n_cards, n_cards);
        out[i] = (n_cards == 7) ? p : hr_at(&v, p);
    }
}

/* ---- scripted opponents: environments/Poker/utils.py:108-123 + Player.py:79-176 ----------------
 * One table; c1,c2 = obs cols 5,6 (hole cards of the seat to act), pot = obs col 9.  Random picks
 * come from Philox4x32-10(seed, table id, step counter) -- the stream the HIP policy kernel uses --
 * so CPU and GPU roll-outs follow the same trajectory.  type: 0 external (untouched), 1 random,
 * 2 heuristic_hands, 3 tight_aggressive, 4 loose_passive, 5 small_ball. */
void oracle_philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset, uint32_t out[4]);

/* ---- deck shuffle (replaces torch.rand(N,52).argsort(dim=1)+1, PokerGPU.py:86) -----------------
 * The framework's own definition (the reference's draw is not reproducible across devices): card c
 * (0..51) gets key word (c & 3) of Philox4x32-10(seed, global table id, episode*16 + (c >> 2)),
 * cut to its top key_bits bits (1..26; anything else means 26 -- torch.rand's float32 keys have 24); the deck lists
 * cards + 1 by ascending key, equal keys in card order (a stable sort -- here an insertion sort; the HIP kernel sorts
 * the words key << 6 | card with a bitonic network, which is the same order). */
void oracle_shuffle_decks(uint64_t seed, uint64_t table_id0, uint64_t episode, int key_bits, int n_tables,
                          int32_t* decks) {
    const int shift = 32 - ((key_bits > 0 && key_bits <= 26) ? key_bits : 26);
    #pragma omp parallel for schedule(static)               /* tables are independent; bench.py's CPU leg shuffles 65,536 per episode */
    for (int t = 0; t < n_tables; t++) {
        uint32_t key[52]; int32_t* d = decks + (size_t)t * 52;
        for (int s = 0; s < 13; s++) {
            uint32_t r[4];
            oracle_philox4x32(seed, table_id0 + (uint64_t)t, episode * 16 + (uint64_t)s, r);
            for (int q = 0; q < 4; q++) key[4 * s + q] = r[q] >> shift;
        }
        int n = 0;
        for (int c = 0; c < 52; c++) {                      /* insert card c after every key <= its own */
            int i = n++;
            while (i > 0 && key[d[i - 1] - 1] > key[c]) { d[i] = d[i - 1]; i--; }
            d[i] = c + 1;
        }
    }
}

static int rand_below(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); }

/* rnd[0] feeds the action's randint, rnd[1] loose_passive's rand() (Player.py:146); see oracle_policy_draw */
int oracle_scripted_action(int type, int c1, int c2, int pot, const uint32_t rnd[2]) {
    const int r1 = pymod(c1, 13), r2 = pymod(c2, 13);
    const int d = r1 > r2 ? r1 - r2 : r2 - r1;
    const int pair = r1 == r2;
    int a = 0;
    if (type == 1) a = rand_below(rnd[0], 13);                                        /* utils.py:121 */
    else if (type == 2) {                                                             /* Player.py:85-102 */
        int fold = r1 < 8 && r2 < 8;
        int raise = (pair || r1 >= 10 || r2 >= 10) && !fold;
        a = raise ? 2 + rand_below(rnd[0], 9) : 0;
    } else if (type == 3) {                                                           /* Player.py:112-124 */
        int fold = r1 < 7 && r2 < 7 && d > 5;
        int raise = (pair || (r1 >= 10 && r2 > 5) || (r2 >= 10 && r1 > 5)) && !fold;
        a = fold ? 0 : 1;
        if (raise) a = 2 + 5 + rand_below(rnd[0], 4);
    } else if (type == 4) {                                                           /* Player.py:134-149 */
        int fold = r1 <= 4 && r2 <= 4 && d > 9;
        int call = ((pair && r1 > 8) || (r1 >= 11 && r2 > 9) || (r2 >= 11 && r1 > 9)) && !fold;
        float u = (float)(rnd[1] >> 8) * (1.0f / 16777216.0f);
        int raise = (u > 0.9f) && call;
        a = call ? 1 : 0;
        if (raise) a = 2 + rand_below(rnd[0], 4);
    } else if (type == 5) {                                                           /* Player.py:159-174 */
        int fold = (r1 < 6 && r2 < 6 && pot > 30) || (r1 < 9 && r2 < 9 && pot > 80);
        int raise = (pair || (r1 >= 10 && r2 > 5) || (r2 >= 10 && r1 > 5)) && !fold;
        a = raise ? 2 + rand_below(rnd[0], 3) : 0;
    }
    return a;
}

/* the same over rows with given draws (tests: the reference's policies are recorded with their draws forced to the
 * ends of their ranges, tests/golden/scripted.npz); type 0 rows are left untouched, as build_actions leaves EXTERNAL seats */
void oracle_scripted_actions_rows(const uint8_t* types, const int32_t* c1, const int32_t* c2, const int32_t* pot, int n,
                                  uint32_t pick, uint32_t coin, int64_t* actions) {
    const uint32_t rnd[2] = {pick, coin};
    for (int i = 0; i < n; i++)
        if (types[i]) actions[i] = oracle_scripted_action(types[i], c1[i], c2[i], pot[i], rnd);
}

/* The scripted opponents' draw for (table, step) -- the definition the HIP kernels are held to: one Philox call
 * serves two consecutive steps, call = Philox4x32-10(seed, table id, step >> 1); an even step takes words (x, y) of
 * the call, an odd step (z, w).  (torch's generator stream of the reference cannot be reproduced in a kernel; only
 * the distributions are the reference's, Player.py:99,121,146 / utils.py:121.) */
void oracle_policy_draw(uint64_t seed, uint64_t table_id, uint64_t step_counter, uint32_t out[2]) {
    uint32_t call[4];
    oracle_philox4x32(seed, table_id, step_counter >> 1, call);
    out[0] = (step_counter & 1) ? call[2] : call[0];
    out[1] = (step_counter & 1) ? call[3] : call[1];
}

/* build_actions over the batch from the observation buffer (utils.py:108-123) */
void oracle_policy(const OraclePoker* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                   uint64_t table_id0, int64_t* actions, int n_threads) {
    const int N = v->n_games;
    #pragma omp parallel for num_threads(n_threads) schedule(static) if (n_threads > 1)
    for (int t = 0; t < N; t++) {
        const int type = agent_types[v->idx[t] & 15];
        if (!type) continue;
        const float* o = v->obs + (size_t)t * v->obs_size;
        uint32_t rnd[2];
        oracle_policy_draw(seed, table_id0 + (uint64_t)t, step_counter, rnd);
        actions[t] = oracle_scripted_action(type, (int)o[5], (int)o[6], (int)o[9], rnd);
    }
}

/* policy + step, the unit bench.py times as one env-step pass on the CPU */
void oracle_policy_step(const OraclePoker* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                        uint64_t table_id0, int64_t* actions, float* rewards, int n_threads) {
    oracle_policy(v, agent_types, seed, step_counter, table_id0, actions, n_threads);
    oracle_step(v, actions, rewards, n_threads);
}
