/*
 * oracle/envs_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Scalar CPU restatements of the reference's other three environment steps and of the two
 * tabular-Q helpers, one env at a time, each citing the reference lines it follows:
 *   Blackjack   /root/reference/environments/blackjack/blackjack.py
 *   2048 (TFE)  /root/reference/environments/2048/TFE.py          (batched form: one board per row)
 *   Particle2D  /root/reference/environments/Particle2D/Particle2D.py
 *   Q helpers   /root/reference/utils/numba.py
 * Randomness the reference draws from a library RNG (numba `random`, torch.rand) is injected:
 * decks are inputs; 2048 tile spawns come from Philox4x32-10(seed, board id, step), the same
 * counter-based stream the HIP kernel uses, so CPU and GPU agree bit for bit.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void oracle_philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)subseq, (uint32_t)(subseq >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* ------------------------------------------------------------------ Blackjack */
typedef struct {
    int32_t batch_size;
    const int32_t* decks;            /* [B,52] values 0..51 (blackjack.py:24-29) */
    int32_t *deck_positions, *players_cards /*[B,20]*/, *players_card_idx, *player_card_sums;
    int32_t *dealer_cards /*[B,20]*/, *dealer_card_idx, *dealer_upcard, *dealer_card_sums;
    uint8_t *terminated, *has_ace, *dealer_has_ace;
    int32_t *rewards, *obs /*[B,3]*/;
} OracleBlackjack;

static inline int32_t bj_rank(int32_t card) { int32_t r = card % 13 + 1; return r > 10 ? 10 : r; }

static void bj_obs(const OracleBlackjack* v, int g) {          /* blackjack.py:103-108 */
    v->obs[g * 3 + 0] = v->player_card_sums[g];
    v->obs[g * 3 + 1] = v->has_ace[g];
    v->obs[g * 3 + 2] = v->dealer_upcard[g];
}

/* blackjack.py:23-101 reset + deal_starting_cards (decks are an input) */
void oracle_blackjack_reset(const OracleBlackjack* v) {
    for (int g = 0; g < v->batch_size; g++) {
        const int32_t* d = v->decks + (size_t)g * 52;
        memset(v->players_cards + (size_t)g * 20, 0, 80);
        memset(v->dealer_cards + (size_t)g * 20, 0, 80);
        v->terminated[g] = 0; v->rewards[g] = 0;
        int32_t r1 = bj_rank(d[0]); int a1 = r1 == 1; if (a1) r1 = 11;           /* :53-59 */
        int32_t d1 = bj_rank(d[1]); int da1 = d1 == 1; if (da1) d1 = 11;         /* :62-69 */
        int32_t r2 = bj_rank(d[2]); int a2 = r2 == 1; if (a2) r2 = 11;           /* :72-78 */
        int32_t d2 = bj_rank(d[3]); int dfirst = !da1 && d2 == 1; if (d2 == 1) d2 = 11;  /* :81-87 */
        v->players_cards[(size_t)g * 20] = r1; v->players_cards[(size_t)g * 20 + 1] = r2;
        v->dealer_cards[(size_t)g * 20] = d1; v->dealer_cards[(size_t)g * 20 + 1] = d2;
        v->players_card_idx[g] = 2; v->dealer_card_idx[g] = 2; v->deck_positions[g] = 4;
        v->dealer_upcard[g] = d1;
        int has = a1 || a2, dhas = da1 || dfirst;
        int32_t ps = r1 + r2, ds = d1 + d2;                                     /* :89-90 */
        if (ps > 21 && has) { ps -= 10; has = 0; }                              /* :93-95 */
        if (ds > 21 && dhas) { ds -= 10; dhas = 0; }                            /* :99-101 */
        v->player_card_sums[g] = ps; v->dealer_card_sums[g] = ds;
        v->has_ace[g] = (uint8_t)has; v->dealer_has_ace[g] = (uint8_t)dhas;
        bj_obs(v, g);
    }
}

/* blackjack.py:113-186 step */
void oracle_blackjack_step(const OracleBlackjack* v, const int64_t* actions) {
    for (int g = 0; g < v->batch_size; g++) {
        const int32_t* d = v->decks + (size_t)g * 52;
        const int hit = actions[g] == 0 && !v->terminated[g];                   /* :117 */
        const int stand = actions[g] == 1 && !v->terminated[g];                 /* :138 */
        if (hit) {                                                              /* :118-135 */
            int32_t rank = bj_rank(d[v->deck_positions[g]]);
            int already = v->has_ace[g], ace = rank == 1;
            if (ace && !already) rank = 11;
            v->players_cards[(size_t)g * 20 + v->players_card_idx[g]] = rank;
            v->has_ace[g] |= (uint8_t)(ace && !already);
            v->player_card_sums[g] += rank;
            v->deck_positions[g] += 1; v->players_card_idx[g] += 1;
            if (v->player_card_sums[g] > 21 && v->has_ace[g]) { v->player_card_sums[g] -= 10; v->has_ace[g] = 0; }
        }
        if (stand) {                                                            /* :139-160 */
            int active = v->dealer_card_sums[g] < 17;
            while (active) {
                int32_t rank = bj_rank(d[v->deck_positions[g]]);
                int already = v->dealer_has_ace[g], ace = rank == 1;
                if (ace && !already) rank = 11;
                v->dealer_cards[(size_t)g * 20 + v->dealer_card_idx[g]] = rank;
                v->dealer_card_idx[g] += 1;
                v->dealer_has_ace[g] |= (uint8_t)(ace && !already);
                v->dealer_card_sums[g] += rank;
                if (v->dealer_card_sums[g] > 21 && v->dealer_has_ace[g]) { v->dealer_card_sums[g] -= 10; v->dealer_has_ace[g] = 0; }
                v->deck_positions[g] += 1;
                active = v->dealer_card_sums[g] < 17 && v->dealer_card_sums[g] <= 21;
            }
        }
        v->rewards[g] = 0;                                                      /* :183 */
        if (hit && v->player_card_sums[g] > 21) { v->rewards[g] = -1; v->terminated[g] = 1; }   /* :166-168 */
        if (stand) {                                                            /* :171-177 */
            int win = v->dealer_card_sums[g] > 21 || v->player_card_sums[g] >= v->dealer_card_sums[g];
            v->rewards[g] = win ? 1 : -1;
            v->terminated[g] = 1;
        }
        bj_obs(v, g);
    }
}

/* ------------------------------------------------------------------ 2048 */
#define TFE_MAX 8

/* TFE.py:38-44 numba_rotate_with_buffer (square boards): out[m-1-j, i] = in[i, j] */
static void tfe_rotate(const int32_t* in, int32_t* out, int n) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) out[(n - 1 - j) * n + i] = in[i * n + j];
}

/* TFE.py:75-108 squash_row_optimized */
static int32_t tfe_squash_row(const int32_t* row, int32_t* res, int n) {
    int32_t score = 0; int w = 0, last_merged = 0;
    for (int i = 0; i < n; i++) res[i] = 0;
    for (int i = 0; i < n; i++) {
        int32_t val = row[i];
        if (val != 0) {
            if (res[w] == 0) res[w] = val;
            else if (res[w] == val && !last_merged) { res[w] = val * 2; score += val * 2; last_merged = 1; }
            else { w++; res[w] = val; last_merged = 0; }
        }
    }
    return score;
}

/* TFE.py:17-34 add_tile_numba with injected uniforms: cell = floor(u32 * n_empty / 2^32),
 * value 4 iff the 24-bit uniform float of the second word is > 0.9. */
static void tfe_add_tile(int32_t* b, int n, uint32_t r_cell, uint32_t r_val) {
    int empties[TFE_MAX * TFE_MAX]; int ne = 0;
    for (int i = 0; i < n * n; i++) if (b[i] == 0) empties[ne++] = i;
    if (!ne) return;
    int k = (int)(((uint64_t)r_cell * (uint64_t)ne) >> 32);
    float u = (float)(r_val >> 8) * (1.0f / 16777216.0f);
    b[empties[k]] = u > 0.9f ? 4 : 2;
}

/* TFE.py:48-67 is_game_over_numba */
static int tfe_game_over(const int32_t* b, int n) {
    for (int i = 0; i < n * n; i++) if (b[i] == 0) return 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < n - 1; j++) if (b[i * n + j] == b[i * n + j + 1]) return 0;
    for (int i = 0; i < n - 1; i++) for (int j = 0; j < n; j++) if (b[i * n + j] == b[(i + 1) * n + j]) return 0;
    return 1;
}

/* TFE.py:143-149 reset: two spawns.  Philox offset 2*step_counter, step_counter = 0 here. */
void oracle_tfe_reset(int32_t* boards, int64_t* total_score, int n_boards, int n, uint64_t seed,
                      uint64_t board_id0) {
    for (int g = 0; g < n_boards; g++) {
        int32_t* b = boards + (size_t)g * n * n;
        memset(b, 0, sizeof(int32_t) * (size_t)n * n);
        total_score[g] = 0;
        uint32_t r[4]; oracle_philox4x32(seed, board_id0 + (uint64_t)g, 0, r);
        tfe_add_tile(b, n, r[0], r[1]);
        tfe_add_tile(b, n, r[2], r[3]);
    }
}

/* TFE.py:152-189 step; `step_counter` (>=1) selects the Philox offset for this step's spawn */
void oracle_tfe_step(int32_t* boards, int64_t* total_score, const int64_t* actions, int32_t* rewards,
                     uint8_t* dones, int n_boards, int n, uint64_t seed, uint64_t board_id0,
                     uint64_t step_counter) {
    for (int g = 0; g < n_boards; g++) {
        int32_t* b = boards + (size_t)g * n * n;
        int32_t x[TFE_MAX * TFE_MAX], y[TFE_MAX * TFE_MAX];
        int k = (int)(actions[g] & 3);                                          /* :154 */
        memcpy(x, b, sizeof(int32_t) * (size_t)n * n);
        for (int i = 0; i < k; i++) { tfe_rotate(x, y, n); memcpy(x, y, sizeof(int32_t) * (size_t)n * n); }   /* :158-163 */
        int32_t score = 0;
        for (int r = 0; r < n; r++) score += tfe_squash_row(x + r * n, y + r * n, n);    /* :166-167 */
        total_score[g] += score;                                                /* :168 */
        memcpy(x, y, sizeof(int32_t) * (size_t)n * n);
        for (int i = 0; i < (4 - k) % 4; i++) { tfe_rotate(x, y, n); memcpy(x, y, sizeof(int32_t) * (size_t)n * n); } /* :171-178 */
        memcpy(b, x, sizeof(int32_t) * (size_t)n * n);                          /* :181 */
        uint32_t r[4]; oracle_philox4x32(seed, board_id0 + (uint64_t)g, step_counter, r);
        tfe_add_tile(b, n, r[0], r[1]);                                         /* :182 */
        int32_t rew = 0;                                                        /* :185-187 */
        if (score > 0) { int bl = 0; uint32_t s = (uint32_t)score; while (s) { bl++; s >>= 1; } rew = bl - 1; }
        rewards[g] = rew;
        dones[g] = (uint8_t)tfe_game_over(b, n);                                /* :189 */
    }
}

/* ------------------------------------------------------------------ Particle2D */
/* Particle2D.py:22-30 step.  state [B,4] = x,y,vx,vy (fp32); action [B,2]; obs_out = state.clone() */
void oracle_particle2d_step(float* state, const float* action, int32_t* steps, float* obs_out,
                            float* rewards, uint8_t* terminated, int n, float dt, int max_steps) {
    for (int i = 0; i < n; i++) {
        float ax = action[2 * i], ay = action[2 * i + 1];
        ax = ax < -1.0f ? -1.0f : (ax > 1.0f ? 1.0f : ax);                      /* :23 */
        ay = ay < -1.0f ? -1.0f : (ay > 1.0f ? 1.0f : ay);
        volatile float dvx = ax * dt, dvy = ay * dt;
        float vx = state[4 * i + 2] + dvx, vy = state[4 * i + 3] + dvy;         /* :24 */
        volatile float dx = vx * dt, dy = vy * dt;
        float x = state[4 * i] + dx, y = state[4 * i + 1] + dy;                 /* :25 */
        volatile float xx = x * x, yy = y * y;
        float dist = sqrtf(xx + yy);                                            /* :26 */
        volatile float a2x = ax * ax, a2y = ay * ay;
        volatile float pen = 0.001f * (a2x + a2y);
        rewards[i] = -dist - pen;                                               /* :27 */
        steps[i] += 1;                                                          /* :28 */
        terminated[i] = (uint8_t)((dist < 0.1f) || (steps[i] >= max_steps));    /* :29 */
        state[4 * i] = x; state[4 * i + 1] = y; state[4 * i + 2] = vx; state[4 * i + 3] = vy;
        obs_out[4 * i] = x; obs_out[4 * i + 1] = y; obs_out[4 * i + 2] = vx; obs_out[4 * i + 3] = vy;
    }
}

/* ------------------------------------------------------------------ tabular-Q helpers */
/* utils/numba.py:5-21 select_action_epsilon_greedy_numba with injected uniforms p, r_int */
int32_t oracle_select_action_epsilon_greedy(const double* q, int n, double epsilon, double p, uint32_t r_int) {
    if (p < epsilon) return (int32_t)(((uint64_t)r_int * (uint64_t)n) >> 32);
    int idx = 0; double mx = q[0];
    for (int i = 1; i < n; i++) if (q[i] > mx) { mx = q[i]; idx = i; }
    return idx;
}

/* utils/numba.py:25-39 update_q_entry */
void oracle_update_q_entry(double* cur, int32_t action, const double* nxt, int n, double alpha,
                           double reward, double gamma, int is_terminal) {
    double mx = nxt[0];
    for (int i = 1; i < n; i++) if (nxt[i] > mx) mx = nxt[i];
    double target = is_terminal ? reward : reward + gamma * mx;
    double old = cur[action];
    cur[action] = old + alpha * (target - old);
}
