/*
 * qnet_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py): scalar restatement of the learner's action
 * selection, environments/Poker/Player.py:178-253, against which the HIP kernel (pulselib_amd/csrc/qnet.hip)
 * is checked.  Pinned by tests/golden/qnetwork.npz (Q values and greedy actions produced by the reference's own
 * PokerQNetwork on CPU torch, tests/golden/make_golden.py: make_qnetwork).
 *
 *   network (:189-201), eval mode (dropout = identity):
 *     Linear(state_dim,128) GELU Linear(128,128) GELU Linear(128,64) GELU Linear(64,32) GELU Linear(32,n_actions)
 *   get_actions (:242-253): greedy = argmax (first maximal index), replaced by a uniform action with
 *     probability epsilon.  The draws are the framework's own definition (torch's generator is not reproducible
 *     across devices): words x, y of Philox4x32-10(seed, global table id, step), explore = unit(x) < epsilon,
 *     action = floor(y * n_actions / 2^32).
 * Sums run in plain k order with one rounding per multiply and per add (fp32); torch and the MFMA kernel order
 * the sums differently, so comparisons carry the tolerance written in the tests.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

void oracle_philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset, uint32_t out[4]);

static float gelu_exact(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }   /* nn.GELU() */

static void linear(const float* w, const float* b, int n_in, int n_out, const float* x, float* y, int act) {
    for (int o = 0; o < n_out; o++) {
        float acc = 0.0f;
        for (int k = 0; k < n_in; k++) acc += w[(size_t)o * n_in + k] * x[k];
        acc += b[o];
        y[o] = act ? gelu_exact(acc) : acc;
    }
}

/* weights: w[i] / b[i] for the five Linear layers, torch layout w[out][in] */
void oracle_qnet_forward(int state_dim, int n_actions, const float* const* w, const float* const* b, const float* states,
                         long row_stride, int n_rows, float* q_out) {
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < n_rows; r++) {
        float h1[128], h2[128], h3[64], h4[32];
        linear(w[0], b[0], state_dim, 128, states + (size_t)r * row_stride, h1, 1);
        linear(w[1], b[1], 128, 128, h1, h2, 1);
        linear(w[2], b[2], 128, 64, h2, h3, 1);
        linear(w[3], b[3], 64, 32, h3, h4, 1);
        linear(w[4], b[4], 32, n_actions, h4, q_out + (size_t)r * n_actions, 0);
    }
}

/* actions[r] for the rows with seat_idx[r] == q_seat (seat_idx NULL: all rows); other rows untouched */
void oracle_qnet_act(int n_actions, const float* q, int n_rows, const int32_t* seat_idx, int q_seat, float epsilon,
                     uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions) {
    for (int r = 0; r < n_rows; r++) {
        if (seat_idx && seat_idx[r] != q_seat) continue;
        const float* qr = q + (size_t)r * n_actions;
        int arg = 0;
        for (int a = 1; a < n_actions; a++) if (qr[a] > qr[arg]) arg = a;                    /* :248 first maximum */
        uint32_t rnd[4];
        oracle_philox4x32(seed, table_id0 + (uint64_t)r, step, rnd);
        const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
        const int explore = u < epsilon;                                                   /* :247 */
        actions[r] = explore ? (int64_t)(((uint64_t)rnd[1] * (uint64_t)n_actions) >> 32) : (int64_t)arg;   /* :249-250 */
    }
}
