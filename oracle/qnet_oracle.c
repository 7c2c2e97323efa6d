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
#include <stdlib.h>
#include <omp.h>
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

/* ---- train_step (Player.py:255-294), scalar ------------------------------------------------------------------
 * Row filter: row_mask[r] (NULL = all) and states[r][12] in {0, 2} (:261).  Forward in train mode: Dropout(.1) after
 * the 2nd and 3rd GELU (:194,:197) with the framework's own draw definition (torch's generator is not reproducible
 * across devices): hidden unit u (0..127 = layer 2, 128..191 = layer 3) of table g drops when the 16-bit uniform
 *   word (u % 8) / 2, half u % 2 of Philox4x32-10(seed ^ 0xD50F0D50F0, g, 32 * step + u / 8)
 * is below (uint32)(p * 65536); kept units are scaled by 1 / (1 - p).  Target r + gamma * max Q_target(s') * !done
 * (:275-277).  Outputs: grad = d/dtheta of sum_rows (q[a] - target)^2 (NOT yet divided by the row count), flat layout
 * w1,b1,...,w5,b5; returns the row count, *sum_sq the summed squared TD error. */
static float gelu_grad_exact(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

static int drop_keep(uint64_t seed, uint64_t g, uint64_t step, int u, uint32_t thr) {
    uint32_t w[4];
    oracle_philox4x32(seed ^ 0xD50F0D50F0ull, g, step * 32 + (uint64_t)(u / 8), w);
    const uint32_t word = w[(u % 8) / 2];
    const uint32_t u16 = (u % 2) ? (word >> 16) : (word & 0xFFFFu);
    return u16 >= thr;
}

int oracle_qnet_train_grads(int state_dim, int n_actions, const float* const* w, const float* const* b,
                            const float* const* tw, const float* const* tb, const float* states, long stride,
                            const int64_t* actions, const float* rewards, const float* next_states, long next_stride,
                            const uint8_t* dones, const uint8_t* row_mask, int n_rows, float gamma, float drop_p,
                            uint64_t seed, uint64_t step, uint64_t table_id0, float* grad, float* sum_sq) {
    const int dims[6] = {state_dim, 128, 128, 64, 32, n_actions};
    size_t off_w[5], off_b[5], o = 0;
    for (int l = 0; l < 5; l++) { off_w[l] = o; o += (size_t)dims[l + 1] * dims[l]; off_b[l] = o; o += dims[l + 1]; }
    for (size_t i = 0; i < o; i++) grad[i] = 0.0f;
    const uint32_t thr = (uint32_t)(drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - drop_p);
    int count = 0; double sq = 0.0;
    /* rows are independent: each thread sums the gradients of its (static, contiguous) share of the rows into a buffer of its
     * own, the buffers are added in thread order afterwards -- deterministic for a given thread count.  (Serial, the 70,000-row
     * case of the GPU suite took minutes on a busy box: an OpenMP region was opened per row by the target forward below.) */
    const int nt = omp_get_max_threads();
    float* gbuf = (float*)calloc((size_t)nt * o, sizeof(float));
    #pragma omp parallel reduction(+:count, sq)
    {
    float* grad_t = gbuf + (size_t)omp_get_thread_num() * o;
    #pragma omp for schedule(static)
    for (int r = 0; r < n_rows; r++) {
        const float* x = states + (size_t)r * stride;
        if (row_mask && !row_mask[r]) continue;
        if (!(x[12] == 0.0f || x[12] == 2.0f)) continue;
        const uint64_t g = table_id0 + (uint64_t)r;
        float a[6][128], gd[6][128];            /* a[l] = output of layer l (a[0] = input), gd[l] = local derivative */
        for (int k = 0; k < state_dim; k++) a[0][k] = x[k];
        for (int l = 0; l < 5; l++) {
            for (int u = 0; u < dims[l + 1]; u++) {
                float z = 0.0f;
                for (int k = 0; k < dims[l]; k++) z += w[l][(size_t)u * dims[l] + k] * a[l][k];
                z += b[l][u];
                if (l == 4) { a[5][u] = z; continue; }
                float m = 1.0f;
                if (l == 1) m = drop_keep(seed, g, step, u, thr) ? scale : 0.0f;
                if (l == 2) m = drop_keep(seed, g, step, 128 + u, thr) ? scale : 0.0f;
                a[l + 1][u] = gelu_exact(z) * m;
                gd[l + 1][u] = gelu_grad_exact(z) * m;
            }
        }
        float qn[32], best = -INFINITY;
        oracle_qnet_forward(state_dim, n_actions, tw, tb, next_states + (size_t)r * next_stride, next_stride, 1, qn);
        for (int u = 0; u < n_actions; u++) if (qn[u] > best) best = qn[u];
        const float target = rewards[r] + gamma * best * (dones[r] ? 0.0f : 1.0f);
        const int act = (int)actions[r];
        const float td = a[5][act] - target;
        count++; sq += (double)td * td;
        float d[128], dprev[128];
        for (int u = 0; u < n_actions; u++) d[u] = (u == act) ? 2.0f * td : 0.0f;
        for (int l = 4; l >= 0; l--) {
            for (int u = 0; u < dims[l + 1]; u++) {
                grad_t[off_b[l] + u] += d[u];
                for (int k = 0; k < dims[l]; k++) grad_t[off_w[l] + (size_t)u * dims[l] + k] += d[u] * a[l][k];
            }
            if (l == 0) break;
            for (int k = 0; k < dims[l]; k++) {
                float s = 0.0f;
                for (int u = 0; u < dims[l + 1]; u++) s += w[l][(size_t)u * dims[l] + k] * d[u];
                dprev[k] = s * gd[l][k];
            }
            for (int k = 0; k < dims[l]; k++) d[k] = dprev[k];
        }
    }
    }
    for (int t = 0; t < nt; t++)
        for (size_t i = 0; i < o; i++) grad[i] += gbuf[(size_t)t * o + i];
    free(gbuf);
    *sum_sq = (float)sq;
    return count;
}

/* gradient mean, clip_grad_norm_(max_norm) (:280), torch.optim.AdamW step (:281), target sync (:289-290); t = the new
 * step number (1-based).  count == 0: nothing changes (:262).  Returns the gradient norm before clipping. */
float oracle_qnet_adamw(int n_params, float* params, float* target, const float* grad, float* m, float* v, int count,
                        long t, float lr, float wd, float beta1, float beta2, float eps, float max_norm, int update_freq) {
    if (count <= 0) return 0.0f;
    const float inv = 1.0f / (float)count;
    double ss = 0.0;
    for (int i = 0; i < n_params; i++) { const float g = grad[i] * inv; ss += (double)g * g; }
    const float norm = (float)sqrt(ss);
    float coef = max_norm / (norm + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
    coef *= inv;
    const float bc1 = 1.0f - powf(beta1, (float)t), bc2 = 1.0f - powf(beta2, (float)t);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    for (int i = 0; i < n_params; i++) {
        const float g = grad[i] * coef;
        float p = params[i] * (1.0f - lr * wd);
        m[i] = beta1 * m[i] + (1.0f - beta1) * g;
        v[i] = beta2 * v[i] + (1.0f - beta2) * g * g;
        p -= step_size * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
        params[i] = p;
        if (update_freq > 0 && t % update_freq == 0) target[i] = p;
    }
    return norm;
}
