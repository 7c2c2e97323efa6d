// qnet.hip -- the learner's action selection on the matrix cores (learner in the loop).
//
// The reference's PokerQNetwork (environments/Poker/Player.py:178-253) is a 5-layer perceptron
//   state_dim -> 128 -> GELU -> 128 -> GELU -> 64 -> GELU -> 32 -> GELU -> n_actions      (:189-201)
// and `get_actions` (:242-253) runs it in eval mode (dropout off) on the states whose seat to act is the
// learner's, takes the argmax and replaces it by a uniform action with probability epsilon.  In the reference
// that is a boolean-mask gather, five GEMMs with four activation kernels, rand, randint, where and a masked
// scatter; here it is ONE kernel on the env's stream:
//   * a wavefront owns 64 consecutive tables, ballots "the learner acts here", and pushes the selected rows
//     32 at a time through the whole network without leaving its registers;
//   * every layer is computed transposed, Y^T[out, row] = W[out, in] . X^T[in, row], on
//     v_mfma_f32_32x32x2_f32 (exact fp32 fma chains, the reference's dtype): the A operand is then W in its
//     torch.nn.Linear layout ([out][in] row-major, no packing pass -- a float4 load feeds four MFMAs), and the
//     32x32 accumulator tile (row index of Y^T in the 16 registers and the lane half, table on lane & 31) is
//     already the B operand of the next layer, whose k order is simply permuted to the accumulator's row order
//     k = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);  bias + exact-erf GELU run on the accumulator registers;
//   * argmax (first maximal index), the epsilon draw and the uniform action use the table's Philox words of
//     this step -- the words the scripted-opponent kernel would use had another seat been to act -- so the
//     result does not depend on launch geometry or on how tables are sharded over GPUs.
// 512 MFMAs (64 cycles each) per 32 rows; the weights (127 KB) stay in L2 and are read once per tile.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pulse_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ int rand_below(uint32_t r, int n) { return (int)__umulhi(r, (uint32_t)n); }
__device__ __forceinline__ float rand_unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// torch.nn.GELU() (approximate='none'): x * 0.5 * (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// accumulator register r of lane-half h holds row rho(r) + 4h of the 32x32 tile
__device__ __forceinline__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// One 32-out tile of a hidden layer: acc[out, table] = sum_k W[out0 + out][k] * in[k][table], K = 32 * KT,
// `in` = the previous layer's KT accumulator tiles.  Lane (c = lane & 31, h = lane >> 5) reads row out0 + c of W.
template <int KT>
__device__ __forceinline__ f32x16 dense_tile(const float* __restrict__ w, int K, int out_row, int h, const f32x16* in) {
    f32x16 acc = zero16();
    const float* wr = w + (size_t)out_row * K + 4 * h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(wr + 32 * kt + 8 * q);     // k = 32kt + 8q + 4h + j
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[kt][4 * q + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[kt][4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, in[kt][4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, in[kt][4 * q + 3], acc, 0, 0, 0);
        }
    }
    return acc;
}

// acc[r] = act(acc[r] + bias[out0 + rho(r) + 4h]) for the rows below n_out
template <bool GELU>
__device__ __forceinline__ void bias_act(f32x16& acc, const float* __restrict__ bias, int out0, int n_out, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = out0 + rho(r) + 4 * h;
        const float b = o < n_out ? bias[o] : 0.0f;
        const float y = acc[r] + b;
        acc[r] = GELU ? gelu(y) : y;
    }
}

struct QNetArgs {
    PulseQNet net;
    const float* states; long long row_stride; int n_rows;
    const int32_t* seat_idx; int q_seat;             // seat_idx == nullptr: every row is selected
    float epsilon; uint64_t seed, step, table_id0;
    int64_t* actions;                                // nullptr: no action selection (plain forward)
    float* q_out;                                    // nullptr or fp32[n_rows, n_actions]
};

// The network in eval mode on up to 32 rows: lane (c, h) carries the row at `xr` (`live` false = padding column,
// computed on zeros) as column c; returns the Q tile (row o = rho(r) + 4h of register r, o < n_actions valid).
template <bool VEC>
__device__ __forceinline__ f32x16 forward_eval(const PulseQNet& n, const float* __restrict__ xr, bool live, int lane) {
    const int c = lane & 31, h = lane >> 5, K1 = n.state_dim;

    // layer 1: state_dim -> 128, inputs straight from the observation rows in the same k order as W1's float4s
    f32x16 h1[4] = {zero16(), zero16(), zero16(), zero16()};
    for (int q = 0; q < (K1 + 7) / 8; ++q) {
        const int k0 = 8 * q + 4 * h;
        float xb[4];
        if (VEC) {
            const float4 x4 = *reinterpret_cast<const float4*>(xr + k0);
            xb[0] = x4.x; xb[1] = x4.y; xb[2] = x4.z; xb[3] = x4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[j] = k0 + j < K1 ? xr[k0 + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[j] = live ? xb[j] : 0.0f;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const float* wr = n.w1 + (size_t)(32 * ot + c) * K1 + k0;
            float wa[4];
            if (VEC) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr);
                wa[0] = w4.x; wa[1] = w4.y; wa[2] = w4.z; wa[3] = w4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wa[j] = k0 + j < K1 ? wr[j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) h1[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[j], h1[ot], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) bias_act<true>(h1[ot], n.b1, 32 * ot, 128, h);

    f32x16 h2[4];                                                          // 128 -> 128
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) { h2[ot] = dense_tile<4>(n.w2, 128, 32 * ot + c, h, h1); bias_act<true>(h2[ot], n.b2, 32 * ot, 128, h); }
    f32x16 h3[2];                                                          // 128 -> 64
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) { h3[ot] = dense_tile<4>(n.w3, 128, 32 * ot + c, h, h2); bias_act<true>(h3[ot], n.b3, 32 * ot, 64, h); }
    f32x16 h4[1];                                                          // 64 -> 32
    h4[0] = dense_tile<2>(n.w4, 64, c, h, h3); bias_act<true>(h4[0], n.b4, 0, 32, h);
    const int A = n.n_actions;                                             // 32 -> n_actions (<= 32): rows past A repeat row A-1, unused
    f32x16 qv = dense_tile<1>(n.w5, 32, min(c, A - 1), h, h4); bias_act<false>(qv, n.b5, 0, A, h);
    return qv;
}

// Action selection / Q output for up to 32 rows: lane (c, h) carries row `row` (< 0 = padding column).
template <bool VEC>
__device__ __forceinline__ void qnet_tile(const QNetArgs& a, int row, int lane) {
    const int h = lane >> 5, A = a.net.n_actions;
    const bool live = row >= 0;
    const f32x16 qv = forward_eval<VEC>(a.net, a.states + (size_t)max(row, 0) * a.row_stride, live, lane);

    if (a.q_out && live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int o = rho(r) + 4 * h; if (o < A) a.q_out[(size_t)row * A + o] = qv[r]; }
    }
    if (a.actions) {
        // first maximal index (torch.argmax): this half's rows, then the other half's through the lane pair
        float best = -INFINITY; int arg = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = rho(r) + 4 * h;
            if (o < A && (qv[r] > best || (qv[r] == best && o < arg) || arg == 0x7fffffff)) { best = qv[r]; arg = o; }
        }
        const float ob = __shfl_xor(best, 32); const int oa = __shfl_xor(arg, 32);
        if (oa != 0x7fffffff && (arg == 0x7fffffff || ob > best || (ob == best && oa < arg))) { best = ob; arg = oa; }
        if (live && h == 0) {
            const U4 rnd = philox4x32(a.seed, a.table_id0 + (uint64_t)row, a.step);
            const bool explore = rand_unit(rnd.x) < a.epsilon;                                   // Player.py:247
            a.actions[row] = explore ? (int64_t)rand_below(rnd.y, A) : (int64_t)arg;             // :248-250
        }
    }
}

// SELECT: 64 candidate rows per wavefront, those with seat_idx == q_seat are compacted and run 32 at a time.
// Dense: 32 consecutive rows per wavefront.
template <bool SELECT, bool VEC>
__global__ __launch_bounds__(64) void qnet_kernel(const QNetArgs a) {
    const int lane = threadIdx.x;
    if (!SELECT) {
        const int row = blockIdx.x * 32 + (lane & 31);
        qnet_tile<VEC>(a, row < a.n_rows ? row : -1, lane);
        return;
    }
    __shared__ int list[64];
    const int row = blockIdx.x * 64 + lane;
    const bool sel = row < a.n_rows && a.seat_idx[row] == a.q_seat;
    const unsigned long long m = __ballot(sel);
    const int count = __popcll(m);
    if (count == 0) return;
    if (sel) list[__popcll(m & ((1ull << lane) - 1ull))] = row;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int t0 = 0; t0 < count; t0 += 32) {
        const int i = t0 + (lane & 31);
        qnet_tile<VEC>(a, i < count ? list[i] : -1, lane);
    }
}

// ================================================================ training step (Player.py:255-294)
// One launch does, for the rows that pass the reference's filters, what train_step does between its masks and
// `loss.backward()`: forward in train mode (dropout after the 2nd and 3rd GELU), TD target from the target
// network, d(loss)/d(parameters) -- accumulated UNNORMALISED (the 1 / #valid-rows of MSELoss, the norm clipping
// and AdamW follow in qnet_adamw_kernel, which knows the global row count).
//
// A wavefront owns 64 candidate rows, compacts the valid ones and takes them 32 at a time (one workgroup = one
// wavefront: the tile's activations live in ~118 KB of LDS, and a training step has a few hundred tiles for 256 CUs):
//   forward   as in forward_eval, transposed orientation; every hidden layer leaves its output a_l and its local
//             derivative g_l = gelu'(z_l) * dropout-scale in LDS as [unit][row];
//   delta_5   = 2 (q[action] - target) on the action's row of the output tile;
//   per layer dW_l = delta_l . a_{l-1}^T: both operands come back out of LDS transposed ([unit][row] read with the
//             row index as the MFMA k), 16 MFMAs per 32x32 block of dW_l, then one f32 atomic per element into the
//             flat gradient buffer; db_l falls out of the same reads;
//             delta_{l-1} = (W_l^T . delta_l) * g_{l-1}: A operand = W_l read down its columns (coalesced over
//             the lanes), B operand = the delta_l accumulator tiles, as in the forward pass.
// fp32 atomics make the summation order of the gradient vary from run to run (rounding-level differences).
constexpr int kLd = 33;                      // LDS row pitch of the [unit][row] tiles (32 rows + 1 pad)

struct TrainArgs {
    PulseQNet net, tgt;
    float* grad; float* stats;                // flat gradient (layout: w1,b1,...,w5,b5), stats[0]=#valid rows, [1]=sum td^2
    const float* states; long long stride;
    const int64_t* actions; const float* rewards;
    const float* next_states; long long next_stride;
    const uint8_t* dones;
    const int* list;                          // [0] = number of rows to train on, [1..] = their ids (qnet_compact_kernel)
    uint64_t seed, step, table_id0;
    float gamma, drop_p;
};

__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// store an accumulator tile as [unit0 + row-of-tile][column]
__device__ __forceinline__ void store_t(float* __restrict__ S, int unit0, const f32x16& v, int c, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(unit0 + rho(r) + 4 * h) * kLd + c] = v[r];
}

// keep-mask bits of the 16 accumulator rows of tile `tile` (units 32*tile + rho(r) + 4h) for table `gid`:
// unit u drops when the 16-bit uniform (call u / 8, word (u % 8) / 2, half u % 2) is below drop_p * 65536.
__device__ __forceinline__ uint32_t dropout_keep_bits(uint64_t seed, uint64_t gid, uint64_t step, int tile, int h, uint32_t thr) {
    uint32_t bits = 0;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {                       // units 32*tile + 8*blk + 4h + {0,1,2,3} = registers 4*blk + j
        const U4 w = philox4x32(seed ^ 0xD50F0D50F0ull, gid, step * 32 + (uint64_t)(4 * tile + blk));
        const uint32_t lo = h ? w.z : w.x, hi = h ? w.w : w.y;
        bits |= (uint32_t)((lo & 0xFFFFu) >= thr) << (4 * blk + 0);
        bits |= (uint32_t)((lo >> 16) >= thr) << (4 * blk + 1);
        bits |= (uint32_t)((hi & 0xFFFFu) >= thr) << (4 * blk + 2);
        bits |= (uint32_t)((hi >> 16) >= thr) << (4 * blk + 3);
    }
    return bits;
}

// hidden layer epilogue in train mode: z = acc + bias -> a = gelu(z) [* keep * scale], g = gelu'(z) [* keep * scale];
// a stays in `acc` (next layer's operand) and both go to LDS.
__device__ __forceinline__ void hidden_epilogue(f32x16& acc, const float* __restrict__ bias, int unit0, int h, int c, uint32_t keep,
                                                float scale, float* __restrict__ As, float* __restrict__ Gs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = unit0 + rho(r) + 4 * h;
        const float z = acc[r] + bias[u];
        const float m = ((keep >> r) & 1u) ? scale : 0.0f;
        const float a = gelu(z) * m, g = gelu_grad(z) * m;
        acc[r] = a;
        As[u * kLd + c] = a;
        Gs[u * kLd + c] = g;
    }
}

// dW block rows [32*ot, +32) x cols [32*it, +32) of a layer with n_out x n_in weights: delta in Ds, a_{l-1} in Ap.
template <int OT, int IT>
__device__ __forceinline__ void weight_grads(const float* __restrict__ Ds, const float* __restrict__ Ap, float* __restrict__ gw,
                                             float* __restrict__ gb, int n_out, int n_in, int c, int h) {
#pragma unroll 1
    for (int ot = 0; ot < OT; ++ot) {
        float ad[16]; float bsum = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) { ad[s2] = Ds[(32 * ot + c) * kLd + 2 * s2 + h]; bsum += ad[s2]; }
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && 32 * ot + c < n_out) unsafeAtomicAdd(gb + 32 * ot + c, bsum);
#pragma unroll 1
        for (int it = 0; it < IT; ++it) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[s2], Ap[(32 * it + c) * kLd + 2 * s2 + h], acc, 0, 0, 0);
            const int in = 32 * it + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + rho(r) + 4 * h;
                if (o < n_out && in < n_in) unsafeAtomicAdd(gw + (size_t)o * n_in + in, acc[r]);
            }
        }
    }
}

// delta_{l-1} tile `it` = (W^T . delta_l) * g_{l-1}: W is n_out x n_in row-major, delta_l = KT accumulator tiles.
template <int KT>
__device__ __forceinline__ f32x16 back_tile(const float* __restrict__ w, int n_out, int n_in, int it, const f32x16* d, const float* __restrict__ Gs,
                                            int c, int h) {
    f32x16 acc = zero16();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = 32 * kt + rho(r) + 4 * h;
            const float a = o < n_out ? w[(size_t)o * n_in + 32 * it + c] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[kt][r], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= Gs[(32 * it + rho(r) + 4 * h) * kLd + c];
    return acc;
}

template <bool VEC>
__device__ void train_tile(const TrainArgs& a, float* __restrict__ lds, int row, int lane) {
    const PulseQNet& n = a.net;
    const int c = lane & 31, h = lane >> 5, K1 = n.state_dim, A = n.n_actions;
    const bool live = row >= 0;
    const int rw = max(row, 0);
    float* Xs = lds;                    // [64][kLd]  a_0 = the input rows (zero above state_dim)
    float* A1 = Xs + 64 * kLd;  float* G1 = A1 + 128 * kLd;
    float* A2 = G1 + 128 * kLd; float* G2 = A2 + 128 * kLd;
    float* A3 = G2 + 128 * kLd; float* G3 = A3 + 64 * kLd;
    float* A4 = G3 + 64 * kLd;  float* G4 = A4 + 32 * kLd;
    float* Ds = G4 + 32 * kLd;          // [128][kLd] the current layer's delta
    const size_t o_b1 = (size_t)128 * K1, o_w2 = o_b1 + 128, o_b2 = o_w2 + 128 * 128, o_w3 = o_b2 + 128, o_b3 = o_w3 + 64 * 128,
                 o_w4 = o_b3 + 64, o_b4 = o_w4 + 32 * 64, o_w5 = o_b4 + 32, o_b5 = o_w5 + (size_t)A * 32;

    // target first (registers only): r + gamma * max_a' Q_target(s', a') * (1 - done)                 (:275-277)
    float target;
    {
        const f32x16 qn = forward_eval<VEC>(a.tgt, a.next_states + (size_t)rw * a.next_stride, live, lane);
        float best = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) if (rho(r) + 4 * h < A) best = fmaxf(best, qn[r]);
        best = fmaxf(best, __shfl_xor(best, 32));
        const float notdone = (live && a.dones[rw]) ? 0.0f : 1.0f;
        target = (live ? a.rewards[rw] : 0.0f) + a.gamma * best * notdone;
    }

    // forward, train mode
    const uint64_t gid = a.table_id0 + (uint64_t)rw;
    const uint32_t thr = (uint32_t)(a.drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - a.drop_p);
    const float* xr = a.states + (size_t)rw * a.stride;
    f32x16 h1[4] = {zero16(), zero16(), zero16(), zero16()};
    for (int q = 0; q < 8; ++q) {
        const int k0 = 8 * q + 4 * h;
        float xb[4];
        if (VEC && k0 + 3 < K1) {
            const float4 x4 = *reinterpret_cast<const float4*>(xr + k0);
            xb[0] = x4.x; xb[1] = x4.y; xb[2] = x4.z; xb[3] = x4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[j] = k0 + j < K1 ? xr[k0 + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { xb[j] = live ? xb[j] : 0.0f; Xs[(k0 + j) * kLd + c] = xb[j]; }
        if (8 * q < K1) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                const float* wr = n.w1 + (size_t)(32 * ot + c) * K1 + k0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float wa = k0 + j < K1 ? wr[j] : 0.0f;
                    h1[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb[j], h1[ot], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) hidden_epilogue(h1[ot], n.b1, 32 * ot, h, c, 0xFFFFu, 1.0f, A1, G1);
    f32x16 h2[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        h2[ot] = dense_tile<4>(n.w2, 128, 32 * ot + c, h, h1);
        hidden_epilogue(h2[ot], n.b2, 32 * ot, h, c, dropout_keep_bits(a.seed, gid, a.step, ot, h, thr), scale, A2, G2);   // Dropout(.1) :194
    }
    f32x16 h3[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        h3[ot] = dense_tile<4>(n.w3, 128, 32 * ot + c, h, h2);
        hidden_epilogue(h3[ot], n.b3, 32 * ot, h, c, dropout_keep_bits(a.seed, gid, a.step, 4 + ot, h, thr), scale, A3, G3);   // Dropout(.1) :197
    }
    f32x16 h4[1];
    h4[0] = dense_tile<2>(n.w4, 64, c, h, h3);
    hidden_epilogue(h4[0], n.b4, 0, h, c, 0xFFFFu, 1.0f, A4, G4);
    f32x16 qv = dense_tile<1>(n.w5, 32, min(c, A - 1), h, h4);
    bias_act<false>(qv, n.b5, 0, A, h);

    // delta_5 and the loss terms                                                                     (:270-279)
    const int act = live ? (int)a.actions[rw] : -1;
    float qa = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) qa += (rho(r) + 4 * h == act) ? qv[r] : 0.0f;
    qa += __shfl_xor(qa, 32);
    const float td = live ? qa - target : 0.0f;
    f32x16 d5[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) d5[0][r] = (rho(r) + 4 * h == act) ? 2.0f * td : 0.0f;
    {
        float sq = (h == 0) ? td * td : 0.0f, cnt = (h == 0 && live) ? 1.0f : 0.0f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
        if (lane == 0) { unsafeAtomicAdd(a.stats + 0, cnt); unsafeAtomicAdd(a.stats + 1, sq); }
    }

    // backward, layer 5 .. 1
    __syncthreads();
    store_t(Ds, 0, d5[0], c, h);
    __syncthreads();
    weight_grads<1, 1>(Ds, A4, a.grad + o_w5, a.grad + o_b5, A, 32, c, h);
    f32x16 d4[1];
    d4[0] = back_tile<1>(n.w5, A, 32, 0, d5, G4, c, h);
    __syncthreads();
    store_t(Ds, 0, d4[0], c, h);
    __syncthreads();
    weight_grads<1, 2>(Ds, A3, a.grad + o_w4, a.grad + o_b4, 32, 64, c, h);
    f32x16 d3[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) d3[it] = back_tile<1>(n.w4, 32, 64, it, d4, G3, c, h);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) store_t(Ds, 32 * it, d3[it], c, h);
    __syncthreads();
    weight_grads<2, 4>(Ds, A2, a.grad + o_w3, a.grad + o_b3, 64, 128, c, h);
    f32x16 d2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) d2[it] = back_tile<2>(n.w3, 64, 128, it, d3, G2, c, h);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) store_t(Ds, 32 * it, d2[it], c, h);
    __syncthreads();
    weight_grads<4, 4>(Ds, A1, a.grad + o_w2, a.grad + o_b2, 128, 128, c, h);
    f32x16 d1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) d1[it] = back_tile<4>(n.w2, 128, 128, it, d2, G1, c, h);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) store_t(Ds, 32 * it, d1[it], c, h);
    __syncthreads();
    weight_grads<4, 2>(Ds, Xs, a.grad + 0, a.grad + o_b1, 128, K1, c, h);
    __syncthreads();
}

constexpr size_t kTrainLdsBytes = (size_t)(64 + 128 * 4 + 64 * 2 + 32 * 2 + 128) * kLd * sizeof(float);

// Row filter of train_step as a compaction: ids of the rows with row_mask set and seat status ACTIVE / ALLIN
// (Player.py:261) go to list[1 + i], their number to list[0] (one atomic per wavefront reserves a range; the order
// of the ids is whatever the atomics give, which only permutes the fp32 summation order of the gradient).
__global__ __launch_bounds__(256) void qnet_compact_kernel(const float* __restrict__ states, long long stride,
                                                           const uint8_t* __restrict__ row_mask, int n_rows, int* __restrict__ list) {
    const int row = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    bool sel = row < n_rows && (row_mask == nullptr || row_mask[row] != 0);
    if (sel) {
        const float status = states[(size_t)row * stride + 12];
        sel = status == 0.0f || status == 2.0f;
    }
    const unsigned long long m = __ballot(sel);
    if (m == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(list, __popcll(m));
    base = __shfl(base, 0);
    if (sel) list[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = row;
}

// One wavefront per workgroup, tiles of 32 compacted rows dealt round-robin to the workgroups.
template <bool VEC>
__global__ __launch_bounds__(64) void qnet_train_kernel(const TrainArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int count = a.list[0];
    for (int t0 = 32 * blockIdx.x; t0 < count; t0 += 32 * gridDim.x) {
        const int i = t0 + (lane & 31);
        train_tile<VEC>(a, lds, i < count ? a.list[1 + i] : -1, lane);
    }
}

// Everything between loss.backward() and the end of train_step (Player.py:281-292), one workgroup:
// gradient /= #valid rows (MSELoss mean), clip_grad_norm_(max_norm) (:280), AdamW (torch semantics: decoupled decay,
// bias-corrected moments), step += 1, target sync every update_freq steps (:289-290); clears the gradient and the
// statistics for the next step.  No valid row: nothing moves (the reference returns before the optimizer, :262).
struct AdamArgs {
    float* params; float* target; float* grad; float* m; float* v; long long* step; float* stats; float* report; int* list;
    int n_params; float lr, wd, beta1, beta2, eps, max_norm; int update_freq;
};

__global__ __launch_bounds__(1024) void qnet_adamw_kernel(const AdamArgs a) {
    __shared__ float red[16];
    __shared__ float s_coef;
    const int tid = threadIdx.x;
    const float count = a.stats[0], sq = a.stats[1];
    const float inv = count > 0.0f ? 1.0f / count : 0.0f;
    float ss = 0.0f;
    for (int i = tid; i < a.n_params; i += 1024) { const float g = a.grad[i] * inv; ss += g * g; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int i = 0; i < 16; ++i) t += red[i];
        const float norm = sqrtf(t);
        s_coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);                       // torch.nn.utils.clip_grad_norm_
        a.report[0] = count; a.report[1] = count > 0.0f ? sq * inv : 0.0f; a.report[2] = norm;
    }
    __syncthreads();
    const long long t_new = *a.step + 1;
    __syncthreads();
    if (count > 0.0f) {
        const float coef = s_coef * inv;
        const float bc1 = 1.0f - powf(a.beta1, (float)t_new), bc2 = 1.0f - powf(a.beta2, (float)t_new);
        const float step_size = a.lr / bc1, bc2_sqrt = sqrtf(bc2);
        const bool sync = a.update_freq > 0 && (t_new % a.update_freq) == 0;
        for (int i = tid; i < a.n_params; i += 1024) {
            const float g = a.grad[i] * coef;
            float p = a.params[i] * (1.0f - a.lr * a.wd);
            const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
            const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
            const float denom = sqrtf(v) / bc2_sqrt + a.eps;
            p -= step_size * (m / denom);
            a.m[i] = m; a.v[i] = v; a.params[i] = p;
            if (sync) a.target[i] = p;
        }
        if (tid == 0) *a.step = t_new;
    }
    for (int i = tid; i < a.n_params; i += 1024) a.grad[i] = 0.0f;
    if (tid == 0) { a.stats[0] = 0.0f; a.stats[1] = 0.0f; a.list[0] = 0; }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

int launch(const QNetArgs& a, void* stream) {
    const PulseQNet& n = a.net;
    if (a.n_rows < 0) return pulse::fail(PULSE_EINVAL, "pulse_qnet: n_rows < 0");
    if (n.state_dim < 1 || n.state_dim > 4096 || n.n_actions < 1 || n.n_actions > 32)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet: state_dim must be 1..4096 and n_actions 1..32");
    if (!n.w1 || !n.b1 || !n.w2 || !n.b2 || !n.w3 || !n.b3 || !n.w4 || !n.b4 || !n.w5 || !n.b5 || !a.states)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet: null weight or state pointer");
    if (a.row_stride < n.state_dim) return pulse::fail(PULSE_EINVAL, "pulse_qnet: row_stride < state_dim");
    if (!aligned16(n.w2) || !aligned16(n.w3) || !aligned16(n.w4) || !aligned16(n.w5))
        return pulse::fail(PULSE_EINVAL, "pulse_qnet: weight matrices must be 16-byte aligned");
    if (a.n_rows == 0) return 0;
    const bool vec = n.state_dim % 8 == 0 && a.row_stride % 4 == 0 && aligned16(a.states) && aligned16(n.w1);
    const bool select = a.seat_idx != nullptr;
    const unsigned grid = select ? (unsigned)((a.n_rows + 63) / 64) : (unsigned)((a.n_rows + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    if (select) { if (vec) hipLaunchKernelGGL((qnet_kernel<true, true>), dim3(grid), dim3(64), 0, st, a); else hipLaunchKernelGGL((qnet_kernel<true, false>), dim3(grid), dim3(64), 0, st, a); }
    else { if (vec) hipLaunchKernelGGL((qnet_kernel<false, true>), dim3(grid), dim3(64), 0, st, a); else hipLaunchKernelGGL((qnet_kernel<false, false>), dim3(grid), dim3(64), 0, st, a); }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_qnet launch");
}

}  // namespace

extern "C" {

int pulse_qnet_forward(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, float* q_out, void* stream) {
    if (!net || !q_out) return pulse::fail(PULSE_EINVAL, "pulse_qnet_forward: null argument");
    QNetArgs a{};
    a.net = *net; a.states = states; a.row_stride = row_stride; a.n_rows = n_rows; a.q_out = q_out;
    return launch(a, stream);
}

int pulse_qnet_act(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, const int32_t* seat_idx,
                   int32_t q_seat, float epsilon, uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions,
                   float* q_out, void* stream) {
    if (!net || !actions) return pulse::fail(PULSE_EINVAL, "pulse_qnet_act: null argument");
    QNetArgs a{};
    a.net = *net; a.states = states; a.row_stride = row_stride; a.n_rows = n_rows; a.seat_idx = seat_idx; a.q_seat = q_seat;
    a.epsilon = epsilon; a.seed = seed; a.step = step; a.table_id0 = table_id0; a.actions = actions; a.q_out = q_out;
    return launch(a, stream);
}


int pulse_qnet_param_count(int32_t state_dim, int32_t n_actions) {
    if (state_dim < 1 || n_actions < 1 || n_actions > 32) return pulse::fail(PULSE_EINVAL, "pulse_qnet_param_count: bad dimensions");
    return 128 * state_dim + 128 + 128 * 128 + 128 + 64 * 128 + 64 + 32 * 64 + 32 + 32 * n_actions + n_actions;
}

int pulse_qnet_train_step(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                          const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                          const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter, uint64_t table_id0,
                          void* stream) {
    if (!t || !states || !actions || !rewards || !next_states || !dones)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: null argument");
    const PulseQNet& n = t->net;
    if (n.state_dim < 13 || n.state_dim > 64 || n.n_actions < 1 || n.n_actions > 32)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: state_dim must be 13..64 (column 12 is the seat status) and n_actions 1..32");
    if (t->target.state_dim != n.state_dim || t->target.n_actions != n.n_actions)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: target network shape differs");
    if (!t->params || !t->target_params || !t->grad || !t->exp_avg || !t->exp_avg_sq || !t->step || !t->stats || !t->report || !t->row_list)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: null optimizer buffer");
    const int np = pulse_qnet_param_count(n.state_dim, n.n_actions);
    // the ten tensors of each network must be the views of the flat buffers in the documented order
    const float* expect = t->params; const float* expect_t = t->target_params;
    const float* got[10] = {n.w1, n.b1, n.w2, n.b2, n.w3, n.b3, n.w4, n.b4, n.w5, n.b5};
    const float* got_t[10] = {t->target.w1, t->target.b1, t->target.w2, t->target.b2, t->target.w3, t->target.b3, t->target.w4, t->target.b4,
                              t->target.w5, t->target.b5};
    const int sizes[10] = {128 * n.state_dim, 128, 128 * 128, 128, 64 * 128, 64, 32 * 64, 32, 32 * n.n_actions, n.n_actions};
    for (int i = 0; i < 10; ++i) {
        if (got[i] != expect || got_t[i] != expect_t)
            return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: network tensors are not the views of the flat parameter buffers");
        expect += sizes[i]; expect_t += sizes[i];
    }
    if (row_stride < n.state_dim || next_stride < n.state_dim) return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: row stride < state_dim");
    if (!aligned16(t->params) || !aligned16(t->target_params))
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: flat parameter buffers must be 16-byte aligned");
    if (!(t->dropout_p >= 0.0f && t->dropout_p < 1.0f)) return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: dropout_p outside [0, 1)");
    if (n_rows < 0 || n_rows > t->row_list_capacity)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: n_rows outside [0, row_list_capacity]");
    hipStream_t st = (hipStream_t)stream;
    if (n_rows > 0) {
        TrainArgs a{};
        a.net = t->net; a.tgt = t->target; a.grad = t->grad; a.stats = t->stats; a.states = states; a.stride = row_stride;
        a.actions = actions; a.rewards = rewards; a.next_states = next_states; a.next_stride = next_stride; a.dones = dones;
        a.list = t->row_list; a.seed = seed; a.step = step_counter; a.table_id0 = table_id0;
        a.gamma = t->gamma; a.drop_p = t->dropout_p;
        const bool vec = n.state_dim % 8 == 0 && row_stride % 4 == 0 && next_stride % 4 == 0 && aligned16(states) && aligned16(next_states);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_train_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsBytes);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_train_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsBytes);
            if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_qnet_train_step: LDS size attribute");
            attr_set = true;
        }
        hipLaunchKernelGGL(qnet_compact_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, states, (long long)row_stride, row_mask,
                           n_rows, t->row_list);
        // one wavefront (118 KB of LDS) per CU and round: 256 workgroups cover the usual few hundred tiles in one or two
        const unsigned grid = (unsigned)std::min((n_rows + 31) / 32, 512);
        if (vec) hipLaunchKernelGGL((qnet_train_kernel<true>), dim3(grid), dim3(64), kTrainLdsBytes, st, a);
        else hipLaunchKernelGGL((qnet_train_kernel<false>), dim3(grid), dim3(64), kTrainLdsBytes, st, a);
    }
    AdamArgs b{};
    b.params = t->params; b.target = t->target_params; b.grad = t->grad; b.m = t->exp_avg; b.v = t->exp_avg_sq; b.step = (long long*)t->step;
    b.stats = t->stats; b.report = t->report; b.list = t->row_list; b.n_params = np; b.lr = t->lr; b.wd = t->weight_decay; b.beta1 = t->beta1; b.beta2 = t->beta2;
    b.eps = t->eps; b.max_norm = t->max_grad_norm; b.update_freq = t->update_freq;
    hipLaunchKernelGGL(qnet_adamw_kernel, dim3(1), dim3(1024), 0, st, b);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_qnet_train_step launch");
}

}  // extern "C"
