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

// ================================================================ cooperative tiles: 4 wavefronts per 32 rows
// The single-wavefront tile above keeps one SIMD busy for 512 dependent MFMAs plus 176 erf evaluations per lane.  The
// masked action selection and the training step run a tile on a whole CU instead: a workgroup of 4 wavefronts owns
// 256 candidate rows, compacts the selected ones, and pushes them 32 at a time through the network with the OUTPUT
// tiles of every layer (or, where a layer has fewer than four, slices of its k range) dealt to the wavefronts.
// Activations travel between layers through LDS as [unit][row] (pitch 33): that is the B-operand layout for any k
// order, and read with the row as k it is the A / B layout of the weight-gradient products.
constexpr int kLd = 33;

struct CoopLds {                 // offsets in floats into the dynamic LDS block
    static constexpr int Xs = 0, A1 = Xs + 64 * kLd, A2 = A1 + 128 * kLd, A3 = A2 + 128 * kLd, A4 = A3 + 64 * kLd,
                         P = A4 + 32 * kLd,                    // 3 x 16 x 64 partial sums
                         List = P + 3 * 16 * 64,               // 256 row ids + 8 counters
                         Tgt = List + 264,                     // 32 TD targets
                         EndEval = Tgt + 32,
                         G1 = EndEval, G2 = G1 + 128 * kLd, G3 = G2 + 128 * kLd, G4 = G3 + 64 * kLd,
                         Da = G4 + 32 * kLd, Db = Da + 128 * kLd, EndTrain = Db + 128 * kLd;
};
constexpr size_t kActLdsBytes = (size_t)CoopLds::EndEval * sizeof(float);
constexpr size_t kTrainLdsBytes = (size_t)CoopLds::EndTrain * sizeof(float);

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

// acc[out, row] = sum over k in [k0, k1) of W[out_row][k] * S[k][row]; k0, k1 multiples of 8.  VEC: W rows are
// 16-byte aligned and K % 8 == 0 (a float4 feeds four MFMAs); else scalar loads guarded by k < K.
template <bool VEC>
__device__ __forceinline__ f32x16 dense_lds(const float* __restrict__ w, int K, int out_row, int c, int h, const float* __restrict__ S,
                                            int k0, int k1) {
    f32x16 acc = zero16();
    const float* wr = w + (size_t)out_row * K;
    for (int k8 = k0; k8 < k1; k8 += 8) {
        const int k = k8 + 4 * h;
        float wa[4];
        if (VEC) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
            wa[0] = w4.x; wa[1] = w4.y; wa[2] = w4.z; wa[3] = w4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[j] = k + j < K ? wr[k + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], S[(k + j) * kLd + c], acc, 0, 0, 0);
    }
    return acc;
}

// hidden layer epilogue: z = acc + bias -> a = gelu(z) * m to As; TRAIN also g = gelu'(z) * m to Gs (m = dropout keep * scale)
template <bool TRAIN>
__device__ __forceinline__ void coop_epilogue(const f32x16& acc, const float* __restrict__ bias, int unit0, int c, int h, uint32_t keep,
                                              float scale, float* __restrict__ As, float* __restrict__ Gs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = unit0 + rho(r) + 4 * h;
        const float z = acc[r] + bias[u];
        const float m = ((keep >> r) & 1u) ? scale : 0.0f;
        As[u * kLd + c] = gelu(z) * m;
        if (TRAIN) Gs[u * kLd + c] = gelu_grad(z) * m;
    }
}

// 32 rows of `x` (row ids per column in `rowc`, < 0 = padding) -> Xs[k][column], zero above state_dim; all 4 wavefronts
__device__ __forceinline__ void coop_load_rows(float* __restrict__ lds, const float* __restrict__ x, long long stride, int K1, int rowc,
                                               int wv, int c, int h) {
    const float* xr = x + (size_t)max(rowc, 0) * stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * wv + 8 * h + j;
        lds[CoopLds::Xs + k * kLd + c] = (rowc >= 0 && k < K1) ? xr[k] : 0.0f;
    }
}

// The network on the 32 rows in Xs.  Returns the Q tile in wavefront 0 (other wavefronts: unspecified).  Leaves
// a_1..a_4 (and g_1..g_4 when TRAIN) in LDS; ends on a barrier-free state: callers barrier before reusing LDS.
template <bool TRAIN, bool VEC>
__device__ __forceinline__ f32x16 coop_forward(const PulseQNet& n, float* __restrict__ lds, int wv, int c, int h, uint64_t seed,
                                               uint64_t gid, uint64_t step, uint32_t thr, float scale) {
    float* Xs = lds + CoopLds::Xs; float* A1 = lds + CoopLds::A1; float* A2 = lds + CoopLds::A2; float* A3 = lds + CoopLds::A3;
    float* A4 = lds + CoopLds::A4; float* P = lds + CoopLds::P;
    float* G1 = lds + CoopLds::G1; float* G2 = lds + CoopLds::G2; float* G3 = lds + CoopLds::G3; float* G4 = lds + CoopLds::G4;
    const int lane = c + 32 * h, K1 = n.state_dim, K1r = (K1 + 7) & ~7;
    __syncthreads();                                                          // Xs complete
    {   // layer 1: wavefront wv -> units [32wv, +32)
        const f32x16 acc = dense_lds<VEC>(n.w1, K1, 32 * wv + c, c, h, Xs, 0, K1r);
        coop_epilogue<TRAIN>(acc, n.b1, 32 * wv, c, h, 0xFFFFu, 1.0f, A1, G1);
    }
    __syncthreads();
    {   // layer 2 (+ Dropout, Player.py:194)
        const f32x16 acc = dense_lds<true>(n.w2, 128, 32 * wv + c, c, h, A1, 0, 128);
        const uint32_t keep = TRAIN ? dropout_keep_bits(seed, gid, step, wv, h, thr) : 0xFFFFu;
        coop_epilogue<TRAIN>(acc, n.b2, 32 * wv, c, h, keep, TRAIN ? scale : 1.0f, A2, G2);
    }
    __syncthreads();
    {   // layer 3: 2 output tiles x 2 halves of k (+ Dropout, :197)
        const int ot = wv & 1, half = wv >> 1;
        f32x16 acc = dense_lds<true>(n.w3, 128, 32 * ot + c, c, h, A2, 64 * half, 64 * half + 64);
        if (half == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(ot * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += P[(ot * 16 + r) * 64 + lane];
            const uint32_t keep = TRAIN ? dropout_keep_bits(seed, gid, step, 4 + ot, h, thr) : 0xFFFFu;
            coop_epilogue<TRAIN>(acc, n.b3, 32 * ot, c, h, keep, TRAIN ? scale : 1.0f, A3, G3);
        }
    }
    __syncthreads();
    {   // layer 4: one output tile, k in quarters
        f32x16 acc = dense_lds<true>(n.w4, 64, c, c, h, A3, 16 * wv, 16 * wv + 16);
        if (wv > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[((wv - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += (P[r * 64 + lane] + P[(16 + r) * 64 + lane]) + P[(32 + r) * 64 + lane];
            coop_epilogue<TRAIN>(acc, n.b4, 0, c, h, 0xFFFFu, 1.0f, A4, G4);
        }
    }
    __syncthreads();
    f32x16 qv = zero16();
    if (wv == 0) {
        qv = dense_lds<true>(n.w5, 32, min(c, n.n_actions - 1), c, h, A4, 0, 32);
        bias_act<false>(qv, n.b5, 0, n.n_actions, h);
    }
    return qv;
}

// 256 candidate rows -> ids of the selected ones in List[0..count), count returned to every thread
__device__ __forceinline__ int coop_compact(float* __restrict__ lds, bool sel, int row) {
    int* list = reinterpret_cast<int*>(lds + CoopLds::List);
    int* wcount = list + 256;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(sel);
    if (lane == 0) wcount[wv] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int n = wcount[i]; base += i < wv ? n : 0; total += n; }
    if (sel) list[base + __popcll(m & ((1ull << lane) - 1ull))] = row;
    __syncthreads();
    return total;
}

// ---- masked action selection, cooperative (pulse_qnet_act with seat_idx) ------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void qnet_act4_kernel(const QNetArgs a) {
    extern __shared__ float lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const bool sel = row < a.n_rows && a.seat_idx[row] == a.q_seat;
    const int count = coop_compact(lds, sel, row);
    const int* list = reinterpret_cast<const int*>(lds + CoopLds::List);
    const int A = a.net.n_actions;
    for (int t0 = 0; t0 < count; t0 += 32) {
        const int rowc = t0 + c < count ? list[t0 + c] : -1;
        __syncthreads();                                                      // previous tile's readers are done
        coop_load_rows(lds, a.states, a.row_stride, a.net.state_dim, rowc, wv, c, h);
        const f32x16 qv = coop_forward<false, VEC>(a.net, lds, wv, c, h, 0, 0, 0, 0, 1.0f);
        if (wv == 0) {
            const bool live = rowc >= 0;
            if (a.q_out && live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int o = rho(r) + 4 * h; if (o < A) a.q_out[(size_t)rowc * A + o] = qv[r]; }
            }
            float best = -INFINITY; int arg = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = rho(r) + 4 * h;
                if (o < A && (qv[r] > best || (qv[r] == best && o < arg) || arg == 0x7fffffff)) { best = qv[r]; arg = o; }
            }
            const float ob = __shfl_xor(best, 32); const int oa = __shfl_xor(arg, 32);
            if (oa != 0x7fffffff && (arg == 0x7fffffff || ob > best || (ob == best && oa < arg))) { best = ob; arg = oa; }
            if (live && h == 0) {
                const U4 rnd = philox4x32(a.seed, a.table_id0 + (uint64_t)rowc, a.step);
                const bool explore = rand_unit(rnd.x) < a.epsilon;                               // Player.py:247
                a.actions[rowc] = explore ? (int64_t)rand_below(rnd.y, A) : (int64_t)arg;        // :248-250
            }
        }
    }
}

// ================================================================ training step (Player.py:255-294)
// One launch does, for the rows that pass the reference's filters, what train_step does between its masks and
// `loss.backward()`: TD target from the target network, forward in train mode (dropout after the 2nd and 3rd GELU),
// d(loss)/d(parameters) -- accumulated UNNORMALISED (the 1 / #valid-rows of MSELoss, the norm clipping and AdamW
// follow in qnet_adamw_kernel, which knows the global row count).  Per tile of 32 rows, on one CU:
//   target    cooperative forward of the target network on s', max over actions -> Tgt[row];
//   forward   cooperative forward of the network on s; every hidden layer leaves a_l and g_l = gelu'(z_l) * dropout
//             scale in LDS;  delta_5 = 2 (q[action] - target) on the action's row of the output tile;
//   backward  per layer, dealt to the 4 wavefronts: the 32x32 blocks of dW_l = delta_l . a_{l-1}^T (both operands read
//             out of LDS with the row index as the MFMA k: 16 MFMAs per block, then one f32 atomic per element into the
//             flat gradient; db_l falls out of the same reads) and the tiles of
//             delta_{l-1} = (W_l^T . delta_l) * g_{l-1} (A operand = W_l read down its columns, coalesced).
// fp32 atomics make the summation order of the gradient vary from run to run (rounding-level differences).
struct TrainArgs {
    PulseQNet net, tgt;
    float* grad; float* stats;                // flat gradient (layout: w1,b1,...,w5,b5), stats[0]=#valid rows, [1]=sum td^2
    const float* states; long long stride;
    const int64_t* actions; const float* rewards;
    const float* next_states; long long next_stride;
    const uint8_t* dones; const uint8_t* row_mask;
    int n_rows;
    uint64_t seed, step, table_id0;
    float gamma, drop_p;
};

// block (ot, it) of dW for a layer with n_out x n_in weights: delta in D, a_{l-1} in Ap; BIAS: also db rows of tile ot
template <bool BIAS>
__device__ __forceinline__ void dw_block(const float* __restrict__ D, const float* __restrict__ Ap, float* __restrict__ gw,
                                         float* __restrict__ gb, int n_out, int n_in, int ot, int it, int c, int h) {
    float ad[16]; float bsum = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) { ad[s2] = D[(32 * ot + c) * kLd + 2 * s2 + h]; bsum += ad[s2]; }
    if (BIAS) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && 32 * ot + c < n_out) unsafeAtomicAdd(gb + 32 * ot + c, bsum);
    }
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

// tile `it` of delta_{l-1} = (W^T . delta_l) * g_{l-1} -> Dn[32 it ..]; W is n_out x n_in, delta_l = units [0, ku) of D
__device__ __forceinline__ void back_block(const float* __restrict__ w, int n_out, int n_in, int it, const float* __restrict__ D, int ku,
                                           const float* __restrict__ G, float* __restrict__ Dn, int c, int h) {
    f32x16 acc = zero16();
    for (int k2 = 0; k2 < ku; k2 += 2) {
        const int k = k2 + h;
        const float wa = k < n_out ? w[(size_t)k * n_in + 32 * it + c] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, D[k * kLd + c], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = 32 * it + rho(r) + 4 * h;
        Dn[u * kLd + c] = acc[r] * G[u * kLd + c];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void qnet_train_kernel(const TrainArgs a) {
    extern __shared__ float lds[];
    const PulseQNet& n = a.net;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int K1 = n.state_dim, A = n.n_actions;
    const int row = blockIdx.x * 256 + threadIdx.x;
    bool sel = row < a.n_rows && (a.row_mask == nullptr || a.row_mask[row] != 0);
    if (sel) {                                                   // seat status ACTIVE or ALLIN, Player.py:261
        const float status = a.states[(size_t)row * a.stride + 12];
        sel = status == 0.0f || status == 2.0f;
    }
    const int count = coop_compact(lds, sel, row);
    const int* list = reinterpret_cast<const int*>(lds + CoopLds::List);
    float* Xs = lds + CoopLds::Xs; float* A1 = lds + CoopLds::A1; float* A2 = lds + CoopLds::A2; float* A3 = lds + CoopLds::A3;
    float* A4 = lds + CoopLds::A4; float* Tgt = lds + CoopLds::Tgt;
    float* G1 = lds + CoopLds::G1; float* G2 = lds + CoopLds::G2; float* G3 = lds + CoopLds::G3; float* G4 = lds + CoopLds::G4;
    float* Da = lds + CoopLds::Da; float* Db = lds + CoopLds::Db;
    const size_t o_b1 = (size_t)128 * K1, o_w2 = o_b1 + 128, o_b2 = o_w2 + 128 * 128, o_w3 = o_b2 + 128, o_b3 = o_w3 + 64 * 128,
                 o_w4 = o_b3 + 64, o_b4 = o_w4 + 32 * 64, o_w5 = o_b4 + 32, o_b5 = o_w5 + (size_t)A * 32;
    const uint32_t thr = (uint32_t)(a.drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - a.drop_p);

    for (int t0 = 0; t0 < count; t0 += 32) {
        const int rowc = t0 + c < count ? list[t0 + c] : -1;
        const bool live = rowc >= 0;
        const int rw = max(rowc, 0);
        const uint64_t gid = a.table_id0 + (uint64_t)rw;
        // target: r + gamma * max_a' Q_target(s', a') * (1 - done)                                   (:275-277)
        __syncthreads();
        coop_load_rows(lds, a.next_states, a.next_stride, K1, rowc, wv, c, h);
        {
            const f32x16 qn = coop_forward<false, VEC>(a.tgt, lds, wv, c, h, 0, 0, 0, 0, 1.0f);
            if (wv == 0) {
                float best = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) if (rho(r) + 4 * h < A) best = fmaxf(best, qn[r]);
                best = fmaxf(best, __shfl_xor(best, 32));
                const float notdone = (live && a.dones[rw]) ? 0.0f : 1.0f;
                if (h == 0) Tgt[c] = (live ? a.rewards[rw] : 0.0f) + a.gamma * best * notdone;
            }
        }
        __syncthreads();
        // forward, train mode
        coop_load_rows(lds, a.states, a.stride, K1, rowc, wv, c, h);
        {
            const f32x16 qv = coop_forward<true, VEC>(n, lds, wv, c, h, a.seed, gid, a.step, thr, scale);
            if (wv == 0) {                                                    // delta_5 and the loss terms (:270-279)
                const int act = live ? (int)a.actions[rw] : -1;
                float qa = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) qa += (rho(r) + 4 * h == act) ? qv[r] : 0.0f;
                qa += __shfl_xor(qa, 32);
                const float td = live ? qa - Tgt[c] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) Da[(rho(r) + 4 * h) * kLd + c] = (rho(r) + 4 * h == act) ? 2.0f * td : 0.0f;
                float sq = (h == 0) ? td * td : 0.0f, cnt = (h == 0 && live) ? 1.0f : 0.0f;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
                if (lane == 0) { unsafeAtomicAdd(a.stats + 0, cnt); unsafeAtomicAdd(a.stats + 1, sq); }
            }
        }
        __syncthreads();
        // layer 5 (delta_5 in Da): dW5 | delta_4 -> Db
        if (wv == 0) dw_block<true>(Da, A4, a.grad + o_w5, a.grad + o_b5, A, 32, 0, 0, c, h);
        if (wv == 1) back_block(n.w5, A, 32, 0, Da, 32, G4, Db, c, h);
        __syncthreads();
        // layer 4 (delta_4 in Db): dW4 blocks on wavefronts 0, 1 | delta_3 tiles on 2, 3 -> Da
        if (wv == 0) dw_block<true>(Db, A3, a.grad + o_w4, a.grad + o_b4, 32, 64, 0, 0, c, h);
        if (wv == 1) dw_block<false>(Db, A3, a.grad + o_w4, a.grad + o_b4, 32, 64, 0, 1, c, h);
        if (wv >= 2) back_block(n.w4, 32, 64, wv - 2, Db, 32, G3, Da, c, h);
        __syncthreads();
        // layer 3 (delta_3 in Da): 8 dW blocks, 2 per wavefront | delta_2 tile wv -> Db
        dw_block<false>(Da, A2, a.grad + o_w3, a.grad + o_b3, 64, 128, 0, wv, c, h);
        dw_block<false>(Da, A2, a.grad + o_w3, a.grad + o_b3, 64, 128, 1, wv, c, h);
        if (wv < 2) {                                                         // db3 rows of tile wv
            float bsum = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bsum += Da[(32 * wv + c) * kLd + 2 * s2 + h];
            bsum += __shfl_xor(bsum, 32);
            if (h == 0) unsafeAtomicAdd(a.grad + o_b3 + 32 * wv + c, bsum);
        }
        back_block(n.w3, 64, 128, wv, Da, 64, G2, Db, c, h);
        __syncthreads();
        // layer 2 (delta_2 in Db): 16 dW blocks, column tile wv of each row tile | delta_1 tile wv -> Da
#pragma unroll 1
        for (int ot = 0; ot < 4; ++ot) dw_block<false>(Db, A1, a.grad + o_w2, a.grad + o_b2, 128, 128, ot, wv, c, h);
        {
            float bsum = 0.0f;                                                // db2 rows of tile wv
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bsum += Db[(32 * wv + c) * kLd + 2 * s2 + h];
            bsum += __shfl_xor(bsum, 32);
            if (h == 0) unsafeAtomicAdd(a.grad + o_b2 + 32 * wv + c, bsum);
        }
        back_block(n.w2, 128, 128, wv, Db, 128, G1, Da, c, h);
        __syncthreads();
        // layer 1 (delta_1 in Da): row tile wv x the two column tiles of the input
        dw_block<true>(Da, Xs, a.grad + 0, a.grad + o_b1, 128, K1, wv, 0, c, h);
        if (K1 > 32) dw_block<false>(Da, Xs, a.grad + 0, a.grad + o_b1, 128, K1, wv, 1, c, h);
    }
}

// Everything between loss.backward() and the end of train_step (Player.py:281-292), one workgroup:
// gradient /= #valid rows (MSELoss mean), clip_grad_norm_(max_norm) (:280), AdamW (torch semantics: decoupled decay,
// bias-corrected moments), step += 1, target sync every update_freq steps (:289-290); clears the gradient and the
// statistics for the next step.  No valid row: nothing moves (the reference returns before the optimizer, :262).
struct AdamArgs {
    float* params; float* target; float* grad; float* m; float* v; long long* step; float* stats; float* report;
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
    if (tid == 0) { a.stats[0] = 0.0f; a.stats[1] = 0.0f; }
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
    if (select && n.state_dim > 64) {        // wider inputs than the cooperative tile's LDS image: one wavefront per tile
        if (vec) hipLaunchKernelGGL((qnet_kernel<true, true>), dim3(grid), dim3(64), 0, st, a); else hipLaunchKernelGGL((qnet_kernel<true, false>), dim3(grid), dim3(64), 0, st, a);
    } else if (select) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_act4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLdsBytes);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_act4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLdsBytes);
            if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_qnet_act: LDS size attribute");
            attr_set = true;
        }
        const unsigned g4 = (unsigned)((a.n_rows + 255) / 256);
        if (vec) hipLaunchKernelGGL((qnet_act4_kernel<true>), dim3(g4), dim3(256), kActLdsBytes, st, a);
        else hipLaunchKernelGGL((qnet_act4_kernel<false>), dim3(g4), dim3(256), kActLdsBytes, st, a);
    }
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
    if (!t->params || !t->target_params || !t->grad || !t->exp_avg || !t->exp_avg_sq || !t->step || !t->stats || !t->report)
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
    if (n_rows < 0) return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: n_rows < 0");
    hipStream_t st = (hipStream_t)stream;
    if (n_rows > 0) {
        TrainArgs a{};
        a.net = t->net; a.tgt = t->target; a.grad = t->grad; a.stats = t->stats; a.states = states; a.stride = row_stride;
        a.actions = actions; a.rewards = rewards; a.next_states = next_states; a.next_stride = next_stride; a.dones = dones;
        a.row_mask = row_mask; a.n_rows = n_rows; a.seed = seed; a.step = step_counter; a.table_id0 = table_id0;
        a.gamma = t->gamma; a.drop_p = t->dropout_p;
        const bool vec = n.state_dim % 8 == 0 && row_stride % 4 == 0 && next_stride % 4 == 0 && aligned16(states) && aligned16(next_states);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_train_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsBytes);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_train_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsBytes);
            if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_qnet_train_step: LDS size attribute");
            attr_set = true;
        }
        // one workgroup (4 wavefronts, 149 KB of LDS) per 256 candidate rows: a CU per tile of 32 valid rows
        const unsigned grid = (unsigned)((n_rows + 255) / 256);
        if (vec) hipLaunchKernelGGL((qnet_train_kernel<true>), dim3(grid), dim3(256), kTrainLdsBytes, st, a);
        else hipLaunchKernelGGL((qnet_train_kernel<false>), dim3(grid), dim3(256), kTrainLdsBytes, st, a);
    }
    AdamArgs b{};
    b.params = t->params; b.target = t->target_params; b.grad = t->grad; b.m = t->exp_avg; b.v = t->exp_avg_sq; b.step = (long long*)t->step;
    b.stats = t->stats; b.report = t->report; b.n_params = np; b.lr = t->lr; b.wd = t->weight_decay; b.beta1 = t->beta1; b.beta2 = t->beta2;
    b.eps = t->eps; b.max_norm = t->max_grad_norm; b.update_freq = t->update_freq;
    hipLaunchKernelGGL(qnet_adamw_kernel, dim3(1), dim3(1024), 0, st, b);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_qnet_train_step launch");
}

}  // extern "C"
