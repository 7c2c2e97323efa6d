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

// The network on up to 32 rows: lane (c, h) carries row `row` (< 0 = padding column) as column c.
template <bool VEC>
__device__ __forceinline__ void qnet_tile(const QNetArgs& a, int row, int lane) {
    const PulseQNet& n = a.net;
    const int c = lane & 31, h = lane >> 5, K1 = n.state_dim;
    const float* xr = a.states + (size_t)max(row, 0) * a.row_stride;
    const bool live = row >= 0;

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

}  // extern "C"
