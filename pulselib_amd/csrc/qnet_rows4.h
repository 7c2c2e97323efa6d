// qnet_rows4.h -- the learner's network in eval mode on FOUR rows per wavefront (masked action selection).
//
// The cooperative 32-row tile of qnet_device.h pushes one tile through five barrier-separated layers on four wavefronts: at
// 65,536 tables every workgroup has ONE tile, and the launch takes as long as that tile's dependent chain (~16 us of a 23 us
// launch), whatever the chip could do in parallel.  Here a row never leaves its wavefront:
//
//   * v_mfma_f32_4x4x1_16B_f32 multiplies, in each of its 16 blocks, a 4-vector A by a 4-vector B.  With the A-broadcast
//     controls (CBSZ = 4, ABID = b0) block b0's A goes to all blocks, so ONE instruction computes, for 4 rows and 64 units,
//     acc[row i][unit 4b + j] += x[row i][k] * W[unit 4b + j][k]: A = the four rows' input k (lanes 4 b0 + i), B = one weight
//     per lane.  Layout checked on the hardware by tools/probes/mfma4x4_probe.hip: D register i, lane 4b + j =
//     A[lane 4 b0 + i] * B[lane 4b + j].  Same fp32 rate as the 32x32x2 form (64 FLOP / cycle / SIMD), 8 cycles an instruction.
//   * The accumulators come out as (register = row, lane = unit); the next layer wants (register j0, lane 4b + i = row i's
//     value of unit 4b + j0): a 4x4 transpose inside every quad, two DPP butterfly steps (16 instructions per 64 units).
//     No LDS, no barrier between layers.
//   * The weights are what every row needs again: the whole network sits in the workgroup's LDS in operand order
//     ([step of 4 inputs][unit] -> float4, 126 KB + biases), one conflict-free ds_read_b128 per four MFMAs, filled once per
//     launch by a persistent workgroup of 16 wavefronts (one per CU).
//
// 10,900 of 65,536 rows are the learner's: 11 four-row wavefronts per CU instead of two 32-row tiles, each 560 MFMAs (4.5 K
// cycles) with nothing to wait for but its own LDS reads.  Per-row results are those of the tiles up to the order of the sums
// (k ascending in two or four interleaved partial sums).  Not part of the ABI.
#pragma once
#include "qnet_device.h"

namespace pulse_qnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kR4Win = 256;          // candidate rows per window (two of the training launch's list windows)
constexpr int kR4Threads = 1024;     // 16 wavefronts: one workgroup per CU, four wavefronts per SIMD at <= 128 registers

// A matrix of U units x KQ steps sits in LDS as float4 [step][unit] at a pitch of U + 1: the reads (one step, consecutive units)
// and the fill's writes (one unit, consecutive steps -- the order the rows lie in memory) are both free of bank conflicts.
template <int KQ1> struct R4Lds {    // offsets in floats; KQ1 = steps of 4 inputs in layer 1 (state_dim <= 4 KQ1)
    static constexpr int W1 = 0, W2 = W1 + KQ1 * 129 * 4, W3 = W2 + 32 * 129 * 4, W4 = W3 + 32 * 65 * 4, W5 = W4 + 16 * 33 * 4,
                         B1 = W5 + 8 * 17 * 4, B2 = B1 + 128, B3 = B2 + 128, B4 = B3 + 64, B5 = B4 + 32,
                         List = B5 + 16,                    // 256 row ids, 4 + 4 wavefront counts
                         End = List + 256 + 8;
    static constexpr size_t bytes = (size_t)End * sizeof(float);
};

template <int AB> __device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0);       // block AB's four A values (rows) to all sixteen blocks
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E;                  // quad_perm:[1,0,3,2] / [2,3,0,1]
constexpr int kDppRowRor1 = 0x121, kDppRowRor2 = 0x122, kDppRowRor4 = 0x124, kDppRowRor8 = 0x128;

// One network matrix into operand order: dst[kq * (U + 1) + u] (float4) = W[u][4 kq .. 4 kq + 3], zero past the real units /
// inputs.  Consecutive threads take consecutive steps of one unit: whole rows of W, coalesced.
template <int U, int KQ>
__device__ __forceinline__ void r4_fill(float* __restrict__ dst, const float* __restrict__ w, int u_real, int K) {
    for (int i = threadIdx.x; i < KQ * U; i += kR4Threads) {
        const int u = i / KQ, kq = i - u * KQ;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (u < u_real && 4 * kq < K) v = *reinterpret_cast<const float4*>(w + (size_t)u * K + 4 * kq);
        reinterpret_cast<float4*>(dst)[kq * (U + 1) + u] = v;
    }
}
#if PULSE_STAMPS
#define R4STAMP(i) do { if ((threadIdx.x & 63) == 0 && g_qstamp_buf) { __builtin_amdgcn_sched_barrier(0); \
    g_qstamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define R4STAMP(i) do { } while (0)
#endif

// The MFMAs of steps KQ .. KQ + CH - 1 for input j0 of each: one instruction per accumulator (ND groups of 64 units x CH
// partial sums), so that consecutive instructions never depend on each other (a dependent 4x4x1 waits 18 cycles, an
// independent one issues after 8 -- the probe).
template <int KQ, int C, int CH, int ND, int NG>
__device__ __forceinline__ void r4_mfma_c(const float (&T)[NG][4], const float (&w)[ND][CH][4], f32x4 (&D)[ND * CH], int j0) {
    if constexpr (C < CH) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (PULSE_QABL & 1) D[d * CH + C][j0] += T[(KQ + C) / 16][j0] * w[d][C][j0];        // (timeline ablation: no MFMAs)
            else D[d * CH + C] = mfma4<(KQ + C) % 16>(T[(KQ + C) / 16][j0], w[d][C][j0], D[d * CH + C]);
        }
        r4_mfma_c<KQ, C + 1, CH, ND, NG>(T, w, D, j0);
    }
}
// Steps KQ .. KQN - 1 of a layer with U units in LDS (lanes past U read the units again: their products are never used) whose
// input is T (NG groups of 64 units in A-operand order): D[d * CH + c] += sum over the steps kq = c mod CH.  The weights of a
// group of CH steps are read one group ahead of the group being multiplied and no further (left alone the compiler hoists
// every read of a layer to its top: 360 spilled registers).
template <int KQ, int U, int ND, int CH>
__device__ __forceinline__ void r4_read(float (&w)[ND][CH][4], const float4* __restrict__ Wl, int lane) {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float4 v = Wl[((PULSE_QABL & 8) ? c : KQ + c) * (U + 1) + ((64 * d + lane) & (U - 1))];   // (ablation 8: a layer's reads collapse to its first group's)
            w[d][c][0] = v.x; w[d][c][1] = v.y; w[d][c][2] = v.z; w[d][c][3] = v.w;
        }
    }
}
template <int KQ, int KQN, int U, int ND, int CH, int NG>
__device__ __forceinline__ void r4_steps_from(const float (&w)[ND][CH][4], const float4* __restrict__ Wl, const float (&T)[NG][4],
                                              f32x4 (&D)[ND * CH], int lane) {
    float wn[ND][CH][4];
    if constexpr (KQ + CH < KQN) r4_read<KQ + CH, U, ND, CH>(wn, Wl, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0) r4_mfma_c<KQ, 0, CH, ND, NG>(T, w, D, j0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KQ + CH < KQN) r4_steps_from<KQ + CH, KQN, U, ND, CH, NG>(wn, Wl, T, D, lane);
}
template <int KQ, int KQN, int U, int ND, int CH, int NG>
__device__ __forceinline__ void r4_steps(const float4* __restrict__ Wl, const float (&T)[NG][4], f32x4 (&D)[ND * CH], int lane) {
    float w[ND][CH][4];
    r4_read<KQ, U, ND, CH>(w, Wl, lane);
    r4_steps_from<KQ, KQN, U, ND, CH, NG>(w, Wl, T, D, lane);
}

// 64 units x 4 rows out of an accumulator (register = row, lane = unit): bias, GELU, and into A-operand order
// (T[j0], lane 4b + i = row i's activation of unit 4b + j0) by a 4x4 transpose inside every quad.
__device__ __forceinline__ void r4_hidden(const f32x4& acc, float bias, float (&T)[4], int lane) {
    const f32x2 z01 = {acc[0] + bias, acc[1] + bias}, z23 = {acc[2] + bias, acc[3] + bias};
    f32x2 y01, y23, dy;
    gelu_pair2(z01, y01, dy); gelu_pair2(z23, y23, dy);
    const float m0 = y01.x, m1 = y01.y, m2 = y23.x, m3 = y23.y;
    const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
    // step 1 exchanges bit 0 of (register, quad lane), step 2 bit 1: T[r][quad lane j] = m[j][quad lane r]
    const float x0 = dpp_f<kDppQuadXor1>(m0), x1 = dpp_f<kDppQuadXor1>(m1), x2 = dpp_f<kDppQuadXor1>(m2), x3 = dpp_f<kDppQuadXor1>(m3);
    const float s0 = odd ? x1 : m0, s1 = odd ? m1 : x0, s2 = odd ? x3 : m2, s3 = odd ? m3 : x2;
    const float y0 = dpp_f<kDppQuadXor2>(s0), y1 = dpp_f<kDppQuadXor2>(s1), y2 = dpp_f<kDppQuadXor2>(s2), y3 = dpp_f<kDppQuadXor2>(s3);
    T[0] = hi ? y2 : s0; T[2] = hi ? s2 : y0; T[1] = hi ? y3 : s1; T[3] = hi ? s3 : y1;
}

// The five layers on the four rows whose inputs X holds in A-operand order (lane 4b + i: X[j0] = row i's input 4b + j0, zero
// past state_dim and for padding rows): Q values as (register = row, lane & 15 = action).
template <int KQ1>
__device__ __forceinline__ f32x4 r4_forward(const float* __restrict__ lds, const float (&X)[1][4], int lane) {
    using L = R4Lds<KQ1>;
    float T2[2][4], T3[2][4], T4[1][4], T5[1][4];
    {
        f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
        r4_steps<0, KQ1, 128, 2, 2, 1>(reinterpret_cast<const float4*>(lds + L::W1), X, D, lane);
        r4_hidden(D[0] + D[1], lds[L::B1 + lane], T2[0], lane);
        r4_hidden(D[2] + D[3], lds[L::B1 + 64 + lane], T2[1], lane);
    }
    {
        f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
        r4_steps<0, 32, 128, 2, 2, 2>(reinterpret_cast<const float4*>(lds + L::W2), T2, D, lane);
        r4_hidden(D[0] + D[1], lds[L::B2 + lane], T3[0], lane);
        r4_hidden(D[2] + D[3], lds[L::B2 + 64 + lane], T3[1], lane);
    }
    {
        f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
        r4_steps<0, 32, 64, 1, 4, 2>(reinterpret_cast<const float4*>(lds + L::W3), T3, D, lane);
        r4_hidden((D[0] + D[1]) + (D[2] + D[3]), lds[L::B3 + lane], T4[0], lane);
    }
    {   // 32 units: the upper 32 lanes repeat them (their activations sit in blocks 8..15, which no step below selects)
        f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
        r4_steps<0, 16, 32, 1, 4, 1>(reinterpret_cast<const float4*>(lds + L::W4), T4, D, lane);
        r4_hidden((D[0] + D[1]) + (D[2] + D[3]), lds[L::B4 + (lane & 31)], T5[0], lane);
    }
    f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
    r4_steps<0, 8, 16, 1, 4, 1>(reinterpret_cast<const float4*>(lds + L::W5), T5, D, lane);
    const float b5 = lds[L::B5 + (lane & 15)];
    f32x4 q = (D[0] + D[1]) + (D[2] + D[3]);
    q[0] += b5; q[1] += b5; q[2] += b5; q[3] += b5;
    return q;
}

// pulse_qnet_act with seat_idx (and the trainer's row lists, as act_window writes them): persistent workgroups, windows of 256
// candidate rows; the learner's rows of a window are listed in LDS and taken four at a time by the 16 wavefronts.
template <int KQ1>
__global__ __launch_bounds__(kR4Threads) void qnet_act_r4_kernel(const QNetArgs a) {
    extern __shared__ float lds[];
    using L = R4Lds<KQ1>;
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int K1 = a.net.state_dim, A = a.net.n_actions;
    R4STAMP(0);
    const int n_win = (a.n_rows + kR4Win - 1) / kR4Win;
    // a window's candidate words (seat index, terminated flag, seat status) are loaded one window ahead: the first window's
    // arrive while the weights are copied, the next one's while this one's rows go through the layers
    int c_seat = -1; bool c_term = false; float c_status = 1.0f;
    auto load_candidates = [&](int win) {
        const int row = win * kR4Win + tid;
        c_seat = -1; c_term = false; c_status = 1.0f;
        if (tid < kR4Win && win < n_win && row < a.n_rows) {
            c_seat = a.seat_idx[row];
            if (a.terminated) c_term = a.terminated[row] != 0;
            if (a.tsel_counts) c_status = a.states[(size_t)row * a.row_stride + 12];
        }
    };
    load_candidates((int)blockIdx.x);
    r4_fill<128, KQ1>(lds + L::W1, a.net.w1, 128, K1);
    r4_fill<128, 32>(lds + L::W2, a.net.w2, 128, 128);
    r4_fill<64, 32>(lds + L::W3, a.net.w3, 64, 128);
    r4_fill<32, 16>(lds + L::W4, a.net.w4, 32, 64);
    r4_fill<16, 8>(lds + L::W5, a.net.w5, A, 32);
    if (tid < 128) { lds[L::B1 + tid] = a.net.b1[tid]; lds[L::B2 + tid] = a.net.b2[tid]; }
    if (tid < 64) lds[L::B3 + tid] = a.net.b3[tid];
    if (tid < 32) lds[L::B4 + tid] = a.net.b4[tid];
    if (tid < 16) lds[L::B5 + tid] = tid < A ? a.net.b5[tid] : 0.0f;
    int* const list = reinterpret_cast<int*>(lds + L::List);
    int* const wcount = list + 256;
    int* const tcount = wcount + 4;
    for (int win = blockIdx.x; win < n_win; win += gridDim.x) {
        const int row = win * kR4Win + tid;
        const bool cand = tid < kR4Win && row < a.n_rows;
        const bool sel = cand && c_seat == a.q_seat;
        const bool live = sel && !c_term;
        if (a.row_mask_out && cand) a.row_mask_out[row] = live ? 1 : 0;       // the trainer's `q_mask & ~terminated` (trainGPU.py:85)
        // the rows the next training launch takes (row_mask & seat status ACTIVE / ALLIN, Player.py:258-261), per window of 128
        const bool tsel = a.tsel_counts && live && (c_status == 0.0f || c_status == 2.0f);
        const unsigned long long m = __ballot(sel), tm = __ballot(tsel);
        if (wv < 4 && lane == 0) { wcount[wv] = __popcll(m); tcount[wv] = __popcll(tm); }
        load_candidates(win + (int)gridDim.x);
        R4STAMP(1);
        __syncthreads();                       // (also: every wavefront has read its rows of the previous window's list)
        R4STAMP(2);
        int base = 0, count = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int n = wcount[i]; base += i < wv ? n : 0; count += n; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (sel) list[base + __popcll(m & below)] = row;
        if (a.tsel_counts && wv < 4) {
            const int w128 = 2 * win + (wv >> 1), tbase = (wv & 1) ? tcount[wv - 1] : 0;
            if (tsel) a.tsel_rows[(size_t)w128 * 128 + tbase + __popcll(tm & below)] = row;
            if (lane == 0 && !(wv & 1) && w128 * 128 < a.n_rows) a.tsel_counts[w128] = tcount[wv] + tcount[wv + 1];
        }
        __syncthreads();
        R4STAMP(3);
        for (int t0 = 4 * wv; t0 < count; t0 += 4 * (kR4Threads / 64)) {
            const int b = lane >> 2, i = lane & 3;
            const int rowi = t0 + i < count ? list[t0 + i] : -1;
            float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (rowi >= 0 && 4 * b < K1) x = *reinterpret_cast<const float4*>(a.states + (size_t)rowi * a.row_stride + 4 * b);
            const float X[1][4] = {{x.x, x.y, x.z, x.w}};
            R4STAMP(4);
            const f32x4 q = r4_forward<KQ1>(lds, X, lane);
            R4STAMP(5);
            // Q output, first maximal index per row (torch.argmax) over the 16 lanes of a DPP row, epsilon draw, action
            const int u = lane & 15;
            int arg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = __builtin_amdgcn_readlane(rowi, r);                     // lane r is (block 0, row r)
                if (a.q_out && rr >= 0 && lane < A) a.q_out[(size_t)rr * A + lane] = q[r];
                float bv = u < A ? q[r] : -INFINITY; int bi = u < A ? u : 0x7fffffff;
#define PULSE_R4_STEP(CTRL) { const float ov = dpp_f<CTRL>(bv); const int oi = dpp_i<CTRL>(bi); const bool take = ov > bv || (ov == bv && oi < bi); bv = take ? ov : bv; bi = take ? oi : bi; }
                PULSE_R4_STEP(kDppRowRor1) PULSE_R4_STEP(kDppRowRor2) PULSE_R4_STEP(kDppRowRor4) PULSE_R4_STEP(kDppRowRor8)
#undef PULSE_R4_STEP
                arg[r] = bi;
            }
            if (lane < 4 && rowi >= 0) {                                               // lane i of block 0 finishes row i
                const int mine = lane == 0 ? arg[0] : lane == 1 ? arg[1] : lane == 2 ? arg[2] : arg[3];
                const U4 rnd = philox4x32(a.seed, a.table_id0 + (uint64_t)rowi, a.step);
                const bool explore = rand_unit(rnd.x) < a.epsilon;                               // Player.py:247
                a.actions[rowi] = explore ? (int64_t)rand_below(rnd.y, A) : (int64_t)min(mine, A - 1);   // :248-250
            }
            R4STAMP(6);
        }
    }
    R4STAMP(7);
}

}  // namespace pulse_qnet
