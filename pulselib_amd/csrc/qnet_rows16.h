// qnet_rows16.h -- the learner's network in eval mode on SIXTEEN rows per wavefront (masked action selection).
//
// The cooperative 32-row tile of qnet_device.h pushes one tile through five barrier-separated layers on four wavefronts: at
// 65,536 tables every workgroup has ONE tile and the launch takes as long as that tile's dependent chain.  Here a row never
// leaves its wavefront and a layer never waits for another wavefront:
//
//   * v_mfma_f32_16x16x4_f32 with the weights as A (16 units x 4 inputs) and the rows as B (4 inputs x 16 rows).  Operand
//     layout checked on the hardware by tools/probes/mfma16x16_probe.hip: A lane l = (unit l % 16, input l / 16), B lane l =
//     (row l % 16, input l / 16), D register r of lane l = (unit 4 (l / 16) + r, row l % 16).  So the four accumulator
//     registers of an output tile ARE four B operands of the next layer (inputs 16 t + 4 (l / 16) + r, r = 0..3), and the A
//     operand that goes with them is a float4 of one weight row: the activations stay in registers from the observation to the
//     Q values -- no LDS round trip, no transpose, no barrier.  Exact fp32 (the reference's dtype), k summed in ascending groups.
//   * The weights are what every row needs again: the whole network sits in the workgroup's LDS in operand order (one
//     ds_read_b128 per lane feeds four MFMAs; 141 KB + biases), copied once per launch by a persistent workgroup of 16
//     wavefronts, one per CU; the windows of candidate rows are taken in turn, their learner's rows listed in LDS and dealt
//     16 at a time to the wavefronts.
//
// Measured and replaced on the way (same interface, profiles/README.md): four rows per wavefront on v_mfma_f32_4x4x1 with the
// A broadcast (tools/probes/mfma4x4_probe.hip): one more transpose per layer and, decisive, a two-pass MFMA leaves the SIMD
// no issue slot for other work (cost = 8 cycles x MFMAs + 4 x vector instructions: 1.8 K cycles per row against 1.6 K here) -- no
// faster than the 32-row tiles at 65,536 tables (22.6 us; this form 18.8) and 395 against 314 us at 2,000,000.
// Not part of the ABI.
#pragma once
#include "qnet_device.h"

namespace pulse_qnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kR16Threads = 1024;    // 16 wavefronts: one workgroup per CU, four wavefronts per SIMD at <= 128 registers
constexpr int kR16Rec = 68;          // float4s per (output tile, input group) record: 4 runs of 16 units at a pitch of 17 -- the
                                     // reads (16 consecutive units per run) and the fill's writes (consecutive inputs of one unit)
                                     // are both free of bank conflicts

// LDS map, offsets in floats.  MT1 = groups of 16 inputs in layer 1 (state_dim <= 16 MT1); WIN = candidate rows per window.
template <int MT1, int WIN> struct R16Lds {
    static constexpr int W1 = 0, W2 = W1 + 8 * MT1 * kR16Rec * 4, W3 = W2 + 8 * 8 * kR16Rec * 4, W4 = W3 + 4 * 8 * kR16Rec * 4,
                         W5 = W4 + 2 * 4 * kR16Rec * 4, B1 = W5 + 1 * 2 * kR16Rec * 4, B2 = B1 + 128, B3 = B2 + 128, B4 = B3 + 64,
                         B5 = B4 + 32, List = B5 + 16,          // WIN row ids, 16 + 16 wavefront counts
                         End = List + WIN + 32;
    static constexpr size_t bytes = (size_t)End * sizeof(float);
};

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }

#if PULSE_STAMPS
#define R16STAMP(i) do { if ((threadIdx.x & 63) == 0 && g_qstamp_buf) { __builtin_amdgcn_sched_barrier(0); \
    g_qstamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define R16STAMP(i) do { } while (0)
#endif

// One network matrix (u_real x K, torch layout) into operand order: record (mo, mt) holds, at kk * 17 + m, the float4
// W[16 mo + m][16 mt + 4 kk .. + 3] (zero past the real units / inputs).  Consecutive threads take consecutive float4s of
// one row of W: coalesced.  In two halves -- all of a thread's loads of ALL matrices go out before its first LDS store (a
// load-store loop is one L2 round trip per iteration: the copy was 8.4 K cycles of a 50 K-cycle launch).
template <int MO, int MT> struct R16Part { static constexpr int KQ = 4 * MT, CNT = 16 * MO * KQ, N = (CNT + kR16Threads - 1) / kR16Threads; };
template <int MO, int MT>
__device__ __forceinline__ void r16_fill_load(float4* v, const float* __restrict__ w, int u_real, int K) {
    using P = R16Part<MO, MT>;
#pragma unroll
    for (int j = 0; j < P::N; ++j) {
        const int i = threadIdx.x + j * kR16Threads, u = i / P::KQ, kq = i - u * P::KQ;
        v[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < P::CNT && u < u_real && 4 * kq < K) v[j] = *reinterpret_cast<const float4*>(w + (size_t)u * K + 4 * kq);
    }
}
template <int MO, int MT>
__device__ __forceinline__ void r16_fill_store(float* __restrict__ dst, const float4* v) {
    using P = R16Part<MO, MT>;
#pragma unroll
    for (int j = 0; j < P::N; ++j) {
        const int i = threadIdx.x + j * kR16Threads, u = i / P::KQ, kq = i - u * P::KQ;
        if (i < P::CNT) reinterpret_cast<float4*>(dst)[((u >> 4) * MT + (kq >> 2)) * kR16Rec + (kq & 3) * 17 + (u & 15)] = v[j];
    }
}

// bias + GELU on an output tile: the result is one input group of the next layer (same lanes, same registers' roles)
__device__ __forceinline__ void r16_hidden(const f32x4& acc, const float4& b, float (&Bn)[4]) {
    const f32x2 z01 = {acc[0] + b.x, acc[1] + b.y}, z23 = {acc[2] + b.z, acc[3] + b.w};
    f32x2 y01, y23, dy;
    gelu_pair2(z01, y01, dy); gelu_pair2(z23, y23, dy);
    Bn[0] = y01.x; Bn[1] = y01.y; Bn[2] = y23.x; Bn[3] = y23.y;
}

// A layer of 16 MO units on 16 MT inputs: D[mo] += W-tile(mo, mt) . B[mt] over the input groups mt.  The A operands of one
// input group and up to four output tiles (16 registers) are read one such stage ahead of the stage being multiplied and no
// further (left alone the compiler hoists a layer's reads to its top and spills); within a stage consecutive MFMAs go to
// different accumulators (a dependent 16x16x4 waits 52 cycles, an independent one issues after 32 -- the probe).
// (Measured and dropped: finishing input group mt + 1 -- bias, GELU -- among the MFMAs of group mt, one MFMA to four vector
// instructions by sched_group_barrier: not faster.  An fp32 MFMA does not run beside vector instructions on this chip, from the
// same wavefront or another: a SIMD's time is 32 cycles x MFMAs + 4 x vector instructions, tools/probes/mfma16x16_probe.hip.)
template <int MO, int MT>
__device__ __forceinline__ void r16_layer(const float4* __restrict__ Wl, const float (&B)[MT][4], f32x4 (&D)[MO], int lane) {
    constexpr int G = MO < 4 ? MO : 4, NG = MO / G, NS = MT * NG;            // stage s = (input group s / NG, output tiles G (s % NG) ..)
    const int pos = (lane >> 4) * 17 + (lane & 15);
    float a[2][G][4];
    auto read = [&](int s, int buf) {
        const int mt = s / NG, mo0 = G * (s % NG);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const float4 v = Wl[((mo0 + j) * MT + mt) * kR16Rec + pos];
            a[buf][j][0] = v.x; a[buf][j][1] = v.y; a[buf][j][2] = v.z; a[buf][j][3] = v.w;
        }
    };
    read(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) read(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int mt = s / NG, mo0 = G * (s % NG);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < G; ++j) D[mo0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][j][r], B[mt][r], D[mo0 + j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bias + GELU on the MO output tiles of a layer: the results are the input groups of the next one (same lanes).  (Reading the
// layer's biases in one batch ahead of the GELUs costs registers the kernel does not have: spills, no gain.)
template <int MO>
__device__ __forceinline__ void r16_hidden(const f32x4 (&D)[MO], const float* __restrict__ bias, int g, float (&H)[MO][4]) {
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
        const float4 b = *reinterpret_cast<const float4*>(bias + 16 * mo + 4 * g);
        const f32x2 z01 = {D[mo][0] + b.x, D[mo][1] + b.y}, z23 = {D[mo][2] + b.z, D[mo][3] + b.w};
        f32x2 y01, y23, dy;
        gelu_pair2(z01, y01, dy); gelu_pair2(z23, y23, dy);
        H[mo][0] = y01.x; H[mo][1] = y01.y; H[mo][2] = y23.x; H[mo][3] = y23.y;
    }
}

// The five layers on the 16 rows whose inputs X holds as B operands (lane (n, g): X[mt][r] = row n's input 16 mt + 4 g + r, zero
// past state_dim and for padding rows): Q values of row n, actions 4 g + r in register r.
template <int MT1, int WIN>
__device__ __forceinline__ f32x4 r16_forward(const float* __restrict__ lds, const float (&X)[MT1][4], int lane) {
    using L = R16Lds<MT1, WIN>;
    const int g = lane >> 4;
    float H1[8][4], H2[8][4], H3[4][4], H4[2][4];
    {
        f32x4 D[8] = {zero4(), zero4(), zero4(), zero4(), zero4(), zero4(), zero4(), zero4()};
        r16_layer<8, MT1>(reinterpret_cast<const float4*>(lds + L::W1), X, D, lane);
        r16_hidden<8>(D, lds + L::B1, g, H1);
    }
    {
        f32x4 D[8] = {zero4(), zero4(), zero4(), zero4(), zero4(), zero4(), zero4(), zero4()};
        r16_layer<8, 8>(reinterpret_cast<const float4*>(lds + L::W2), H1, D, lane);
        r16_hidden<8>(D, lds + L::B2, g, H2);
    }
    {
        f32x4 D[4] = {zero4(), zero4(), zero4(), zero4()};
        r16_layer<4, 8>(reinterpret_cast<const float4*>(lds + L::W3), H2, D, lane);
        r16_hidden<4>(D, lds + L::B3, g, H3);
    }
    {
        f32x4 D[2] = {zero4(), zero4()};
        r16_layer<2, 4>(reinterpret_cast<const float4*>(lds + L::W4), H3, D, lane);
        r16_hidden<2>(D, lds + L::B4, g, H4);
    }
    f32x4 D[1] = {zero4()};
    r16_layer<1, 2>(reinterpret_cast<const float4*>(lds + L::W5), H4, D, lane);
    const float4 b5 = *reinterpret_cast<const float4*>(lds + L::B5 + 4 * g);
    f32x4 q = D[0];
    q[0] += b5.x; q[1] += b5.y; q[2] += b5.z; q[3] += b5.w;
    return q;
}

// pulse_qnet_act with seat_idx (and the trainer's row lists, as act_window writes them): persistent workgroups, windows of WIN
// candidate rows (256: one window per CU at 65,536 tables; 1,024 for large batches: ~11 tiles for the 16 wavefronts); the
// learner's rows of a window are listed in LDS and taken 16 at a time by the wavefronts.
template <int MT1, int WIN>
__global__ __launch_bounds__(kR16Threads) void qnet_act_r16_kernel(const QNetArgs a) {
    extern __shared__ float lds[];
    using L = R16Lds<MT1, WIN>;
    constexpr int NWV = WIN / 64;                                  // wavefronts that hold candidates
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int K1 = a.net.state_dim, A = a.net.n_actions;
    R16STAMP(0);
    const int n_win = (a.n_rows + WIN - 1) / WIN;
    // A window's candidate words are loaded ahead of its turn: seat index and terminated flag TWO windows ahead, the seat status
    // (observation column 12) ONE window ahead and only for the learner's live rows -- read for every candidate it pulled every
    // line of the observation through the chip (320 MB at 2,000,000 tables, a quarter of the launch).  The first window's words
    // arrive while the weights are copied.
    const int stride_w = (int)gridDim.x;
    int c_seat = -1, n_seat = -1; bool c_term = false, n_term = false; float c_status = 1.0f;
    auto load_seat = [&](int win, int& seat, bool& term) {
        const int row = win * WIN + tid;
        seat = -1; term = false;
        if (tid < WIN && win < n_win && row < a.n_rows) {
            seat = a.seat_idx[row];
            if (a.terminated) term = a.terminated[row] != 0;
        }
    };
    auto load_status = [&](int win, int seat, bool term) -> float {
        const int row = win * WIN + tid;
        const bool inside = tid < WIN && win < n_win && row < a.n_rows;
        return (a.tsel_counts && inside && seat == a.q_seat && !term) ? a.states[(size_t)row * a.row_stride + 12] : 1.0f;
    };
    load_seat((int)blockIdx.x, c_seat, c_term);
    load_seat((int)blockIdx.x + stride_w, n_seat, n_term);
    c_status = load_status((int)blockIdx.x, c_seat, c_term);
    {
        float4 v1[R16Part<8, MT1>::N], v2[R16Part<8, 8>::N], v3[R16Part<4, 8>::N], v4[R16Part<2, 4>::N], v5[R16Part<1, 2>::N];
        r16_fill_load<8, MT1>(v1, a.net.w1, 128, K1);
        r16_fill_load<8, 8>(v2, a.net.w2, 128, 128);
        r16_fill_load<4, 8>(v3, a.net.w3, 64, 128);
        r16_fill_load<2, 4>(v4, a.net.w4, 32, 64);
        r16_fill_load<1, 2>(v5, a.net.w5, A, 32);
        float bias = 0.0f;                                   // 368 bias words: b1, b2, b3, b4, b5 (zero past the real actions) in one run
        if (tid < 128) bias = a.net.b1[tid];
        else if (tid < 256) bias = a.net.b2[tid - 128];
        else if (tid < 320) bias = a.net.b3[tid - 256];
        else if (tid < 352) bias = a.net.b4[tid - 320];
        else if (tid < 352 + A) bias = a.net.b5[tid - 352];
        r16_fill_store<8, MT1>(lds + L::W1, v1);
        r16_fill_store<8, 8>(lds + L::W2, v2);
        r16_fill_store<4, 8>(lds + L::W3, v3);
        r16_fill_store<2, 4>(lds + L::W4, v4);
        r16_fill_store<1, 2>(lds + L::W5, v5);
        if (tid < 368) lds[L::B1 + tid] = bias;
    }
    int* const list = reinterpret_cast<int*>(lds + L::List);
    int* const wcount = list + WIN;
    int* const tcount = wcount + 16;
    for (int win = blockIdx.x; win < n_win; win += gridDim.x) {
        const int row = win * WIN + tid;
        const bool cand = tid < WIN && row < a.n_rows;
        const bool sel = cand && c_seat == a.q_seat;
        const bool live = sel && !c_term;
        if (a.row_mask_out && cand) a.row_mask_out[row] = live ? 1 : 0;       // the trainer's `q_mask & ~terminated` (trainGPU.py:85)
        // the rows the next training launch takes (row_mask & seat status ACTIVE / ALLIN, Player.py:258-261), per window of 128
        const bool tsel = a.tsel_counts && live && (c_status == 0.0f || c_status == 2.0f);
        const unsigned long long m = __ballot(sel), tm = __ballot(tsel);
        if (wv < NWV && lane == 0) { wcount[wv] = __popcll(m); tcount[wv] = __popcll(tm); }
        // next window: its status words (its seat words arrived a window ago); the window after: its seat words
        const float nn_status = load_status(win + stride_w, n_seat, n_term);
        int nn_seat; bool nn_term;
        load_seat(win + 2 * stride_w, nn_seat, nn_term);
        R16STAMP(1);
        __syncthreads();                       // (also: every wavefront has read its rows of the previous window's list)
        R16STAMP(2);
        int base = 0, count = 0;
#pragma unroll 4
        for (int i = 0; i < NWV; ++i) { const int n = wcount[i]; base += i < wv ? n : 0; count += n; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (sel) list[base + __popcll(m & below)] = row;
        if (a.tsel_counts && wv < NWV) {
            const int w128 = (WIN / 128) * win + (wv >> 1), tbase = (wv & 1) ? tcount[wv - 1] : 0;
            if (tsel) a.tsel_rows[(size_t)w128 * 128 + tbase + __popcll(tm & below)] = row;
            if (lane == 0 && !(wv & 1) && w128 * 128 < a.n_rows) a.tsel_counts[w128] = tcount[wv] + tcount[wv + 1];
        }
        __syncthreads();
        R16STAMP(3);
        for (int t0 = 16 * wv; t0 < count; t0 += 16 * (kR16Threads / 64)) {
            const int n = lane & 15, g = lane >> 4;
            const int rown = t0 + n < count ? list[t0 + n] : -1;
            float X[MT1][4];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (rown >= 0 && 16 * mt + 4 * g < K1) x = *reinterpret_cast<const float4*>(a.states + (size_t)rown * a.row_stride + 16 * mt + 4 * g);
                X[mt][0] = x.x; X[mt][1] = x.y; X[mt][2] = x.z; X[mt][3] = x.w;
            }
            R16STAMP(4);
            const f32x4 q = r16_forward<MT1, WIN>(lds, X, lane);
            R16STAMP(5);
            // Q output; first maximal index of the row (torch.argmax): this lane's four actions, then the column's other three lanes
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                if (a.q_out && rown >= 0 && o < A) a.q_out[(size_t)rown * A + o] = q[r];
                if (o < A && (q[r] > bv || bi == 0x7fffffff)) { bv = q[r]; bi = o; }
            }
#pragma unroll
            for (int x = 16; x <= 32; x <<= 1) {
                const float ov = __shfl_xor(bv, x); const int oi = __shfl_xor(bi, x);
                const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi));
                bv = take ? ov : bv; bi = take ? oi : bi;
            }
            if (g == 0 && rown >= 0) {
                const U4 rnd = philox4x32(a.seed, a.table_id0 + (uint64_t)rown, a.step);
                const bool explore = rand_unit(rnd.x) < a.epsilon;                               // Player.py:247
                a.actions[rown] = explore ? (int64_t)rand_below(rnd.y, A) : (int64_t)bi;         // :248-250
            }
            R16STAMP(6);
        }
        c_seat = n_seat; c_term = n_term; c_status = nn_status; n_seat = nn_seat; n_term = nn_term;
    }
    R16STAMP(7);
}

}  // namespace pulse_qnet
