// qnet_device.h -- the learner's network on the matrix cores as device functions (see qnet.hip for the design): the
// cooperative 32-row tile of four wavefronts (group_forward), its helpers, and the masked action selection of one window of
// candidate rows (act_window) -- shared by qnet.hip (the stand-alone kernels, the training kernels) and poker_step.hip (the
// launch that picks the learner's actions AND steps the tables).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pulse_internal.h"

#ifndef PULSE_QABL
#define PULSE_QABL 0      // ablations for the timeline tools (diagnostic builds only; results are wrong): 1 no MFMAs, 2 no GELU math, 4 no dropout draws
#endif
#ifndef QSTAMP
#define QSTAMP(i) do { } while (0)      // (qnet.hip defines it for the stamped diagnostic build)
#endif

namespace pulse_qnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
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
    const uint8_t* terminated; uint8_t* row_mask_out; // masked form only: row_mask_out[r] = selected && !terminated[r]
    int32_t* tsel_rows; int32_t* tsel_counts;         // nullptr or the training launch's row lists (windows of kActWin rows)
    int32_t* asel_rows; int32_t* asel_counts;         // nullptr, or: the two-launch form for large batches -- the window launch only
                                                      // LISTS the learner's rows here, qnet_act_rows_kernel runs them in full tiles
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
                         P = A4 + 32 * kLd,                    // 2 x (3 x 16 x 64) partial sums (two networks in flight in training)
                         List = P + 2 * 3 * 16 * 64,           // act: row ids of the window + wavefront counts; training: first positions of
                                                               // the threads' windows (256) + counts
                         Tgt = List + 520,                     // (List: up to 512 first positions + 8 totals) 32 words: max_a' Q_target per column, wavefront sums at the end
                         EndEval = Tgt + 32,
                         G1 = EndEval, G2 = G1 + 128 * kLd, G3 = G2 + 128 * kLd, G4 = G3 + 64 * kLd,
                         Da = G4 + 32 * kLd, Db = Da + 128 * kLd, EndTrain = Db + 128 * kLd;
};
// The act window's own map: one network in eval mode needs no layer's input after the next layer has consumed it, so the five
// activation buffers are two regions used in turn (x, a_2, a_4 in R0; a_1, a_3 in R1), and one network's exchange space:
// 47 KB instead of 80 -- three workgroups fit a CU's LDS (the registers, 168 at three wavefronts per SIMD, allow it too).
struct ActLds {
    static constexpr int R0 = 0, R1 = R0 + 128 * kLd, P = R1 + 128 * kLd, List = P + 3 * 16 * 64, End = List + 264;
};
constexpr size_t kActLdsBytes = (size_t)ActLds::End * sizeof(float);
constexpr int kActWin = 128;
constexpr size_t kTrainLdsBytes = (size_t)CoopLds::EndTrain * sizeof(float);

// GELU and its derivative for the cooperative kernels, sharing one exponential: with e = exp(-x^2 / 2),
//   erf(|x| / sqrt 2) = 1 - (a1 t + ... + a5 t^5) e,  t = 1 / (1 + p |x| / sqrt 2)      (Abramowitz-Stegun 7.1.26,
// |error| <= 1.5e-7, the size of an fp32 rounding of the result), and the density in gelu' is e / sqrt(2 pi).  Branch-free
// and ~20 instructions for the pair; the library erff (three data-dependent branches, all taken in a wavefront
// of mixed arguments) was 60 % of the training kernel.  Differences to torch's erff-based GELU stay at 1e-7 |x|.
__device__ __forceinline__ void gelu_pair(float x, float& y, float& dy) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);          // exp(-x^2 / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));                 // 1 ulp: below the formula's own error
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = 1.0f - poly * e;
    const float cdf = fmaf(copysignf(erf_abs, x), 0.5f, 0.5f);
    y = x * cdf;
    dy = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// The same on two values at once: the multiplies and fused multiply-adds are the packed instructions (v_pk_mul_f32 /
// v_pk_fma_f32, two lanes' worth of fp32 per issue slot); only exp2, rcp and the two sign operations stay per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { f32x2 r = {v, v}; return r; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_pair2(f32x2 x, f32x2& y, f32x2& dy) {
    f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 den = fma2(ax, splat2(0.3275911f * 0.70710678118654752440f), splat2(1.0f));
    const f32x2 ea = (x * x) * splat2(-0.72134752044448170368f);
    const f32x2 e = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 p = fma2(t, splat2(1.061405429f), splat2(-1.453152027f));
    p = fma2(t, p, splat2(1.421413741f));
    p = fma2(t, p, splat2(-0.284496736f));
    p = fma2(t, p, splat2(0.254829592f));
    const f32x2 erf_abs = fma2(t * p, -e, splat2(1.0f));
    const f32x2 s = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
    const f32x2 cdf = fma2(s, splat2(0.5f), splat2(0.5f));
    y = x * cdf;
    dy = fma2(x * splat2(0.39894228040143267794f), e, cdf);
}

// keep-mask bits of the 16 accumulator rows of tile `tile` (units 32*tile + rho(r) + 4h) for table `gid`:
// unit u drops when the 16-bit uniform (call u / 8, word (u % 8) / 2, half u % 2) is below drop_p * 65536.
__device__ __forceinline__ uint32_t dropout_keep_bits(uint64_t seed, uint64_t gid, uint64_t step, int tile, int h, uint32_t thr) {
    {   // keep the ten rounds' key schedule (uniform: 20 scalar registers per call site) from being hoisted out of the tile loop
        uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));
        seed = ((uint64_t)hi << 32) | lo;
    }
    // The two lanes of a column (h = 0, 1) need the same four calls, one half of each call's words: lane h makes calls
    // 2h and 2h + 1 and hands the partner its half of them through the lane pair.
    if (PULSE_QABL & 4) return 0xFFFFu;
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = 2 * h + i;                            // units 32*tile + 8*blk + 4h' + {0,1,2,3} = registers 4*blk + j of half h'
        const U4 w = philox4x32(seed ^ 0xD50F0D50F0ull, gid, step * 32 + (uint64_t)(4 * tile + blk));
        const uint32_t mine_lo = h ? w.z : w.x, mine_hi = h ? w.w : w.y, send_lo = h ? w.x : w.z, send_hi = h ? w.y : w.w;
        const uint32_t got_lo = (uint32_t)__shfl_xor((int)send_lo, 32), got_hi = (uint32_t)__shfl_xor((int)send_hi, 32);
        // this lane's call blk = 2h + i; the partner's = 2(1 - h) + i
        lo[i] = h ? got_lo : mine_lo; hi[i] = h ? got_hi : mine_hi;                  // calls 0, 1
        lo[2 + i] = h ? mine_lo : got_lo; hi[2 + i] = h ? mine_hi : got_hi;          // calls 2, 3
    }
    uint32_t bits = 0;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        bits |= (uint32_t)((lo[blk] & 0xFFFFu) >= thr) << (4 * blk + 0);
        bits |= (uint32_t)((lo[blk] >> 16) >= thr) << (4 * blk + 1);
        bits |= (uint32_t)((hi[blk] & 0xFFFFu) >= thr) << (4 * blk + 2);
        bits |= (uint32_t)((hi[blk] >> 16) >= thr) << (4 * blk + 3);
    }
    return bits;
}

// A layer's product acc[out, row] = sum over NK8 steps of 8 k from k0 of W[out_row][k] * S[k][row] in two halves: the weight loads
// (load_w) and the MFMAs on them (mfma_w), so that a caller can issue the loads a layer ahead (all of a call's loads go out
// together: a loop over k is otherwise one L2 round trip per four MFMAs).  VEC: W rows are 16-byte aligned and K % 8 == 0
// (a float4 feeds four MFMAs); else scalar loads guarded by k < K.
template <bool VEC, int NK8>
__device__ __forceinline__ void load_w(float (&wa)[NK8][4], const float* __restrict__ w, int K, int out_row, int h, int k0, int k1) {
    const float* wr = w + (size_t)out_row * K;
#pragma unroll
    for (int i = 0; i < NK8; ++i) {
        const int k = k0 + 8 * i + 4 * h;
        if (VEC) {
            float4 w4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k0 + 8 * i < k1) w4 = *reinterpret_cast<const float4*>(wr + k);
            wa[i][0] = w4.x; wa[i][1] = w4.y; wa[i][2] = w4.z; wa[i][3] = w4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[i][j] = (k0 + 8 * i < k1 && k + j < K) ? wr[k + j] : 0.0f;
        }
    }
}
// The B operands (one LDS word per MFMA) are read 16 at a time, one chunk ahead of the chunk being multiplied: left to
// itself the compiler reads two, waits, multiplies, and every pair of MFMAs then pays an LDS round trip.  All NK8 steps
// of 8 k from k0 are taken: steps past a layer's inputs multiply zero weights (load_w / load_layer) with rows of S that
// exist and are finite (layer 1: Xs is zero-filled up to 64 inputs).
template <int NK8>
__device__ __forceinline__ f32x16 mfma_w(const float (&wa)[NK8][4], int c, int h, const float* __restrict__ S, int k0) {
    f32x16 acc = zero16();
    constexpr int CH = NK8 < 4 ? NK8 : 4, NCH = (NK8 + CH - 1) / CH;
    float bv[2][CH][4];
    auto read = [&](int ch) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = k0 + 8 * (ch * CH + i) + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[ch & 1][i][j] = (ch * CH + i < NK8) ? S[(k + j) * kLd + c] : 0.0f;
        }
    };
    read(0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) read(ch + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (ch * CH + i < NK8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (PULSE_QABL & 1) acc[j] += wa[ch * CH + i < NK8 ? ch * CH + i : 0][j] * bv[ch & 1][i][j];
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch * CH + i < NK8 ? ch * CH + i : 0][j], bv[ch & 1][i][j], acc, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
// Workgroup barrier for waves that talk through LDS only: waits for this wave's LDS traffic, not for its global loads
// (__syncthreads' release fence is s_waitcnt vmcnt(0) too, which puts every weight load issued ahead of a barrier back on
// the critical path).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A network as the training kernels see it: ONE base pointer into the flat w1,b1,...,w5,b5 buffer (pulse_env.h:
// PulseQNetTrain -- the host checks that the ten tensors are those views).  Twenty pointers per network in scalar registers
// were most of the 200 scalar spills of these kernels (a v_readlane per use).
struct FlatNet { const float* base; int32_t state_dim, n_actions; };
__device__ __forceinline__ int layer_base(int layer, int K1) {
    const int base[5] = {0, 128 * K1 + 128, 128 * K1 + 128 + 128 * 128 + 128, 128 * K1 + 128 + 128 * 128 + 128 + 64 * 128 + 64,
                         128 * K1 + 128 + 128 * 128 + 128 + 64 * 128 + 64 + 32 * 64 + 32};
    return base[layer];
}
__device__ __forceinline__ const float* net_w(const FlatNet& n, int layer) { return n.base + layer_base(layer, n.state_dim); }
__device__ __forceinline__ const float* net_b(const FlatNet& n, int layer) {
    const int nw[5] = {128 * n.state_dim, 128 * 128, 64 * 128, 32 * 64, 32 * n.n_actions};
    return n.base + layer_base(layer, n.state_dim) + nw[layer];
}
__device__ __forceinline__ const float* net_w(const PulseQNet& n, int layer) {
    return layer == 0 ? n.w1 : layer == 1 ? n.w2 : layer == 2 ? n.w3 : layer == 3 ? n.w4 : n.w5;
}
__device__ __forceinline__ const float* net_b(const PulseQNet& n, int layer) {
    return layer == 0 ? n.b1 : layer == 1 ? n.b2 : layer == 2 ? n.b3 : layer == 3 ? n.b4 : n.b5;
}

// This lane's A operands of layer `layer` (0..4): W[out_row][k0 + 8 i + 4 h + j] from the torch layout.
// (Tried and dropped: reading them from a transposed copy [in][out] kept in step by the AdamW launch, so that a wavefront's
// load is two runs of 32 consecutive floats instead of 64 rows x 16 bytes -- no faster, the kernels are not bound by how
// the weights arrive)
template <bool VEC, int NK8, class Net>
__device__ __forceinline__ void load_layer(float (&wa)[NK8][4], const Net& n, int layer, int out_row, int h, int k0, int k1) {
    const int n_in = layer == 0 ? n.state_dim : layer == 1 ? 128 : layer == 2 ? 128 : layer == 3 ? 64 : 32;
    load_w<VEC, NK8>(wa, net_w(n, layer), n_in, out_row, h, k0, k1);
}

// hidden layer epilogue: z = acc + bias -> a = gelu(z) * m to As; TRAIN also g = gelu'(z) * m to Gs (m = dropout keep * scale)
// the 16 biases of this lane's accumulator rows (units unit0 + rho(r) + 4h); `bias` global or LDS
__device__ __forceinline__ void load_bias16(float (&bz)[16], const float* __restrict__ bias, int unit0, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bz[r] = bias[unit0 + rho(r) + 4 * h];
}

template <bool TRAIN>
__device__ __forceinline__ void coop_epilogue(const f32x16& acc, const float (&bz)[16], int unit0, int c, int h, uint32_t keep,
                                              float scale, float* __restrict__ As, float* __restrict__ Gs) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {                             // registers r, r + 1 are units u, u + 1
        const int u = unit0 + rho(r) + 4 * h;
        const f32x2 z = {acc[r] + bz[r], acc[r + 1] + bz[r + 1]};
        const f32x2 m = {((keep >> r) & 1u) ? scale : 0.0f, ((keep >> (r + 1)) & 1u) ? scale : 0.0f};
        f32x2 y, dy;
        if (PULSE_QABL & 2) { y = z; dy = z; } else gelu_pair2(z, y, dy);
        y = y * m;
        As[u * kLd + c] = y.x; As[(u + 1) * kLd + c] = y.y;
        if (TRAIN) { dy = dy * m; Gs[u * kLd + c] = dy.x; Gs[(u + 1) * kLd + c] = dy.y; }
    }
}

// the same for accumulator registers 0..7 of `acc` standing for units unit0 + rho(i) + 4h (no dropout)
template <bool TRAIN>
__device__ __forceinline__ void coop_epilogue8(const f32x16& acc, const float (&bz)[8], int unit0, int c, int h, float* __restrict__ As,
                                               float* __restrict__ Gs) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const int u = unit0 + rho(r) + 4 * h;
        const f32x2 z = {acc[r] + bz[r], acc[r + 1] + bz[r + 1]};
        f32x2 y, dy;
        if (PULSE_QABL & 2) { y = z; dy = z; } else gelu_pair2(z, y, dy);
        As[u * kLd + c] = y.x; As[(u + 1) * kLd + c] = y.y;
        if (TRAIN) { Gs[u * kLd + c] = dy.x; Gs[(u + 1) * kLd + c] = dy.y; }
    }
}

// 32 rows of `x` (row ids per column in `rowc`, < 0 = padding) -> Xs[k][column], zero above state_dim; all 4 wavefronts.
// VEC (16-byte aligned rows of a multiple of 8 inputs): a lane's 8 inputs are two 16-byte loads.
template <bool VEC>
__device__ __forceinline__ void coop_fetch_rows(float (&v)[8], const float* __restrict__ x, long long stride, int K1, int rowc, int wv, int h) {
    const float* xr = x + (size_t)max(rowc, 0) * stride;
    const int k0 = 16 * wv + 8 * h;
    if (VEC) {
        float4 lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hi = lo;
        if (rowc >= 0 && k0 < K1) { lo = *reinterpret_cast<const float4*>(xr + k0); hi = *reinterpret_cast<const float4*>(xr + k0 + 4); }
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (rowc >= 0 && k0 + j < K1) ? xr[k0 + j] : 0.0f;
    }
}
__device__ __forceinline__ void coop_store_rows(float* __restrict__ dst, const float (&v)[8], int wv, int c, int h) {
    const int k0 = 16 * wv + 8 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[(k0 + j) * kLd + c] = v[j];
}
template <bool VEC>
__device__ __forceinline__ void coop_load_rows(float* __restrict__ dst, const float* __restrict__ x, long long stride, int K1, int rowc,
                                               int wv, int c, int h) {
    float v[8];
    coop_fetch_rows<VEC>(v, x, stride, K1, rowc, wv, h);
    coop_store_rows(dst, v, wv, c, h);
}

// Training: the target network on s' (eval) and the network on s (train mode) taken through the layers TOGETHER -- every
// stage issues both networks' MFMAs and epilogues between one pair of barriers, so the two forwards cost 7 barrier
// phases instead of 14.  The target's activations borrow the backward pass's delta buffers (free until then):
// x' and a'_2, a'_4 in Db, a'_1 and a'_3 in Da.  q_tgt / q come back in wavefront 0.  Layer 1's weights (w1t, w1c) are
// loaded by the caller ahead of its row gather; every layer issues the next layer's weight loads behind its MFMAs and the
// barriers in here wait for LDS only, so those loads land during the epilogues.  Biases: global (L2) reads.
template <bool VEC, int NK1>
__device__ __forceinline__ void coop_forward_pair(const float (&w1t)[NK1][4], const float (&w1c)[NK1][4], const FlatNet& nt, const FlatNet& n,
                                                  float* __restrict__ lds, int wv, int c, int h, uint64_t seed, uint64_t gid, uint64_t step,
                                                  uint32_t thr, float scale, f32x16& q_tgt, f32x16& q) {
    float* Xs = lds + CoopLds::Xs; float* A1 = lds + CoopLds::A1; float* A2 = lds + CoopLds::A2; float* A3 = lds + CoopLds::A3;
    float* A4 = lds + CoopLds::A4; float* P = lds + CoopLds::P; float* P2 = P + 3 * 16 * 64;
    float* G1 = lds + CoopLds::G1; float* G2 = lds + CoopLds::G2; float* G3 = lds + CoopLds::G3; float* G4 = lds + CoopLds::G4;
    float* Xn = lds + CoopLds::Db; float* T1 = lds + CoopLds::Da; float* T2 = lds + CoopLds::Db; float* T3 = lds + CoopLds::Da;
    float* T4 = lds + CoopLds::Db;
    const int lane = c + 32 * h;
    const int ot = wv & 1, half = wv >> 1;
    lds_barrier();                                                            // Xs, Xn complete
    float w2t[16][4], w2c[16][4];
    {   // layer 1 (each phase: this layer's biases, the MFMAs, the next layer's weights, the epilogues -- a wait for the
        // biases then never includes the younger weight loads)
        float bt[16], bo[16];
        load_bias16(bt, net_b(nt, 0), 32 * wv, h); load_bias16(bo, net_b(n, 0), 32 * wv, h);
        const f32x16 at = mfma_w<NK1>(w1t, c, h, Xn, 0);
        const f32x16 ac = mfma_w<NK1>(w1c, c, h, Xs, 0);
        load_layer<true, 16>(w2t, nt, 1, 32 * wv + c, h, 0, 128);
        load_layer<true, 16>(w2c, n, 1, 32 * wv + c, h, 0, 128);
        coop_epilogue<false>(at, bt, 32 * wv, c, h, 0xFFFFu, 1.0f, T1, nullptr);
        coop_epilogue<true>(ac, bo, 32 * wv, c, h, 0xFFFFu, 1.0f, A1, G1);
    }
    lds_barrier();
    float w3t[8][4], w3c[8][4];
    {   // layer 2 (+ Dropout on the training side, Player.py:194)
        float bt[16], bo[16];
        load_bias16(bt, net_b(nt, 1), 32 * wv, h); load_bias16(bo, net_b(n, 1), 32 * wv, h);
        const f32x16 at = mfma_w<16>(w2t, c, h, T1, 0);
        const f32x16 ac = mfma_w<16>(w2c, c, h, A1, 0);
        load_layer<true, 8>(w3t, nt, 2, 32 * ot + c, h, 64 * half, 64 * half + 64);
        load_layer<true, 8>(w3c, n, 2, 32 * ot + c, h, 64 * half, 64 * half + 64);
        coop_epilogue<false>(at, bt, 32 * wv, c, h, 0xFFFFu, 1.0f, T2, nullptr);
        coop_epilogue<true>(ac, bo, 32 * wv, c, h, dropout_keep_bits(seed, gid, step, wv, h, thr), scale, A2, G2);
    }
    lds_barrier();
    float w4t[2][4], w4c[2][4];
    {   // layer 3: 2 output tiles x 2 halves of k (+ Dropout, :197).  The two wavefronts of a tile swap partial sums: the one
        // with the lower half of k finishes the target network's tile, the other the trained network's (one epilogue each)
        float bz[16];
        load_bias16(bz, half == 0 ? net_b(nt, 2) : net_b(n, 2), 32 * ot, h);
        f32x16 at = mfma_w<8>(w3t, c, h, T2, 64 * half);
        f32x16 ac = mfma_w<8>(w3c, c, h, A2, 64 * half);
        load_layer<true, 2>(w4t, nt, 3, c, h, 16 * wv, 16 * wv + 16);
        load_layer<true, 2>(w4c, n, 3, c, h, 16 * wv, 16 * wv + 16);
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P2[(ot * 16 + r) * 64 + lane] = ac[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(ot * 16 + r) * 64 + lane] = at[r];
        }
        lds_barrier();
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) at[r] += P[(ot * 16 + r) * 64 + lane];
            coop_epilogue<false>(at, bz, 32 * ot, c, h, 0xFFFFu, 1.0f, T3, nullptr);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ac[r] = P2[(ot * 16 + r) * 64 + lane] + ac[r];
            coop_epilogue<true>(ac, bz, 32 * ot, c, h, dropout_keep_bits(seed, gid, step, 4 + ot, h, thr), scale, A3, G3);
        }
    }
    lds_barrier();
    float w5t[4][4], w5c[4][4];
    {   // layer 4: one output tile per network, k in quarters.  Group g = (network g >> 1, accumulator registers 8 (g & 1) .. +8)
        // is finished by wavefront g: the others send it their partial sums of those registers (3 x 8 words per lane and group)
        const int own_net = wv >> 1, rb = 8 * (wv & 1);
        float bz8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bz8[i] = (own_net ? net_b(n, 3) : net_b(nt, 3))[2 * rb + rho(i) + 4 * h];      // units of registers rb + i: rho(i) + 16 (rb / 8) + 4h
        const f32x16 at = mfma_w<2>(w4t, c, h, T3, 16 * wv);
        const f32x16 ac = mfma_w<2>(w4c, c, h, A3, 16 * wv);
        load_layer<true, 4>(w5t, nt, 4, min(c, n.n_actions - 1), h, 0, wv == 0 ? 32 : 0);
        load_layer<true, 4>(w5c, n, 4, min(c, n.n_actions - 1), h, 0, wv == 0 ? 32 : 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g != wv) {
                const int slot = (wv - g - 1) & 3;                // 0..2
#pragma unroll
                for (int i = 0; i < 8; ++i) P[((g * 3 + slot) * 8 + i) * 64 + lane] = (g >> 1) ? ac[8 * (g & 1) + i] : at[8 * (g & 1) + i];
            }
        }
        lds_barrier();
        f32x16 fin = zero16();                                     // registers 0..7: this wavefront's group
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float own = own_net ? (rb ? ac[8 + i] : ac[i]) : (rb ? at[8 + i] : at[i]);
            fin[i] = own + ((P[((wv * 3 + 0) * 8 + i) * 64 + lane] + P[((wv * 3 + 1) * 8 + i) * 64 + lane]) + P[((wv * 3 + 2) * 8 + i) * 64 + lane]);
        }
        if (own_net) coop_epilogue8<true>(fin, bz8, 2 * rb, c, h, A4, G4);
        else coop_epilogue8<false>(fin, bz8, 2 * rb, c, h, T4, nullptr);
    }
    lds_barrier();
    q_tgt = zero16(); q = zero16();
    if (wv == 0) {
        q_tgt = mfma_w<4>(w5t, c, h, T4, 0);
        bias_act<false>(q_tgt, net_b(nt, 4), 0, n.n_actions, h);
        q = mfma_w<4>(w5c, c, h, A4, 0);
        bias_act<false>(q, net_b(n, 4), 0, n.n_actions, h);
    }
}

template <bool TRAIN, int NR>
__device__ __forceinline__ void coop_epilogue_n(const float (&v)[NR], const float (&bz)[NR], int unit0, int c, int h, uint32_t keep, float scale,
                                                float* __restrict__ As, float* __restrict__ Gs) {
#pragma unroll
    for (int r = 0; r < NR; r += 2) {                             // values r, r + 1 are units u, u + 1 (rho(i) = i below 4)
        const int u = unit0 + rho(r) + 4 * h;
        const f32x2 z = {v[r] + bz[r], v[r + 1] + bz[r + 1]};
        const f32x2 m = {((keep >> r) & 1u) ? scale : 0.0f, ((keep >> (r + 1)) & 1u) ? scale : 0.0f};
        f32x2 y, dy;
        if (PULSE_QABL & 2) { y = z; dy = z; } else gelu_pair2(z, y, dy);
        y = y * m;
        As[u * kLd + c] = y.x; As[(u + 1) * kLd + c] = y.y;
        if (TRAIN) { dy = dy * m; Gs[u * kLd + c] = dy.x; Gs[(u + 1) * kLd + c] = dy.y; }
    }
}

// One network on the 32 rows in X, on the four wavefronts wq = 0..3 of a group; B1..B4 receive a_1..a_4 (G1..G4: g_1..g_4
// when TRAIN), P is the group's 3072 floats of exchange space.  Returns the output tile in wavefront wq = 0.  7 barriers.
template <bool TRAIN, int NK1, class Net>
__device__ __forceinline__ f32x16 group_forward(const float (&w1r)[NK1][4], const Net& n, const float* __restrict__ X,
                                                float* __restrict__ B1, float* __restrict__ B2, float* __restrict__ B3, float* __restrict__ B4,
                                                float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ G3, float* __restrict__ G4,
                                                float* __restrict__ P, int wq, int c, int h, uint64_t seed, uint64_t gid, uint64_t step,
                                                uint32_t thr, float scale) {
    const int lane = c + 32 * h, ot = wq & 1, half = wq >> 1, A = n.n_actions;
    lds_barrier();                                                            // X complete
    float w2r[16][4];
    {   // layer 1
        float bz[16];
        load_bias16(bz, net_b(n, 0), 32 * wq, h);
        const f32x16 acc = mfma_w<NK1>(w1r, c, h, X, 0);
        load_layer<true, 16>(w2r, n, 1, 32 * wq + c, h, 0, 128);
        coop_epilogue<TRAIN>(acc, bz, 32 * wq, c, h, 0xFFFFu, 1.0f, B1, G1);
    }
    lds_barrier();
    float w3r[8][4];
    {   // layer 2 (+ Dropout when TRAIN, Player.py:194)
        float bz[16];
        load_bias16(bz, net_b(n, 1), 32 * wq, h);
        const f32x16 acc = mfma_w<16>(w2r, c, h, B1, 0);
        load_layer<true, 8>(w3r, n, 2, 32 * ot + c, h, 64 * half, 64 * half + 64);
        coop_epilogue<TRAIN>(acc, bz, 32 * wq, c, h, TRAIN ? dropout_keep_bits(seed, gid, step, wq, h, thr) : 0xFFFFu, TRAIN ? scale : 1.0f, B2, G2);
    }
    lds_barrier();
    float w4r[2][4];
    {   // layer 3: 2 output tiles x 2 halves of k; the two wavefronts of a tile finish 8 accumulator registers each (+ Dropout, :197)
        float bz8[8], own[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bz8[i] = net_b(n, 2)[32 * ot + 16 * half + rho(i) + 4 * h];
        const f32x16 acc = mfma_w<8>(w3r, c, h, B2, 64 * half);
        load_layer<true, 2>(w4r, n, 3, c, h, 16 * wq, 16 * wq + 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) P[((ot * 2 + (1 - half)) * 8 + i) * 64 + lane] = half ? acc[i] : acc[8 + i];     // the partner's registers
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) own[i] = (half ? acc[8 + i] : acc[i]) + P[((ot * 2 + half) * 8 + i) * 64 + lane];
        const uint32_t keep = TRAIN ? (dropout_keep_bits(seed, gid, step, 4 + ot, h, thr) >> (8 * half)) : 0xFFFFu;
        coop_epilogue_n<TRAIN, 8>(own, bz8, 32 * ot + 16 * half, c, h, keep, TRAIN ? scale : 1.0f, B3, G3);
    }
    lds_barrier();
    float w5r[4][4];
    {   // layer 4: one output tile, k in quarters; wavefront wq finishes registers 4 wq .. 4 wq + 3 (units 8 wq + i + 4h)
        float bz4[4], own[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bz4[i] = net_b(n, 3)[8 * wq + i + 4 * h];
        const f32x16 acc = mfma_w<2>(w4r, c, h, B3, 16 * wq);
        load_layer<true, 4>(w5r, n, 4, min(c, A - 1), h, 0, wq == 0 ? 32 : 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g != wq) {
                const int slot = (wq - g - 1) & 3;                // 0..2
#pragma unroll
                for (int i = 0; i < 4; ++i) P[((g * 3 + slot) * 4 + i) * 64 + lane] = acc[4 * g + i];
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mine = wq == 0 ? acc[i] : wq == 1 ? acc[4 + i] : wq == 2 ? acc[8 + i] : acc[12 + i];
            own[i] = mine + ((P[((wq * 3 + 0) * 4 + i) * 64 + lane] + P[((wq * 3 + 1) * 4 + i) * 64 + lane]) + P[((wq * 3 + 2) * 4 + i) * 64 + lane]);
        }
        coop_epilogue_n<TRAIN, 4>(own, bz4, 8 * wq, c, h, 0xFFFFu, 1.0f, B4, G4);
    }
    lds_barrier();
    f32x16 qv = zero16();
    if (wq == 0) {
        qv = mfma_w<4>(w5r, c, h, B4, 0);
        bias_act<false>(qv, net_b(n, 4), 0, A, h);
    }
    return qv;
}

// 256 candidate rows -> ids of the selected ones in List[0..count), count returned to every thread
__device__ __forceinline__ int coop_compact(int* __restrict__ list, bool sel, int row) {
    int* wcount = list + 256;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
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

// what wavefront 0 does with a tile's Q values: optional Q output, first maximal index (torch.argmax), epsilon draw, action
__device__ __forceinline__ void act_tile_finish(const QNetArgs& a, const f32x16& qv, int rowc, int h) {
    const int A = a.net.n_actions;
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

// ---- masked action selection, cooperative (pulse_qnet_act with seat_idx) ------------------------------------
// WIN candidate rows per workgroup: 128 -- two workgroups fit a CU's LDS and overlap each other's barrier phases; a sixth
// of the candidates being the learner's, a window is one tile (256 candidates were a full tile plus a third of one).
// NK1: steps of 8 inputs in layer 1 (5 for the 40-column observation: 12 registers less than the general 8).
// As a device function of a workgroup's FIRST 256 threads (all of them: it holds workgroup barriers), window `win`, `lds` =
// kActLdsBytes of LDS: qnet_act4_kernel is this and nothing else; poker_step.hip calls it in front of the tables' step.
template <bool VEC, int WIN, int NK1>
__device__ __forceinline__ void act_window(const QNetArgs& a, float* __restrict__ lds, int win) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int row = win * WIN + threadIdx.x;
    const bool cand = threadIdx.x < WIN && row < a.n_rows;
    const bool sel = cand && a.seat_idx[row] == a.q_seat;
    if (a.row_mask_out && cand)                      // the trainer's `q_mask & ~terminated` (trainGPU.py:85) rides along
        a.row_mask_out[row] = (sel && !(a.terminated && a.terminated[row])) ? 1 : 0;
    QSTAMP(0);
    const int K1 = a.net.state_dim, K1r = (K1 + 7) & ~7;
    float w1r[NK1][4];
    if (!a.asel_counts) load_layer<VEC, NK1>(w1r, a.net, 0, 32 * wv + c, h, 0, K1r);           // in flight during the compaction (not when this launch only lists)
    // The rows the NEXT training launch will use (row_mask_out & seat status ACTIVE / ALLIN, Player.py:258-261) are known
    // here already -- it trains on this observation: their lists are written now and the select launch is not needed.
    int* const list_w = reinterpret_cast<int*>(lds + ActLds::List);
    int* tcount = list_w + 260;
    bool tsel = false; unsigned long long tm = 0ull;
    if (a.tsel_counts) {
        tsel = sel && !(a.terminated && a.terminated[row]);
        if (tsel) { const float status = a.states[(size_t)row * a.row_stride + 12]; tsel = status == 0.0f || status == 2.0f; }
        tm = __ballot(tsel);
        if (lane == 0) tcount[wv] = __popcll(tm);
    }
    const int count = coop_compact(list_w, sel, row);             // (its barriers publish tcount too)
    if (a.tsel_counts) {
        int tbase = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) tbase += i < wv ? tcount[i] : 0;
        if (tsel) a.tsel_rows[(size_t)win * WIN + tbase + __popcll(tm & ((1ull << lane) - 1ull))] = row;
        if (threadIdx.x == 0) a.tsel_counts[win] = (tcount[0] + tcount[1]) + (tcount[2] + tcount[3]);
    }
    QSTAMP(1);
    const int* list = list_w;
    if (a.asel_counts) {                                                      // (WIN <= 256 threads: one store per listed row)
        if ((int)threadIdx.x < count) a.asel_rows[(size_t)win * WIN + threadIdx.x] = list[threadIdx.x];
        if (threadIdx.x == 0) a.asel_counts[win] = count;
        return;
    }
    for (int t0 = 0; t0 < count; t0 += 32) {
        const int rowc = t0 + c < count ? list[t0 + c] : -1;
        lds_barrier();                                                        // previous tile's readers are done
        coop_load_rows<VEC>(lds + ActLds::R0, a.states, a.row_stride, K1, rowc, wv, c, h);
        // (a layer writes the region its input's input was read from: that reading ended behind the barrier between the layers)
        const f32x16 qv = group_forward<false, NK1>(w1r, a.net, lds + ActLds::R0, lds + ActLds::R1, lds + ActLds::R0, lds + ActLds::R1, lds + ActLds::R0,
                                                    nullptr, nullptr, nullptr, nullptr, lds + ActLds::P, wv, c, h, 0, 0, 0, 0, 1.0f);
        if (wv == 0) act_tile_finish(a, qv, rowc, h);
        QSTAMP(7);
    }
    QSTAMP(8);
}

}  // namespace pulse_qnet
