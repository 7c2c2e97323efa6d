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
#include <cstdlib>

#include "pulse_internal.h"
#include "pulse_internal.h"

// In-kernel timeline of the training kernel (diagnostic build only, -DPULSE_STAMPS=1 -> libpulse_hip_stamps.so,
// tools/qnet_stamp_timeline.py): lane 0 of wavefront 0 stores the clock at phase boundaries of its first tile.
#ifndef PULSE_STAMPS
#define PULSE_STAMPS 0
#endif
#if PULSE_STAMPS
__device__ unsigned long long* g_qstamp_buf = nullptr;
#define QSTAMP(i) do { if (threadIdx.x == 0 && g_qstamp_buf) { __builtin_amdgcn_sched_barrier(0); \
    g_qstamp_buf[(size_t)blockIdx.x * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define QSTAMP(i) do { } while (0)
#endif

#include "qnet_device.h"
#include "qnet_rows16.h"

namespace {

using namespace pulse_qnet;

template <bool VEC, int WIN, int NK1>
__global__ __launch_bounds__(256, 3) void qnet_act4_kernel(const QNetArgs a) {
    extern __shared__ float lds[];
    act_window<VEC, WIN, NK1>(a, lds, (int)blockIdx.x);
}

// The second launch of the two-launch form (large batches): the learner's rows of the whole batch, listed per window by the
// window launch (a.asel_rows / a.asel_counts), are positions [0, T); persistent workgroups take the FULL 32-row tiles of
// [0, T) in turn.  A window of 128 candidates holds ~21 of the learner's rows, i.e. a window launch runs its tiles two thirds
// full: at 2,000,000 tables 15,625 tiles against 10,400 here (522 -> ~370 us), for one more launch (which a small batch,
// one round of tiles either way, would only pay for).  Per-row results do not depend on the tile a row rides in.
template <bool VEC, int WIN, int NK1>
__global__ __launch_bounds__(256, 3) void qnet_act_rows_kernel(const QNetArgs a) {
    extern __shared__ float lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int n_windows = (a.n_rows + WIN - 1) / WIN, per = (n_windows + 255) / 256;
    int* chunk = reinterpret_cast<int*>(lds + ActLds::List);     // [256] first position of thread t's windows; [257..260] wavefront totals
    int T;
    {
        int mine = 0;
        for (int j = 0; j < per; ++j) { const int w = threadIdx.x * per + j; mine += w < n_windows ? a.asel_counts[w] : 0; }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); incl += lane >= off ? o : 0; }
        int* wtot = chunk + 257;
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) base += i < wv ? wtot[i] : 0;
        chunk[threadIdx.x] = base + incl - mine;
        T = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
        __syncthreads();
    }
    const int n_tiles = (T + 31) / 32;
    if ((int)blockIdx.x >= n_tiles) return;
    const int K1 = a.net.state_dim, K1r = (K1 + 7) & ~7;
    float w1r[NK1][4];
    load_layer<VEC, NK1>(w1r, a.net, 0, 32 * wv + c, h, 0, K1r);
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        int rowc = -1;
        const int p = 32 * ti + c;
        if (p < T) {                                             // its window by bisection of the threads' first positions, then along that thread's windows
            int t = 0;
#pragma unroll
            for (int s = 128; s >= 1; s >>= 1) t += (chunk[t + s] <= p) ? s : 0;
            int w = t * per, acc = chunk[t], cnt = a.asel_counts[w];
            while (p >= acc + cnt) { acc += cnt; ++w; cnt = a.asel_counts[w]; }
            rowc = a.asel_rows[(size_t)w * WIN + (p - acc)];
        }
        lds_barrier();                                                        // previous tile's readers are done
        coop_load_rows<VEC>(lds + ActLds::R0, a.states, a.row_stride, K1, rowc, wv, c, h);
        const f32x16 qv = group_forward<false, NK1>(w1r, a.net, lds + ActLds::R0, lds + ActLds::R1, lds + ActLds::R0, lds + ActLds::R1, lds + ActLds::R0,
                                                    nullptr, nullptr, nullptr, nullptr, lds + ActLds::P, wv, c, h, 0, 0, 0, 0, 1.0f);
        if (wv == 0) act_tile_finish(a, qv, rowc, h);
    }
}

// ================================================================ training step (Player.py:255-294)
// Launch 1 (qnet_train_kernel) does, for the rows that pass the reference's filters, what train_step does between
// its masks and `loss.backward()`: TD target from the target network, forward in train mode (dropout after the 2nd
// and 3rd GELU), d(loss)/d(parameters) -- UNNORMALISED sums (the 1 / #valid-rows of MSELoss, the norm clipping and
// AdamW follow in launch 2, which knows the global row count).  Per tile of <= 32 rows, on one CU:
//   forwards  the target network on s' (eval) and the network on s (train mode) through the layers together; every hidden
//             layer leaves a_l and g_l = gelu'(z_l) * dropout scale in LDS;  target = r + gamma max_a' Q_target(s', a'),
//             delta_5 = 2 (q[action] - target) on the action's row of the output tile;
//   backward  per layer, dealt to the 4 wavefronts: the 32x32 blocks of dW_l = delta_l . a_{l-1}^T (both operands read
//             out of LDS with the row index as the MFMA k, 16 MFMAs per block per tile; db_l falls out of the same
//             reads) and the tiles of delta_{l-1} = (W_l^T . delta_l) * g_{l-1} (A operand = W_l read down its
//             columns, coalesced).
// Every wavefront owns the same 8-10 blocks of dW for the whole launch and accumulates them, tile after tile, in its
// workgroup's private slice of `partials` (plain read-modify-write, L2-resident; the workgroups are persistent: at
// most `max_blocks` of them take the tiles of the batch's row lists in turn) -- no atomics: float atomics from 8 XCDs
// onto 32 K shared addresses were 80 % of this kernel's time in the first version.
// Launch 2 (qnet_grad_reduce_kernel) sums the slices into the flat gradient and its squared norm and, on one GPU, applies
// mean / clip_grad_norm_ / AdamW / target sync to the parameters it holds the sums of (qnet_adamw_kernel: the same as a
// launch of its own, for data-parallel training where an all-reduce comes between the two).
constexpr int kMeetUsed = 400;                 // meet[400]: how many of the training launch's workgroups wrote a slice (workgroups 0 .. that - 1)
struct TrainArgs {
    FlatNet net, tgt;
    float* partials;                          // [gridDim.x][kSlicePitch]: gradient blocks, biases, then {rows, sum td^2, -, used}
    float* scal;                              // scal[0] = squared gradient norm of the reduce launch: cleared here for it
    int n_params;
    const float* states; long long stride;
    const int64_t* actions; const float* rewards;
    const float* next_states; long long next_stride;
    const uint8_t* dones;
    const int32_t* sel_rows; const int32_t* sel_counts;   // the row lists: sel_rows[(w << win_shift) + i], i < sel_counts[w]
    int win_shift;                                        // 8: the select launch's windows; 7: the act launch's
    const uint8_t* row_mask; uint8_t* terminated; int book; // book: the per-candidate bookkeeping is done here (lists from act)
    unsigned* meet;
    int n_rows;
    uint64_t seed, step, table_id0;
    float gamma, drop_p;
};

// Launch 0 (qnet_select_kernel; not needed after pulse_qnet_act_select), one workgroup per window of 256 candidate rows: everything train_step and the trainer do
// per CANDIDATE row -- the reference's filters (row_mask, seat status ACTIVE / ALLIN, Player.py:258-261) compacted into
// the window's list of selected rows, `terminated |= dones` (trainGPU.py:86), the window's reward sum over the row_mask
// rows (trainGPU.py:96).  The training launch then deals the selected rows of the WHOLE batch evenly to its workgroups:
// with per-window tiles a window holding 33 selected rows (7 % of them at a tenth selected) cost its workgroup two tiles
// and the launch twice the time of the other 93 %.
struct SelectArgs {
    const float* states; long long stride; const float* rewards; const uint8_t* dones; const uint8_t* row_mask;
    uint8_t* terminated; int n_rows;
    int32_t* sel_rows; int32_t* sel_counts; float* win_reward;
};

__global__ __launch_bounds__(256) void qnet_select_kernel(const SelectArgs a) {
    __shared__ int wcount[4];
    __shared__ float wrew[4];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = blockIdx.x * 256 + threadIdx.x;
    bool sel = row < a.n_rows && (a.row_mask == nullptr || a.row_mask[row] != 0);
    float rew = sel ? a.rewards[row] : 0.0f;                     // episode reward: rows of row_mask, before the status filter
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rew += __shfl_xor(rew, off);
    if (a.terminated && row < a.n_rows && a.dones[row]) a.terminated[row] = 1;
    if (sel) {                                                   // seat status ACTIVE or ALLIN, Player.py:261
        const float status = a.states[(size_t)row * a.stride + 12];
        sel = status == 0.0f || status == 2.0f;
    }
    const unsigned long long m = __ballot(sel);
    if (lane == 0) { wcount[wv] = __popcll(m); wrew[wv] = rew; }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += i < wv ? wcount[i] : 0;
    if (sel) a.sel_rows[(size_t)blockIdx.x * 256 + base + __popcll(m & ((1ull << lane) - 1ull))] = row;
    if (threadIdx.x == 0) {
        a.sel_counts[blockIdx.x] = (wcount[0] + wcount[1]) + (wcount[2] + wcount[3]);
        a.win_reward[blockIdx.x] = (wrew[0] + wrew[1]) + (wrew[2] + wrew[3]);
    }
}

// A workgroup's gradient slice is private scratch, so its layout is the accumulators' own: 35 blocks of 32x32 (layer 1:
// 4x2, layer 2: 4x4, layer 3: 2x4, layer 4: 1x2, layer 5: 1x1 -- padded rows / columns included), each stored as
// [lane][16 registers], then the five bias vectors, then 4 statistics.  A wavefront then writes a block with four
// 16-byte stores per lane instead of sixteen 4-byte ones (global stores are issue-bound: the dword form made the
// weight-gradient blocks 4x slower than their MFMAs); qnet_grad_reduce_kernel maps parameters to this layout.
constexpr int kSliceBlk1 = 0, kSliceBlk2 = 8, kSliceBlk3 = 24, kSliceBlk4 = 32, kSliceBlk5 = 34, kSliceBlocks = 35;
constexpr int kSliceBias = kSliceBlocks * 1024;                 // b1 @+0, b2 @+128, b3 @+256, b4 @+320, b5 @+352 (32 slots)
constexpr int kSliceStats = kSliceBias + 384, kSlicePitch = kSliceStats + 4;

// block `blk` (= dW rows [32 ot, +32) x columns [32 it, +32) of its layer) += delta . a^T for this tile (`first`: nothing
// accumulated yet); delta in D, a_{l-1} in Ap; bsum: this tile's db rows of tile ot
__device__ __forceinline__ void dw_accum(const float* __restrict__ D, const float* __restrict__ Ap, float* __restrict__ slice, int blk,
                                         int ot, int it, int c, int h, bool first, float* bsum) {
    // registers 4q .. 4q+3 of all 64 lanes form one contiguous KB of the slice: a store instruction writes whole lines.  The
    // block's old values are asked for FIRST (not after the MFMAs, where every call waited out their round trip: with ~20 tiles
    // per workgroup at 2,000,000 tables the slices live in the Infinity Cache, not in L2)
    float4* dst = reinterpret_cast<float4*>(slice + (size_t)blk * 1024) + (c + 32 * h);
    float4 old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { old[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); if (!first) old[q] = dst[64 * q]; }
    float ad[16], ap[16]; float bs = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) { ad[s2] = D[(32 * ot + c) * kLd + 2 * s2 + h]; ap[s2] = Ap[(32 * it + c) * kLd + 2 * s2 + h]; }
    __builtin_amdgcn_sched_barrier(0);                            // (all 32 LDS reads ahead of the MFMAs, as in mfma_w)
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) bs += ad[s2];
    if (bsum) *bsum += bs + __shfl_xor(bs, 32);
    f32x16 acc = zero16();
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[s2], ap[s2], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        dst[64 * q] = make_float4(acc[4 * q] + old[q].x, acc[4 * q + 1] + old[q].y, acc[4 * q + 2] + old[q].z, acc[4 * q + 3] + old[q].w);
}

// flat parameter index (order w1,b1,...,w5,b5) of slice element j, or -1 for a padding element
__device__ __forceinline__ int slice_param(int j, int K1, int A) {
    const int n_out[5] = {128, 128, 64, 32, A}, n_in[5] = {K1, 128, 128, 64, 32};
    const int blk0[6] = {kSliceBlk1, kSliceBlk2, kSliceBlk3, kSliceBlk4, kSliceBlk5, kSliceBlocks}, its[5] = {2, 4, 4, 2, 1};
    const int bias0[6] = {0, 128, 256, 320, 352, 384};
    int base[5], bbase[5], acc = 0;
#pragma unroll
    for (int l = 0; l < 5; ++l) { base[l] = acc; acc += n_out[l] * n_in[l]; bbase[l] = acc; acc += n_out[l]; }
    if (j >= kSliceBias) {
        const int u = j - kSliceBias;
#pragma unroll
        for (int l = 0; l < 5; ++l)
            if (u >= bias0[l] && u < bias0[l + 1]) return (u - bias0[l]) < n_out[l] ? bbase[l] + (u - bias0[l]) : -1;
        return -1;
    }
    const int blk = j >> 10, lane = (j >> 2) & 63, r = 4 * ((j >> 8) & 3) + (j & 3), c = lane & 31, h = lane >> 5;   // [block][q][lane][4]
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        if (blk >= blk0[l] && blk < blk0[l + 1]) {
            const int b = blk - blk0[l], ot = b / its[l], it = b - ot * its[l];
            const int o = 32 * ot + rho(r) + 4 * h, in = 32 * it + c;
            return (o < n_out[l] && in < n_in[l]) ? base[l] + o * n_in[l] + in : -1;
        }
    }
    return -1;
}

// tile `it` of delta_{l-1} = (W^T . delta_l) * g_{l-1} -> Dn[32 it ..]; W is n_out x n_in, delta_l = units [0, KU) of D.
// back_load: the KU / 2 weights (one per MFMA, down a column of W: coalesced), issued a phase ahead by the caller.
template <int KU>
__device__ __forceinline__ void back_load(float (&wa)[KU / 2], const float* __restrict__ w, int n_out, int n_in, int it, int c, int h) {
#pragma unroll
    for (int i = 0; i < KU / 2; ++i) {
        const int k = 2 * i + h;
        wa[i] = k < n_out ? w[(size_t)k * n_in + 32 * it + c] : 0.0f;
    }
}
template <int KU>
__device__ __forceinline__ void back_mul(const float (&wa)[KU / 2], int it, const float* __restrict__ D, const float* __restrict__ G,
                                         float* __restrict__ Dn, int c, int h) {
    float dv[KU / 2], gv[16];
#pragma unroll
    for (int i = 0; i < KU / 2; ++i) dv[i] = D[(2 * i + h) * kLd + c];
#pragma unroll
    for (int r = 0; r < 16; ++r) gv[r] = G[(32 * it + rho(r) + 4 * h) * kLd + c];
    __builtin_amdgcn_sched_barrier(0);                            // (LDS reads ahead of the MFMAs)
    f32x16 acc = zero16();
#pragma unroll
    for (int i = 0; i < KU / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i], dv[i], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Dn[(32 * it + rho(r) + 4 * h) * kLd + c] = acc[r] * gv[r];
}

template <bool VEC, int NK1>
__global__ __launch_bounds__(256) void qnet_train_kernel(const TrainArgs a) {
    extern __shared__ float lds[];
    const FlatNet& n = a.net;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c0 = lane & 31, h0 = lane >> 5;
    const int K1 = n.state_dim, A = n.n_actions;
    float* Xs = lds + CoopLds::Xs; float* A1 = lds + CoopLds::A1; float* A2 = lds + CoopLds::A2; float* A3 = lds + CoopLds::A3;
    float* A4 = lds + CoopLds::A4;
    float* G1 = lds + CoopLds::G1; float* G2 = lds + CoopLds::G2; float* G3 = lds + CoopLds::G3; float* G4 = lds + CoopLds::G4;
    float* Da = lds + CoopLds::Da; float* Db = lds + CoopLds::Db;
    const uint32_t thr = (uint32_t)(a.drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - a.drop_p);

    // this wavefront's rows of db, live across every tile of the launch (its blocks of dW accumulate in the slice)
    float b5 = 0.0f, b4 = 0.0f, b3 = 0.0f, b2 = 0.0f, b1 = 0.0f;
    float rows_sum = 0.0f, sq_sum = 0.0f;                        // wavefront 0, lane-replicated after the reductions
    bool used = false;
    float* part = a.partials + (size_t)blockIdx.x * kSlicePitch;

    // scal[0] is read by every workgroup of the previous step's AdamW launch and accumulated by this step's reduce
    // launch: this kernel sits between the two on the stream, so its first thread clears it
    if (blockIdx.x == 0 && threadIdx.x < 8) a.meet[threadIdx.x] = 0u;          // the fused reduce launch's arrival counters
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[0] = 0.0f;
    QSTAMP(0);
    // The selected rows of the whole batch, in window order, are positions [0, T); every workgroup computes the same
    // exclusive sums of the windows' counts (thread t: windows [t per, (t + 1) per)) and takes the tiles ti = blockIdx.x,
    // + gridDim.x, ... of an even split of [0, T) into n_tiles <= 32-row pieces.
    float reward_sum = 0.0f;
    if (a.book) {                                                // what the select launch does per candidate row, when act made the lists
        for (int win = blockIdx.x; win * 256 < a.n_rows; win += gridDim.x) {
            const int row = win * 256 + threadIdx.x;
            const bool cand = row < a.n_rows && (a.row_mask == nullptr || a.row_mask[row] != 0);
            float rew = cand ? a.rewards[row] : 0.0f;            // episode reward: rows of row_mask, before the status filter (trainGPU.py:96)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) rew += __shfl_xor(rew, off);
            reward_sum += rew;
            if (a.terminated && row < a.n_rows && a.dones[row]) a.terminated[row] = 1;      // trainGPU.py:86
        }
    }
    const int W = 1 << a.win_shift;
    const int n_windows = (a.n_rows + W - 1) >> a.win_shift, per = (n_windows + 255) / 256;
    int* chunk = reinterpret_cast<int*>(lds + CoopLds::List);    // [256] first position of thread t's windows, [256] = T
    int T;
    {
        int mine = 0;
        for (int j = 0; j < per; ++j) { const int w = threadIdx.x * per + j; mine += w < n_windows ? a.sel_counts[w] : 0; }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); incl += lane >= off ? o : 0; }
        int* wtot = chunk + 257;
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) base += i < wv ? wtot[i] : 0;
        chunk[threadIdx.x] = base + incl - mine;
        T = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
        __syncthreads();
    }
    const int G = (int)gridDim.x;
    // full tiles: fewer slices to reduce than with T rows spread over all the workgroups (a tile costs the same with 25 rows
    // as with 32), and every workgroup at most one more tile than any other
    const int n_tiles = (T + 31) / 32;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.meet[kMeetUsed] = (unsigned)min(n_tiles, G);      // workgroups 0 .. n_tiles - 1 hold a gradient slice (the reduce launch sums exactly those)
    for (int ti = blockIdx.x; ti < n_tiles; ti += G) {
        {
            const bool first = !used;
            used = true;
            // the addresses below depend only on the wavefront and the lane: without this opaque zero the compiler
            // hoists a few hundred of them out of the tile loop and spills them
            int opaque = 0;
            asm volatile("" : "+s"(opaque));
            float* part_t = part + opaque;
            int c = c0, h = h0;
            asm volatile("" : "+v"(c), "+v"(h));
            // column c = position lo + c of the batch's selected rows: its window by bisection of the threads' first
            // positions (the last t with chunk[t] <= p has a non-empty range holding p), then along that thread's windows
            const int lo = (int)((long long)ti * T / n_tiles), hi = (int)((long long)(ti + 1) * T / n_tiles);
            int rowc = -1;
            if (lo + c < hi) {
                const int p = lo + c;
                int t = 0;
#pragma unroll
                for (int s = 128; s >= 1; s >>= 1) t += (chunk[t + s] <= p) ? s : 0;
                int w = t * per, acc = chunk[t], cnt = a.sel_counts[w];
                while (p >= acc + cnt) { acc += cnt; ++w; cnt = a.sel_counts[w]; }
                rowc = a.sel_rows[((size_t)w << a.win_shift) + (p - acc)];
            }
            const bool live = rowc >= 0;
            const int rw = max(rowc, 0);
            const uint64_t gid = a.table_id0 + (uint64_t)rw;
            // both forwards together: target r + gamma * max_a' Q_target(s', a') * (1 - done) (:275-277), network in train mode
            lds_barrier();
            QSTAMP(1);
            float w1t[NK1][4], w1c[NK1][4];
            load_layer<VEC, NK1>(w1t, a.tgt, 0, 32 * wv + c, h, 0, (K1 + 7) & ~7);
            load_layer<VEC, NK1>(w1c, n, 0, 32 * wv + c, h, 0, (K1 + 7) & ~7);
            coop_load_rows<VEC>(lds + CoopLds::Db, a.next_states, a.next_stride, K1, rowc, wv, c, h);
            coop_load_rows<VEC>(lds + CoopLds::Xs, a.states, a.stride, K1, rowc, wv, c, h);
            float wb5[16];
            if (wv == 1) back_load<32>(wb5, net_w(n, 4), A, 32, 0, c, h);
            // the row's transition, ahead of the forwards that need it last
            const float row_done = (live && a.dones[rw]) ? 1.0f : 0.0f, row_reward = live ? a.rewards[rw] : 0.0f;
            const int act = live ? (int)a.actions[rw] : -1;
            {
                f32x16 qn, qv;
                coop_forward_pair<VEC, NK1>(w1t, w1c, a.tgt, n, lds, wv, c, h, a.seed, gid, a.step, thr, scale, qn, qv);
                QSTAMP(2);
                if (wv == 0) {                                                // delta_5 and the loss terms (:270-279)
                    float best = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (rho(r) + 4 * h < A) best = fmaxf(best, qn[r]);
                    best = fmaxf(best, __shfl_xor(best, 32));
                    const float target = row_reward + a.gamma * best * (1.0f - row_done);
                    float qa = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) qa += (rho(r) + 4 * h == act) ? qv[r] : 0.0f;
                    qa += __shfl_xor(qa, 32);
                    const float td = live ? qa - target : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) Da[(rho(r) + 4 * h) * kLd + c] = (rho(r) + 4 * h == act) ? 2.0f * td : 0.0f;
                    float sq = (h == 0) ? td * td : 0.0f, cnt = (h == 0 && live) ? 1.0f : 0.0f;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
                    rows_sum += cnt; sq_sum += sq;
                }
            }
            lds_barrier();
            QSTAMP(3);
            // The backward phases talk through LDS only (a wavefront's slice blocks are its own), so their barriers do not wait
            // for global memory either, and every phase issues the weights of the NEXT phase's delta product first.
            // layer 5 (delta_5 in Da): dW5 | delta_4 -> Db
            float wb4[16];
            if (wv >= 2) back_load<32>(wb4, net_w(n, 3), 32, 64, wv - 2, c, h);
            if (wv == 0) dw_accum(Da, A4, part_t, kSliceBlk5, 0, 0, c, h, first, &b5);
            if (wv == 1) back_mul<32>(wb5, 0, Da, G4, Db, c, h);
            lds_barrier();
            QSTAMP(4);
            // layer 4 (delta_4 in Db): dW4 blocks on wavefronts 0, 1 | delta_3 tiles on 2, 3 -> Da
            float wb3[32];
            back_load<64>(wb3, net_w(n, 2), 64, 128, wv, c, h);
            if (wv == 0) dw_accum(Db, A3, part_t, kSliceBlk4 + 0, 0, 0, c, h, first, &b4);
            if (wv == 1) dw_accum(Db, A3, part_t, kSliceBlk4 + 1, 0, 1, c, h, first, nullptr);
            if (wv >= 2) back_mul<32>(wb4, wv - 2, Db, G3, Da, c, h);
            lds_barrier();
            QSTAMP(5);
            // layer 3 (delta_3 in Da): 8 dW blocks, column tile wv of both row tiles | delta_2 tile wv -> Db
            float wb2[64];
            back_load<128>(wb2, net_w(n, 1), 128, 128, wv, c, h);
            dw_accum(Da, A2, part_t, kSliceBlk3 + wv, 0, wv, c, h, first, wv == 0 ? &b3 : nullptr);
            dw_accum(Da, A2, part_t, kSliceBlk3 + 4 + wv, 1, wv, c, h, first, wv == 1 ? &b3 : nullptr);
            back_mul<64>(wb3, wv, Da, G2, Db, c, h);
            lds_barrier();
            QSTAMP(6);
            // layer 2 (delta_2 in Db): 16 dW blocks, column tile wv of each row tile | delta_1 tile wv -> Da
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) dw_accum(Db, A1, part_t, kSliceBlk2 + 4 * ot + wv, ot, wv, c, h, first, ot == wv ? &b2 : nullptr);
            QSTAMP(10);
            back_mul<128>(wb2, wv, Db, G1, Da, c, h);
            QSTAMP(11);
            lds_barrier();
            QSTAMP(7);
            // layer 1 (delta_1 in Da): row tile wv x the two column tiles of the input
            dw_accum(Da, Xs, part_t, kSliceBlk1 + 2 * wv, wv, 0, c, h, first, &b1);
            if (K1 > 32) dw_accum(Da, Xs, part_t, kSliceBlk1 + 2 * wv + 1, wv, 1, c, h, first, nullptr);
        }
    }

    QSTAMP(8);
    if (used && K1 <= 32) {                                       // (no tile for this workgroup: the reduce launch skips its slice)                                        // the second column tile of layer 1 was never touched
        for (int i = threadIdx.x; i < 4 * 1024; i += 256) part[(size_t)(kSliceBlk1 + 2 * (i >> 10) + 1) * 1024 + (i & 1023)] = 0.0f;
    }
    // db rows: every bias is written by exactly one wavefront
    if (used && h0 == 0) {
        if (wv == 0) { part[kSliceBias + 352 + c0] = b5; part[kSliceBias + 320 + c0] = b4; }
        if (wv < 2) part[kSliceBias + 256 + 32 * wv + c0] = b3;
        part[kSliceBias + 128 + 32 * wv + c0] = b2;
        part[kSliceBias + 32 * wv + c0] = b1;
    }
    float* wave_reward = lds + CoopLds::Tgt;                      // (32 spare words)
    lds_barrier();
    if (lane == 0) wave_reward[wv] = reward_sum;
    lds_barrier();
    if (wv == 0 && lane == 0) {
        float* ps = part + kSliceStats;
        ps[0] = rows_sum; ps[1] = sq_sum; ps[2] = (wave_reward[0] + wave_reward[1]) + (wave_reward[2] + wave_reward[3]);
        ps[3] = used ? 1.0f : 0.0f;
    }
    QSTAMP(9);
}

// ---- the same launch with EIGHT wavefronts per workgroup --------------------------------------------------------------
// One wavefront per SIMD issues roughly one instruction per 8 cycles, and two thirds of the four-wavefront kernel above is
// instructions that are neither MFMA nor GELU.  Here every SIMD has two wavefronts to issue from: in the
// forwards wavefronts 0-3 take the target network through its layers and 4-7 the trained network (each as the act kernel's
// four do, the epilogues of layers 3 and 4 split between them); the backward phases deal their 16 + 4 products per layer to
// all eight (delta tiles on 0-3, the weight-gradient blocks on 4-7, equal MFMA counts).  Same tiles, same LDS image, same
// slice layout; a block of the slice is still owned by one wavefront for the whole launch.
template <bool VEC, int NK1>
__global__ __launch_bounds__(512) void qnet_train8_kernel(const TrainArgs a) {
    extern __shared__ float lds[];
    const FlatNet& n = a.net;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c0 = lane & 31, h0 = lane >> 5;
    const int K1 = n.state_dim, A = n.n_actions;
    float* Xs = lds + CoopLds::Xs; float* A1 = lds + CoopLds::A1; float* A2 = lds + CoopLds::A2; float* A3 = lds + CoopLds::A3;
    float* A4 = lds + CoopLds::A4; float* P = lds + CoopLds::P;
    float* G1 = lds + CoopLds::G1; float* G2 = lds + CoopLds::G2; float* G3 = lds + CoopLds::G3; float* G4 = lds + CoopLds::G4;
    float* Da = lds + CoopLds::Da; float* Db = lds + CoopLds::Db; float* Tg = lds + CoopLds::Tgt;
    const uint32_t thr = (uint32_t)(a.drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - a.drop_p);

    float b5 = 0.0f, b4 = 0.0f, b3 = 0.0f, b2 = 0.0f, b1 = 0.0f;  // this wavefront's rows of db (see the stores at the end)
    float rows_sum = 0.0f, sq_sum = 0.0f;                        // wavefront 4
    bool used = false;
    float* part = a.partials + (size_t)blockIdx.x * kSlicePitch;
    if (blockIdx.x == 0 && threadIdx.x < 8) a.meet[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[0] = 0.0f;
    QSTAMP(0);
    float reward_sum = 0.0f;
    if (a.book) {
        for (int win = blockIdx.x; win * 512 < a.n_rows; win += gridDim.x) {
            const int row = win * 512 + threadIdx.x;
            const bool cand = row < a.n_rows && (a.row_mask == nullptr || a.row_mask[row] != 0);
            float rew = cand ? a.rewards[row] : 0.0f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) rew += __shfl_xor(rew, off);
            reward_sum += rew;
            if (a.terminated && row < a.n_rows && a.dones[row]) a.terminated[row] = 1;
        }
    }
    const int W = 1 << a.win_shift;
    // (512 first positions, one per thread: a column's walk along its thread's windows below is half as long as with 256 -- at
    // 2,000,000 tables 31 windows instead of 62, a dependent load each)
    const int n_windows = (a.n_rows + W - 1) >> a.win_shift, per = (n_windows + 511) / 512;
    int* chunk = reinterpret_cast<int*>(lds + CoopLds::List);
    int T;
    {
        int mine = 0;
        for (int j = 0; j < per; ++j) { const int w = threadIdx.x * per + j; mine += w < n_windows ? a.sel_counts[w] : 0; }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); incl += lane >= off ? o : 0; }
        int* wtot = chunk + 512;
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int base = 0; T = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int n = wtot[i]; base += i < wv ? n : 0; T += n; }
        chunk[threadIdx.x] = base + incl - mine;
        __syncthreads();
    }
    const int G = (int)gridDim.x;
    // full tiles: fewer slices to reduce than with T rows spread over all the workgroups (a tile costs the same with 25 rows
    // as with 32), and every workgroup at most one more tile than any other
    const int n_tiles = (T + 31) / 32;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.meet[kMeetUsed] = (unsigned)min(n_tiles, G);      // workgroups 0 .. n_tiles - 1 hold a gradient slice (the reduce launch sums exactly those)
    // column c of tile ti = position lo + c of the batch's listed rows: its thread's first position by bisection, then along that
    // thread's windows (a dependent load each)
    auto tile_row = [&](int ti, int c) -> int {
        const int lo = (int)((long long)ti * T / n_tiles), hi = (int)((long long)(ti + 1) * T / n_tiles);
        int rowc = -1;
        if (lo + c < hi) {
            const int p = lo + c;
            int t = 0;
#pragma unroll
            for (int s = 256; s >= 1; s >>= 1) t += (chunk[t + s] <= p) ? s : 0;
            int w = t * per, acc = chunk[t], cnt = a.sel_counts[w];
            while (p >= acc + cnt) { acc += cnt; ++w; cnt = a.sel_counts[w]; }
            rowc = a.sel_rows[((size_t)w << a.win_shift) + (p - acc)];
        }
        return rowc;
    };
    // The NEXT tile's rows are looked up by wavefront 7 while it has nothing to do (the layer-5 backward phase runs on two
    // wavefronts) and their observation rows are asked for by everybody in the layer-4 phase: with ~20 tiles per workgroup the
    // walk and the gather's round trip were 5 % of a tile, in front of its first layer.
    int* const next_rows = reinterpret_cast<int*>(Tg);           // (free between delta_5 and the next tile's max Q_target)
    int rowc_next = blockIdx.x < n_tiles ? tile_row((int)blockIdx.x, c0) : -1;
    float xpre[8];
    bool have_pre = false;
    for (int ti = blockIdx.x; ti < n_tiles; ti += G) {
        const bool first = !used;
        used = true;
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        float* part_t = part + opaque;
        int c = c0, h = h0;
        asm volatile("" : "+v"(c), "+v"(h));
        const int wq = wv & 3, grp = wv >> 2;
        const int rowc = rowc_next;
        const bool live = rowc >= 0;
        const int rw = max(rowc, 0);
        const uint64_t gid = a.table_id0 + (uint64_t)rw;
        lds_barrier();                                            // the previous tile's readers are done
        QSTAMP(1);
        // group 0: the target network on s' (x' and a'_2, a'_4 in Db, a'_1 and a'_3 in Da); group 1: the network on s
        float w1r[NK1][4];
        load_layer<VEC, NK1>(w1r, grp ? n : a.tgt, 0, 32 * wq + c, h, 0, (K1 + 7) & ~7);
        if (!have_pre) {
            if (grp) coop_fetch_rows<VEC>(xpre, a.states, a.stride, K1, rowc, wq, h);
            else coop_fetch_rows<VEC>(xpre, a.next_states, a.next_stride, K1, rowc, wq, h);
        }
        coop_store_rows(grp ? Xs : Db, xpre, wq, c, h);
        const float row_done = (live && a.dones[rw]) ? 1.0f : 0.0f, row_reward = live ? a.rewards[rw] : 0.0f;
        const int act = live ? (int)a.actions[rw] : -1;
        f32x16 qv;
        if (grp) qv = group_forward<true, NK1>(w1r, n, Xs, A1, A2, A3, A4, G1, G2, G3, G4, P + 3 * 16 * 64, wq, c, h, a.seed, gid, a.step, thr, scale);
        else qv = group_forward<false, NK1>(w1r, a.tgt, Db, Da, Db, Da, Db, nullptr, nullptr, nullptr, nullptr, P, wq, c, h, 0, 0, 0, 0, 1.0f);
        QSTAMP(2);
        if (wv == 0) {                                            // max_a' Q_target(s', a') per row -> Tg
            float best = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) if (rho(r) + 4 * h < A) best = fmaxf(best, qv[r]);
            best = fmaxf(best, __shfl_xor(best, 32));
            if (h == 0) Tg[c] = best;
        }
        lds_barrier();
        if (wv == 4) {                                            // delta_5 and the loss terms (:270-279)
            const float target = row_reward + a.gamma * Tg[c] * (1.0f - row_done);
            float qa = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa += (rho(r) + 4 * h == act) ? qv[r] : 0.0f;
            qa += __shfl_xor(qa, 32);
            const float td = live ? qa - target : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) Da[(rho(r) + 4 * h) * kLd + c] = (rho(r) + 4 * h == act) ? 2.0f * td : 0.0f;
            float sq = (h == 0) ? td * td : 0.0f, cnt = (h == 0 && live) ? 1.0f : 0.0f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
            rows_sum += cnt; sq_sum += sq;
        }
        lds_barrier();
        QSTAMP(3);
        // layer 5 (delta_5 in Da): dW5 on wavefront 0 | delta_4 -> Db on wavefront 1
        // (each delta product loads its weights at the start of its own phase: the other wavefront of the SIMD covers the wait,
        // and weights held a phase ahead do not fit 256 registers next to the phase's own operands)
        if (wv == 0) dw_accum(Da, A4, part_t, kSliceBlk5, 0, 0, c, h, first, &b5);
        if (wv == 1) { float wb5[16]; back_load<32>(wb5, net_w(n, 4), A, 32, 0, c, h); back_mul<32>(wb5, 0, Da, G4, Db, c, h); }
        const bool more = ti + G < n_tiles;
        if (wv == 7 && more) { const int r = tile_row(ti + G, c); if (h == 0) next_rows[c] = r; }
        lds_barrier();
        QSTAMP(4);
        rowc_next = more ? next_rows[c] : -1;
        have_pre = more;
        if (more) {
            if (grp) coop_fetch_rows<VEC>(xpre, a.states, a.stride, K1, rowc_next, wq, h);
            else coop_fetch_rows<VEC>(xpre, a.next_states, a.next_stride, K1, rowc_next, wq, h);
        }
        // layer 4 (delta_4 in Db): dW4 blocks on wavefronts 0, 1 | delta_3 tiles on 2, 3 -> Da
        if (wv == 0) dw_accum(Db, A3, part_t, kSliceBlk4 + 0, 0, 0, c, h, first, &b4);
        if (wv == 1) dw_accum(Db, A3, part_t, kSliceBlk4 + 1, 0, 1, c, h, first, nullptr);
        if (wv == 2 || wv == 3) { float wb4[16]; back_load<32>(wb4, net_w(n, 3), 32, 64, wv - 2, c, h); back_mul<32>(wb4, wv - 2, Db, G3, Da, c, h); }
        lds_barrier();
        QSTAMP(5);
        // layer 3 (delta_3 in Da): delta_2 tile wv -> Db on wavefronts 0-3 | dW3 block (row tile j >> 1, column tiles 2 (j & 1), + 1) on 4 + j
        if (wv < 4) {
            float wb3[32];
            back_load<64>(wb3, net_w(n, 2), 64, 128, wv, c, h);
            back_mul<64>(wb3, wv, Da, G2, Db, c, h);
        } else {
            const int j = wv - 4, rt = j >> 1, ct = 2 * (j & 1);
            dw_accum(Da, A2, part_t, kSliceBlk3 + 4 * rt + ct, rt, ct, c, h, first, (j & 1) == 0 ? &b3 : nullptr);
            dw_accum(Da, A2, part_t, kSliceBlk3 + 4 * rt + ct + 1, rt, ct + 1, c, h, first, nullptr);
        }
        lds_barrier();
        QSTAMP(6);
        // layer 2 (delta_2 in Db): delta_1 tile wv -> Da on wavefronts 0-3 | the four dW2 blocks of row tile j on 4 + j
        if (wv < 4) {
            float wb2[64];
            back_load<128>(wb2, net_w(n, 1), 128, 128, wv, c, h);
            back_mul<128>(wb2, wv, Db, G1, Da, c, h);
        } else {
            const int j = wv - 4;
#pragma unroll
            for (int it = 0; it < 4; ++it) dw_accum(Db, A1, part_t, kSliceBlk2 + 4 * j + it, j, it, c, h, first, it == 0 ? &b2 : nullptr);
        }
        lds_barrier();
        QSTAMP(7);
        // layer 1 (delta_1 in Da): block (row tile wv >> 1, input column tile wv & 1) per wavefront
        if ((wv & 1) == 0 || K1 > 32) dw_accum(Da, Xs, part_t, kSliceBlk1 + wv, wv >> 1, wv & 1, c, h, first, (wv & 1) == 0 ? &b1 : nullptr);
    }

    QSTAMP(8);
    if (used && K1 <= 32) {                                       // the second column tile of layer 1 was never touched
        for (int i = threadIdx.x; i < 4 * 1024; i += 512) part[(size_t)(kSliceBlk1 + 2 * (i >> 10) + 1) * 1024 + (i & 1023)] = 0.0f;
    }
    if (used && h0 == 0) {                                        // db rows: every bias is written by exactly one wavefront
        if (wv == 0) { part[kSliceBias + 352 + c0] = b5; part[kSliceBias + 320 + c0] = b4; }
        if (wv == 4 || wv == 6) part[kSliceBias + 256 + 32 * ((wv - 4) >> 1) + c0] = b3;
        if (wv >= 4) part[kSliceBias + 128 + 32 * (wv - 4) + c0] = b2;
        if ((wv & 1) == 0) part[kSliceBias + 32 * (wv >> 1) + c0] = b1;
    }
    float* wave_reward = Tg;                                      // (32 spare words)
    lds_barrier();
    if (lane == 0) wave_reward[wv] = reward_sum;
    if (wv == 4 && lane == 0) { wave_reward[8] = rows_sum; wave_reward[9] = sq_sum; }
    lds_barrier();
    if (wv == 0 && lane == 0) {
        float* ps = part + kSliceStats;
        ps[0] = wave_reward[8]; ps[1] = wave_reward[9];
        ps[2] = ((wave_reward[0] + wave_reward[1]) + (wave_reward[2] + wave_reward[3])) + ((wave_reward[4] + wave_reward[5]) + (wave_reward[6] + wave_reward[7]));
        ps[3] = used ? 1.0f : 0.0f;
    }
    QSTAMP(9);
}

// Launch 2: flat gradient = sum of the used slices; scal[0] += its squared norm (one atomic per workgroup); workgroup 0
// also totals the row count / squared TD error / reward and advances the optimizer step when there is something to learn.
struct ReduceArgs {
    const float* partials; int n_blocks, n_params, state_dim, n_actions;
    float* grad; float* scal;                 // scal: [0] sum g^2 (zeroed here by the previous step's AdamW), [1] rows, [2] sum td^2
    long long* step; double* reward_sum;      // reward_sum: nullptr or += sum of rewards over row_mask rows
    const float* win_reward; int n_windows;   // the select launch's per-window reward sums
    unsigned* meet;                           // fused form: [0..8) arrival counters (zero at launch), [8 + b] workgroup b's sum of g^2
    long long wait_ticks;                     // fused form: how long a workgroup waits at the meeting (100 MHz ticks)
    unsigned extra_arrivals;                  // fused form, test hook: arrivals counter 0 expects beyond the grid's (never met)
    long long* gave_up;                       // fused form: pinned host word, += 1 by a launch whose meeting was called off
};
constexpr unsigned kMeetOff = 0x80000000u;     // an arrival counter with this bit set: the meeting was called off

struct AdamArgs {
    float* params; float* target; const float* grad; float* m; float* v; const long long* step; float* scal; float* report;
    int n_params; float lr, wd, beta1, beta2, eps, max_norm; int update_freq;
};

// the AdamW update of parameter i (torch semantics: decoupled decay, bias-corrected moments) from the gradient SUM g over
// `count` rows, `ss` = squared norm of the summed gradient, t = optimizer step of this update
__device__ __forceinline__ void adamw_one(const AdamArgs& a, int i, float g_sum, float count, float ss, long long t) {
    const float inv = 1.0f / count;
    const float norm = sqrtf(ss) * inv;                                           // norm of the mean gradient
    const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f) * inv;            // torch.nn.utils.clip_grad_norm_
    const float bc1 = 1.0f - powf(a.beta1, (float)t), bc2 = 1.0f - powf(a.beta2, (float)t);
    const float step_size = a.lr / bc1, bc2_sqrt = sqrtf(bc2);
    const float g = g_sum * coef;
    float p = a.params[i] * (1.0f - a.lr * a.wd);
    const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
    p -= step_size * (m / (sqrtf(v) / bc2_sqrt + a.eps));
    a.m[i] = m; a.v[i] = v; a.params[i] = p;
    if (a.update_freq > 0 && (t % a.update_freq) == 0) a.target[i] = p;
}

// FUSED: the AdamW step rides in the same launch.  Every workgroup needs the squared norm of the WHOLE gradient for the
// clipping factor, so the workgroups (282 of 256 threads: all resident at once) meet at a counter after publishing their part
// of it; each then updates the parameters whose gradient sums it holds in registers.  Saves the third launch's dispatch
// and its 4.7 us for a wait of about one.  Everything a workgroup reads that another wrote in this launch goes through
// device-scope atomics (the norm, the counter): the per-XCD L2s are not coherent for plain loads.
template <bool FUSED>
__global__ __launch_bounds__(256) void qnet_grad_reduce_kernel(const ReduceArgs a, const AdamArgs w) {
    __shared__ float red[4];
    __shared__ float4 part8[8][32];
    // 128 slice elements per workgroup as 32 float4 columns; the workgroups' slices are split between the eight 32-lane groups
    // of the workgroup (a fixed split: the sum order does not depend on timing), eight 16-byte loads in flight per thread.
    // One element per thread over all 256 slices was a chain of 32 dependent L2 round trips on 127 of the 256 CUs:
    // 17.5 us for 33 MB; this form 9-12 us.
    static_assert(kSliceStats % 4 == 0 && kSlicePitch % 4 == 0, "float4 columns");
    const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int j0 = blockIdx.x * 128 + 4 * lane;                  // first of this thread's four slice elements
    const size_t pitch = (size_t)kSlicePitch;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (j0 < kSliceStats) {
        const float* p = a.partials + j0;
        // The slices that hold a gradient are those of workgroups 0 .. used - 1 (the training launch left the count): their
        // columns are loaded UNCONDITIONALLY, sixteen in flight per thread -- testing every slice's own flag first made each
        // pass two dependent round trips, and with a remainder loop a launch was eight of them (13.3 -> 11.5 us for 30 MB).
        // (Prefetching the slices' statistics for the totals below as well made the launch slower, 13.7 us: dropped.)
        const int used = min(a.n_blocks, (int)a.meet[kMeetUsed]);
        int b = (int)((long long)used * grp / 8);
        const int end = (int)((long long)used * (grp + 1) / 8);
        for (; b < end; b += 16) {
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (b + i < end) v[i] = *reinterpret_cast<const float4*>(p + (size_t)(b + i) * pitch);   // (the two halves of a wavefront are on different slices)
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    part8[grp][lane] = acc;
    __syncthreads();
    float g = 0.0f;                                              // threads 0..127: element j0' = blockIdx.x * 128 + threadIdx.x
    int pi = -1;
    if (threadIdx.x < 128) {
        const int j = blockIdx.x * 128 + threadIdx.x;
        if (j < kSliceStats) {
            const int col = threadIdx.x >> 2, comp = threadIdx.x & 3;
#pragma unroll
            for (int q = 0; q < 8; ++q) g += reinterpret_cast<const float*>(&part8[q][col])[comp];
            const int i = slice_param(j, a.state_dim, a.n_actions);  // -1: a padding element of the slice layout
            if (i >= 0 && i < a.n_params) { a.grad[i] = g; pi = i; } else g = 0.0f;
        }
    }
    float ss = g * g;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    __shared__ float bc[2];
    __shared__ long long step_old;
    if (threadIdx.x == 0) {
        const float part = (red[0] + red[1]) + (red[2] + red[3]);
        if (FUSED) {
            // this workgroup's part of the norm goes to its own word, then it is counted on one of 8 counters (282 arrivals
            // on one address serialise at ~60 ns each: 6 us of the first version of this meeting)
            step_old = *a.step;                                    // (workgroup 0 advances it after the meeting)
            __hip_atomic_store(reinterpret_cast<float*>(a.meet + 8 + blockIdx.x), part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the store has left before the arrival is sent after it
            asm volatile("" :: "v"((int)step_old));
            __hip_atomic_fetch_add(a.meet + (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsafeAtomicAdd(a.scal + 0, part);
        }
    }
    float rows = 0.0f, sq = 0.0f; double rew = 0.0;
    if ((FUSED || blockIdx.x == 0) && threadIdx.x < 64) {        // fused: every workgroup totals the row count itself
        for (int b = threadIdx.x; b < a.n_blocks; b += 64) {
            const float* ps = a.partials + b * pitch + kSliceStats;
            rows += ps[0]; sq += ps[1]; rew += (double)ps[2];
        }
        if (blockIdx.x == 0 && a.reward_sum) for (int wdw = threadIdx.x; wdw < a.n_windows; wdw += 64) rew += (double)a.win_reward[wdw];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { rows += __shfl_xor(rows, off); sq += __shfl_xor(sq, off); rew += __shfl_xor(rew, off); }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.scal[1] = rows; a.scal[2] = sq;
            if (!FUSED && rows > 0.0f && a.step) *a.step += 1;
            if (a.reward_sum) *a.reward_sum += rew;
        }
    }
    if (FUSED) {
        if (threadIdx.x < 64) {                                   // wavefront 0: lanes 0..7 watch one counter each
            // The meeting is all or nothing.  "Go" = all eight counters read exactly full.  A workgroup whose wait runs out
            // calls the meeting off by setting kMeetOff in a counter that is still short -- with a compare-and-swap from the
            // value it read, so the bit lands either before the missing arrival (that counter never reads full again: nobody
            // goes, before or after) or not at all (the counter moved: look again).  A full counter is never marked, a marked
            // one never reads full: no workgroup can apply its part of the update while another skips its own.
            unsigned want = (gridDim.x + 7 - (threadIdx.x & 7)) / 8;            // workgroups b with b % 8 == lane
            if (threadIdx.x == 0) want += a.extra_arrivals;
            bool met = false;
            for (const long long t0 = wall_clock64();;) {
                const unsigned got = threadIdx.x < 8 ? __hip_atomic_load(a.meet + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                if (__ballot((got & kMeetOff) != 0u) != 0ull) break;
                const unsigned long long short_of = __ballot(got != want);
                if (short_of == 0ull) { met = true; break; }
                if (wall_clock64() - t0 > a.wait_ticks) {
                    const int first = __ffsll((long long)short_of) - 1;
                    unsigned mine = 0u;
                    if ((int)threadIdx.x == first)
                        mine = atomicCAS(a.meet + threadIdx.x, got, got | kMeetOff) == got ? 1u : 0u;
                    if (__ballot(mine != 0u) != 0ull) break;
                    continue;                                      // the counter moved under the mark: read it again
                }
                __builtin_amdgcn_s_sleep(2);
            }
            float tot = 0.0f;                                      // the same order in every workgroup: the update does not depend on timing
            for (unsigned b = threadIdx.x; b < gridDim.x; b += 64)
                tot += __hip_atomic_load(reinterpret_cast<float*>(a.meet + 8 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
            if (threadIdx.x == 0) {
                bc[0] = tot; bc[1] = met ? rows : 0.0f;
                if (blockIdx.x == 0) {
                    w.report[3] = met ? 0.0f : -1.0f;
                    if (!met && a.gave_up) __hip_atomic_fetch_add(a.gave_up, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        __syncthreads();
        const float total_ss = bc[0], count = bc[1];
        const long long t = step_old + 1;
        if (count > 0.0f && pi >= 0) adamw_one(w, pi, g, count, total_ss, t);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (count > 0.0f) *a.step = t;
            a.scal[0] = total_ss;
            const float inv = count > 0.0f ? 1.0f / count : 0.0f;
            w.report[0] = count; w.report[1] = sq * inv; w.report[2] = sqrtf(total_ss) * inv;
        }
    }
}

// Launch 3, elementwise over the parameters: everything between loss.backward() and the end of train_step
// (Player.py:280-292): gradient /= #valid rows (MSELoss mean), clip_grad_norm_(max_norm), AdamW (torch semantics:
// decoupled decay, bias-corrected moments), target sync every update_freq optimizer steps (:289-290).  No valid row:
// nothing moves (the reference returns before the optimizer, :262).

__global__ __launch_bounds__(256) void qnet_adamw_kernel(const AdamArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float count = a.scal[1], sq = a.scal[2], ss = a.scal[0];
    const float inv = count > 0.0f ? 1.0f / count : 0.0f;
    if (count > 0.0f && i < a.n_params) adamw_one(a, i, a.grad[i], count, ss, *a.step);   // (step already advanced by the reduce launch)
    if (i == 0) { a.report[0] = count; a.report[1] = count > 0.0f ? sq * inv : 0.0f; a.report[2] = sqrtf(ss) * inv; }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
// The masked action selection runs sixteen rows per wavefront with the activations in registers (qnet_rows16.h) where the
// observation rows and layer 1 are whole float4s and the actions fit one output tile; anything else -- and PULSE_ACT_TILES=1,
// the A/B switch of the tests and tools/bench_trainer.py -- takes the cooperative 32-row tiles.
bool g_rows16_unavailable = false;       // the device refused the kernel's 147-156 KB of LDS once: the tiles from then on
bool act_rows16_ok(const QNetArgs& a) {
    const char* e = getenv("PULSE_ACT_TILES");                    // (read per call: the tests compare the two forms in one process)
    const bool tiles_forced = (e && e[0] == '1') || g_rows16_unavailable;
    const PulseQNet& n = a.net;
    return !tiles_forced && a.seat_idx && a.actions && n.state_dim % 4 == 0 && n.state_dim >= 13 && n.state_dim <= 64 && n.n_actions <= 16 &&
           a.row_stride % 4 == 0 && aligned16(a.states) && aligned16(n.w1);
}
int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus[dev] = prop.multiProcessorCount;
        else { (void)hipGetLastError(); cus[dev] = 256; }
    }
    return cus[dev];
}
constexpr int kActTwoLaunchRows = 262144;       // from here on the masked action selection lists first and runs full tiles second

// The fused reduce + AdamW launch is a meeting of all its workgroups inside one ordinary launch: it is only taken where the
// device can hold the whole grid at once (asked once per device; a smaller part, e.g. a CPX partition, gets the two-launch
// form), and a meeting that does not come about -- a co-tenant holding the CUs -- is called off as a whole, counted in
// this pinned word and reported by the next call (PULSE_EINTERNAL once; nothing was updated, training may go on).
long long* g_meet_gave_up = nullptr;
long long g_meet_gave_up_seen = 0;
bool fused_apply_fits(unsigned grid) {
    static int resident[64] = {0};                                 // per device: 0 = not asked yet, -1 = unknown
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (resident[dev] == 0) {
        int per_cu = 0; hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qnet_grad_reduce_kernel<true>, 256, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, dev) == hipSuccess) resident[dev] = std::max(per_cu * prop.multiProcessorCount, 1);
        else { (void)hipGetLastError(); resident[dev] = -1; }
    }
    if (!g_meet_gave_up) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
        g_meet_gave_up = static_cast<long long*>(p); *g_meet_gave_up = 0;
    }
    return resident[dev] >= (int)grid;
}

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
    if (select && act_rows16_ok(a)) {
        const int cus = device_cus();
        const bool wide = a.n_rows >= 1024 * cus;                 // large batches: windows of 1,024 candidates (~11 tiles for the 16 wavefronts)
        const int slot = (n.state_dim <= 48 ? 0 : 1) + (wide ? 2 : 0);
        const void* fns[4] = {reinterpret_cast<const void*>(&qnet_act_r16_kernel<3, 256>), reinterpret_cast<const void*>(&qnet_act_r16_kernel<4, 256>),
                              reinterpret_cast<const void*>(&qnet_act_r16_kernel<3, 1024>), reinterpret_cast<const void*>(&qnet_act_r16_kernel<4, 1024>)};
        const size_t lds_sizes[4] = {R16Lds<3, 256>::bytes, R16Lds<4, 256>::bytes, R16Lds<3, 1024>::bytes, R16Lds<4, 1024>::bytes};
        static const void* attr_set[4] = {nullptr, nullptr, nullptr, nullptr};
        if (attr_set[slot] != fns[slot]) {
            const hipError_t e = hipFuncSetAttribute(fns[slot], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sizes[slot]);
            if (e != hipSuccess) {                           // a device (partition) without that much LDS per workgroup: the tile kernels
                (void)hipGetLastError();
                g_rows16_unavailable = true;
                return launch(a, stream);
            }
            attr_set[slot] = fns[slot];
        }
        const int win = wide ? 1024 : 256;
        const unsigned n_win = (unsigned)((a.n_rows + win - 1) / win);
        void* params[1] = {const_cast<QNetArgs*>(&a)};
        const hipError_t le = hipLaunchKernel(fns[slot], dim3(std::min(n_win, (unsigned)cus)), dim3(kR16Threads), params, lds_sizes[slot], st);
        if (le != hipSuccess) return pulse::fail_hip((int)le, "pulse_qnet_act (sixteen rows per wavefront) launch");
    } else if (select && n.state_dim > 64) {        // wider inputs than the cooperative tile's LDS image: one wavefront per tile
        if (vec) hipLaunchKernelGGL((qnet_kernel<true, true>), dim3(grid), dim3(64), 0, st, a); else hipLaunchKernelGGL((qnet_kernel<true, false>), dim3(grid), dim3(64), 0, st, a);
    } else if (select) {
        const int slot = vec ? (n.state_dim <= 40 ? 0 : 1) : 2;
        const void* fns[3] = {reinterpret_cast<const void*>(&qnet_act4_kernel<true, kActWin, 5>),
                              reinterpret_cast<const void*>(&qnet_act4_kernel<true, kActWin, 8>),
                              reinterpret_cast<const void*>(&qnet_act4_kernel<false, kActWin, 8>)};
        const void* fn = fns[slot];
        static const void* attr_set[3] = {nullptr, nullptr, nullptr};
        if (attr_set[slot] != fn) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLdsBytes);
            if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_qnet_act: LDS size attribute");
            attr_set[slot] = fn;
        }
        const unsigned g4 = (unsigned)((a.n_rows + kActWin - 1) / kActWin);
        void* params[1] = {const_cast<QNetArgs*>(&a)};
        const hipError_t le = hipLaunchKernel(fn, dim3(g4), dim3(256), params, kActLdsBytes, st);
        if (le != hipSuccess) return pulse::fail_hip((int)le, "pulse_qnet_act launch");
        if (a.asel_counts) {                                 // two-launch form: the window launch listed the rows, this one runs them in full tiles
            const void* fns2[3] = {reinterpret_cast<const void*>(&qnet_act_rows_kernel<true, kActWin, 5>),
                                   reinterpret_cast<const void*>(&qnet_act_rows_kernel<true, kActWin, 8>),
                                   reinterpret_cast<const void*>(&qnet_act_rows_kernel<false, kActWin, 8>)};
            static const void* attr_set2[3] = {nullptr, nullptr, nullptr};
            if (attr_set2[slot] != fns2[slot]) {
                const hipError_t e2 = hipFuncSetAttribute(fns2[slot], hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLdsBytes);
                if (e2 != hipSuccess) return pulse::fail_hip((int)e2, "pulse_qnet_act: LDS size attribute");
                attr_set2[slot] = fns2[slot];
            }
            const unsigned tiles_at_most = (unsigned)((a.n_rows + 31) / 32);
            const hipError_t l2 = hipLaunchKernel(fns2[slot], dim3(std::min(tiles_at_most, 768u)), dim3(256), params, kActLdsBytes, st);   // three workgroups per CU
            if (l2 != hipSuccess) return pulse::fail_hip((int)l2, "pulse_qnet_act (rows) launch");
        }
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

namespace {
int act_call(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, const int32_t* seat_idx,
             int32_t q_seat, float epsilon, uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions,
             float* q_out, const uint8_t* terminated, uint8_t* row_mask_out, int32_t* select_scratch, int64_t select_words, void* stream);
}

int pulse_qnet_act(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, const int32_t* seat_idx,
                   int32_t q_seat, float epsilon, uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions,
                   float* q_out, const uint8_t* terminated, uint8_t* row_mask_out, void* stream) {
    return act_call(net, states, row_stride, n_rows, seat_idx, q_seat, epsilon, seed, step, table_id0, actions, q_out, terminated, row_mask_out,
                    nullptr, 0, stream);
}

int pulse_qnet_act_select(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, const int32_t* seat_idx,
                          int32_t q_seat, float epsilon, uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions,
                          const uint8_t* terminated, uint8_t* row_mask_out, int32_t* select_scratch, int64_t select_words, void* stream) {
    if (!select_scratch || !seat_idx || !row_mask_out || !net || net->state_dim < 13 || net->state_dim > 64)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_act_select: needs seat_idx, row_mask_out, select_scratch and state_dim 13..64");
    if (select_words < (int64_t)((n_rows + 255) / 256) * 259 + 512)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_act_select: select_scratch must hold 259 words per 256 rows + 512");
    return act_call(net, states, row_stride, n_rows, seat_idx, q_seat, epsilon, seed, step, table_id0, actions, nullptr, terminated, row_mask_out,
                    select_scratch, select_words, stream);
}

namespace {
int act_call(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, const int32_t* seat_idx,
             int32_t q_seat, float epsilon, uint64_t seed, uint64_t step, uint64_t table_id0, int64_t* actions,
             float* q_out, const uint8_t* terminated, uint8_t* row_mask_out, int32_t* select_scratch, int64_t select_words, void* stream) {
    if (!net || !actions) return pulse::fail(PULSE_EINVAL, "pulse_qnet_act: null argument");
    if (row_mask_out && (!seat_idx || net->state_dim > 64))
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_act: row_mask_out needs seat_idx and state_dim <= 64");
    QNetArgs a{};
    a.net = *net; a.states = states; a.row_stride = row_stride; a.n_rows = n_rows; a.seat_idx = seat_idx; a.q_seat = q_seat;
    a.epsilon = epsilon; a.seed = seed; a.step = step; a.table_id0 = table_id0; a.actions = actions; a.q_out = q_out;
    a.terminated = terminated; a.row_mask_out = row_mask_out;
    if (select_scratch) {
        const size_t nw = (size_t)((n_rows + 255) / 256);
        a.tsel_rows = select_scratch; a.tsel_counts = select_scratch + nw * 256;
        // large batches, scratch permitting: list the learner's rows per window, then run them in full tiles (qnet_act_rows_kernel)
        if (!act_rows16_ok(a) && n_rows >= kActTwoLaunchRows && select_words >= (int64_t)(nw * 517 + 512) && net->state_dim <= 64) {
            a.asel_rows = select_scratch + nw * 259 + 512; a.asel_counts = a.asel_rows + nw * 256;
        }
    }
    return launch(a, stream);
}
}  // namespace


#if PULSE_STAMPS
int pulse_debug_set_qnet_stamp_buffer(unsigned long long* buf) {
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_qstamp_buf), &buf, sizeof(buf));
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_debug_set_qnet_stamp_buffer");
}
#endif

int pulse_qnet_slice_floats(void) { return kSlicePitch; }

int64_t pulse_qnet_called_off_meetings(void) { return g_meet_gave_up ? (int64_t)__atomic_load_n(g_meet_gave_up, __ATOMIC_ACQUIRE) : 0; }

int pulse_qnet_param_count(int32_t state_dim, int32_t n_actions) {
    if (state_dim < 1 || n_actions < 1 || n_actions > 32) return pulse::fail(PULSE_EINVAL, "pulse_qnet_param_count: bad dimensions");
    return 128 * state_dim + 128 + 128 * 128 + 128 + 64 * 128 + 64 + 32 * 64 + 32 + 32 * n_actions + n_actions;
}

namespace {
AdamArgs adam_args(const PulseQNetTrain* t, int np) {
    AdamArgs b{};
    b.params = t->params; b.target = t->target_params; b.grad = t->grad; b.m = t->exp_avg; b.v = t->exp_avg_sq; b.step = (const long long*)t->step;
    b.scal = t->stats; b.report = t->report; b.n_params = np; b.lr = t->lr; b.wd = t->weight_decay; b.beta1 = t->beta1; b.beta2 = t->beta2;
    b.eps = t->eps; b.max_norm = t->max_grad_norm; b.update_freq = t->update_freq;
    return b;
}

int train_launches(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                   const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                   const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter, uint64_t table_id0,
                   uint8_t* terminated, double* reward_sum, bool grads, bool apply, void* stream) {
    if (!t || !states || !actions || !rewards || !next_states || !dones)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: null argument");
    const PulseQNet& n = t->net;
    if (n.state_dim < 13 || n.state_dim > 64 || n.n_actions < 1 || n.n_actions > 32)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: state_dim must be 13..64 (column 12 is the seat status) and n_actions 1..32");
    if (t->target.state_dim != n.state_dim || t->target.n_actions != n.n_actions)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: target network shape differs");
    if (!t->params || !t->target_params || !t->grad || !t->exp_avg || !t->exp_avg_sq || !t->step || !t->stats || !t->report || !t->partials)
        return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: null optimizer buffer");
    if (t->max_blocks < 1) return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: max_blocks < 1");
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
    const unsigned eg = (unsigned)((np + 255) / 256);
    bool fused = false;
    if (g_meet_gave_up) {
        const long long n = __atomic_load_n(g_meet_gave_up, __ATOMIC_ACQUIRE);
        if (n != g_meet_gave_up_seen) {
            g_meet_gave_up_seen = n;
            return pulse::fail(PULSE_EINTERNAL, "pulse_qnet_train_step: an earlier reduce + AdamW launch could not gather its workgroups within its wait "
                                                "(another process or stream holds the GPU's compute units?) and applied NO update (report[3] = -1); "
                                                "set PulseQNetTrain.separate_apply to run AdamW as a launch of its own");
        }
    }
    if (grads && n_rows > 0) {
        TrainArgs a{};
        a.net = FlatNet{t->params, n.state_dim, n.n_actions}; a.tgt = FlatNet{t->target_params, n.state_dim, n.n_actions}; a.partials = t->partials; a.scal = t->stats; a.n_params = np; a.states = states; a.stride = row_stride;
        a.actions = actions; a.rewards = rewards; a.next_states = next_states; a.next_stride = next_stride; a.dones = dones;
        a.n_rows = n_rows; a.seed = seed; a.step = step_counter; a.table_id0 = table_id0;
        a.gamma = t->gamma; a.drop_p = t->dropout_p;
        // select_scratch (int32 words), nw = ceil(n_rows / 256): [0, 256 nw) row lists; [256 nw, 258 nw) window counts (256-row
        // windows from the select launch, 128-row windows from pulse_qnet_act_select); [258 nw, 259 nw) window reward sums;
        // [259 nw, +512) the fused reduce launch's meeting words
        const int nw = (n_rows + 255) / 256;
        if (!t->select_scratch || t->select_words < (int64_t)nw * 259 + 512)
            return pulse::fail(PULSE_EINVAL, "pulse_qnet_train_step: select_scratch must hold 259 words per 256 rows + 512");
        int32_t* sel_rows = t->select_scratch; int32_t* sel_counts = t->select_scratch + (size_t)nw * 256;
        float* win_reward = reinterpret_cast<float*>(t->select_scratch + (size_t)nw * 258);
        a.sel_rows = sel_rows; a.sel_counts = sel_counts;
        a.meet = reinterpret_cast<unsigned*>(t->select_scratch + (size_t)nw * 259);
        int n_windows = 0;                                         // (windows with a reward sum: the select launch's)
        if (t->select_from_act) {                                  // the lists are pulse_qnet_act_select's, on this observation
            a.win_shift = 7; a.book = 1; a.row_mask = row_mask; a.terminated = terminated;
        } else {
            a.win_shift = 8; a.book = 0;
            n_windows = nw;
            SelectArgs sa{};
            sa.states = states; sa.stride = row_stride; sa.rewards = rewards; sa.dones = dones; sa.row_mask = row_mask; sa.terminated = terminated;
            sa.n_rows = n_rows; sa.sel_rows = sel_rows; sa.sel_counts = sel_counts; sa.win_reward = win_reward;
            hipLaunchKernelGGL(qnet_select_kernel, dim3((unsigned)nw), dim3(256), 0, st, sa);
        }
        const bool vec = n.state_dim % 8 == 0 && row_stride % 4 == 0 && next_stride % 4 == 0 && aligned16(states) && aligned16(next_states);
        // instances: layer-1 steps 5 (16-byte rows of <= 40 inputs) or 8; four or eight wavefronts per tile
        static const bool four = [] { const char* e = getenv("PULSE_TRAIN_WAVES"); return e && e[0] == '4'; }();
        const void* fns[6] = {reinterpret_cast<const void*>(&qnet_train_kernel<false, 8>), reinterpret_cast<const void*>(&qnet_train_kernel<true, 8>),
                              reinterpret_cast<const void*>(&qnet_train_kernel<true, 5>), reinterpret_cast<const void*>(&qnet_train8_kernel<false, 8>),
                              reinterpret_cast<const void*>(&qnet_train8_kernel<true, 8>), reinterpret_cast<const void*>(&qnet_train8_kernel<true, 5>)};
        const int slot = (vec ? (n.state_dim <= 40 ? 2 : 1) : 0) + (four ? 0 : 3);
        static const void* attr_set[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (attr_set[slot] != fns[slot]) {
            const hipError_t e = hipFuncSetAttribute(fns[slot], hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsBytes);
            if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_qnet_train_step: LDS size attribute");
            attr_set[slot] = fns[slot];
        }
        // persistent workgroups (157 KB of LDS: one per CU), one per possible tile of 32 rows at most
        const int grid = std::min((n_rows + 31) / 32, (int)t->max_blocks);
        void* params[1] = {&a};
        const hipError_t le = hipLaunchKernel(fns[slot], dim3((unsigned)grid), dim3(four ? 256 : 512), params, kTrainLdsBytes, st);
        if (le != hipSuccess) return pulse::fail_hip((int)le, "pulse_qnet_train_step launch");
    ReduceArgs r{};
    r.partials = t->partials; r.n_blocks = grid; r.n_params = np; r.state_dim = n.state_dim; r.n_actions = n.n_actions;
    r.grad = t->grad; r.scal = t->stats;
    r.step = apply ? (long long*)t->step : nullptr;        // gradients only: the caller advances the step after its all-reduce
    r.reward_sum = reward_sum; r.win_reward = win_reward; r.n_windows = n_windows; r.meet = a.meet;
    AdamArgs b = adam_args(t, np);
    const unsigned rg = (unsigned)((kSliceStats + 127) / 128);
    // one GPU: AdamW rides in the reduce launch -- where the whole grid fits the device at once
    fused = apply && !t->separate_apply && fused_apply_fits(rg);
    r.wait_ticks = t->meet_wait_ticks > 0 ? (long long)t->meet_wait_ticks : 500000000ll;     // 5 s of the 100 MHz clock
    r.extra_arrivals = t->debug_meet_extra > 0 ? (unsigned)t->debug_meet_extra : 0u;
    r.gave_up = g_meet_gave_up;
    if (fused) hipLaunchKernelGGL(qnet_grad_reduce_kernel<true>, dim3(rg), dim3(256), 0, st, r, b);
    else hipLaunchKernelGGL(qnet_grad_reduce_kernel<false>, dim3(rg), dim3(256), 0, st, r, b);
    }
    if (apply && !fused && (n_rows > 0 || !grads)) {
        const AdamArgs b = adam_args(t, np);
        hipLaunchKernelGGL(qnet_adamw_kernel, dim3(eg), dim3(256), 0, st, b);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_qnet_train_step launch");
}
}  // namespace

int pulse_qnet_train_step(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                          const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                          const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter, uint64_t table_id0,
                          uint8_t* terminated, double* reward_sum, void* stream) {
    return train_launches(t, states, row_stride, actions, rewards, next_states, next_stride, dones, row_mask, n_rows, seed, step_counter,
                          table_id0, terminated, reward_sum, true, true, stream);
}

int pulse_qnet_train_grads(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                           const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                           const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter, uint64_t table_id0,
                           uint8_t* terminated, double* reward_sum, void* stream) {
    return train_launches(t, states, row_stride, actions, rewards, next_states, next_stride, dones, row_mask, n_rows, seed, step_counter,
                          table_id0, terminated, reward_sum, true, false, stream);
}

int pulse_qnet_train_apply(const PulseQNetTrain* t, void* stream) {
    static const float dummy_f = 0.0f; static const int64_t dummy_a = 0; static const uint8_t dummy_d = 0;
    return train_launches(t, &dummy_f, 64, &dummy_a, &dummy_f, &dummy_f, 64, &dummy_d, nullptr, 0, 0, 0, 0, nullptr, nullptr, false, true, stream);
}

}  // extern "C"
