// envs.hip -- gfx950 step kernels for Blackjack, 2048 and Particle2D.
//
//   Blackjack   environments/blackjack/blackjack.py:23-186   one lane per game; the dealer's
//               data-dependent `while active_dealers.any()` loop (a host sync per card in the
//               reference) becomes a per-lane loop of at most a dozen iterations.
//   2048        environments/2048/TFE.py:17-108,152-189       one lane per board, the 4x4 board in
//               16 VGPRs (four int4 loads); rotation is an index permutation folded into the row
//               squash instead of buffer copies.
//   Particle2D  environments/Particle2D/Particle2D.py:22-30   one lane per particle, one float4 in,
//               two float4 out; fp32 with torch's op order (no contraction).
// These are HBM-streaming integer/fp32 kernels: one coalesced read and one coalesced write per
// state word; no LDS, no MFMA.
#include <hip/hip_runtime.h>

#include <mutex>

#include "pulse_internal.h"
#include "tfe_device.h"

namespace {

using namespace pulse_tfe;

constexpr int kBlock = 256;

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per word pair (v_mad_u64_u32): 32-bit integer multiplies are the slow vector instructions here
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// ------------------------------------------------------------------------------------ Blackjack
__device__ __forceinline__ int bj_rank(int card) { const int r = card % 13 + 1; return r > 10 ? 10 : r; }

// One lane per game.  The deck lives in the workgroup's LDS while it is shuffled (52 bytes per game: cards are 0..51)
// and leaves as coalesced dwords -- consecutive lanes write consecutive words of the block's 256 x 52 deck region,
// likewise for the two 256 x 20 card-row regions (a lane writing its own 52-word row strides 208 B between lanes:
// 1.5 ms per 1 M games; this form is bound by the 0.4 GB it writes).
__global__ __launch_bounds__(kBlock) void blackjack_reset_kernel(const PulseBlackjackView v, const int32_t* __restrict__ decks_src,
                                                                int32_t* __restrict__ decks_out, uint64_t seed, uint64_t episode) {
    __shared__ uint8_t deck[kBlock][53];                 // 53: odd stride, a lane's swaps do not pile onto one LDS bank
    __shared__ int32_t first[kBlock][4];                 // the four cards dealt at reset, as rank values (players 0/1, dealer 0/1)
    const int g0 = blockIdx.x * kBlock, g = g0 + threadIdx.x;
    const int in_block = min(kBlock, v.batch_size - g0);
    uint8_t* d = deck[threadIdx.x];
    const bool live = g < v.batch_size;
    if (live) {
        int c0, c1, c2, c3;                                  // the first four cards of the deck
        if (decks_src) {                                     // injected decks are copied as they are (any int32), below
            const int32_t* src = decks_src + (size_t)g * 52;
            c0 = src[0]; c1 = src[1]; c2 = src[2]; c3 = src[3];
        } else {
            // Fisher-Yates with Philox draws (replaces argsort(rand), blackjack.py:24-29)
            for (int c = 0; c < 52; ++c) d[c] = (uint8_t)c;
            for (int i = 51, q = 0; i > 0; --i, ++q) {
                const U4 r = philox4x32(seed, (uint64_t)g, episode * 16 + (uint64_t)(q >> 2));
                const uint32_t w = (q & 3) == 0 ? r.x : (q & 3) == 1 ? r.y : (q & 3) == 2 ? r.z : r.w;
                const int j = (int)__umulhi(w, (uint32_t)(i + 1));
                const uint8_t tmp = d[i]; d[i] = d[j]; d[j] = tmp;
            }
            c0 = d[0]; c1 = d[1]; c2 = d[2]; c3 = d[3];
        }
        int r1 = bj_rank(c0); const bool a1 = r1 == 1; if (a1) r1 = 11;                   // :53-59
        int d1 = bj_rank(c1); const bool da1 = d1 == 1; if (da1) d1 = 11;                 // :62-69
        int r2 = bj_rank(c2); const bool a2 = r2 == 1; if (a2) r2 = 11;                   // :72-78
        int d2 = bj_rank(c3); const bool dfirst = !da1 && d2 == 1; if (d2 == 1) d2 = 11;  // :81-87
        first[threadIdx.x][0] = r1; first[threadIdx.x][1] = r2; first[threadIdx.x][2] = d1; first[threadIdx.x][3] = d2;
        bool has = a1 || a2, dhas = da1 || dfirst;
        int ps = r1 + r2, ds = d1 + d2;
        if (ps > 21 && has) { ps -= 10; has = false; }                                    // :93-95
        if (ds > 21 && dhas) { ds -= 10; dhas = false; }                                  // :99-101
        v.players_card_idx[g] = 2; v.dealer_card_idx[g] = 2; v.deck_positions[g] = 4;
        v.dealer_upcard[g] = d1; v.player_card_sums[g] = ps; v.dealer_card_sums[g] = ds;
        v.has_ace[g] = has; v.dealer_has_ace[g] = dhas; v.terminated[g] = 0; v.rewards[g] = 0;
        v.obs[g * 3 + 0] = ps; v.obs[g * 3 + 1] = has; v.obs[g * 3 + 2] = d1;
    }
    __syncthreads();
    // coalesced write-out of the block's regions
    int32_t* dk = decks_out + (size_t)g0 * 52;
    if (decks_src) { for (int i = threadIdx.x; i < in_block * 52; i += kBlock) dk[i] = decks_src[(size_t)g0 * 52 + i]; }
    else { for (int i = threadIdx.x; i < in_block * 52; i += kBlock) dk[i] = deck[i / 52][i % 52]; }
    int32_t* pc = v.players_cards + (size_t)g0 * 20;
    int32_t* dc = v.dealer_cards + (size_t)g0 * 20;
    for (int i = threadIdx.x; i < in_block * 20; i += kBlock) {
        const int row = i / 20, c = i % 20;
        pc[i] = c < 2 ? first[row][c] : 0;
        dc[i] = c < 2 ? first[row][2 + c] : 0;
    }
}

__global__ __launch_bounds__(kBlock) void blackjack_step_kernel(const PulseBlackjackView v, const int64_t* __restrict__ actions) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= v.batch_size) return;
    const int32_t* d = v.decks + (size_t)g * 52;
    const long long a = actions[g];
    bool term = v.terminated[g] != 0;
    const bool hit = a == 0 && !term, stand = a == 1 && !term;                          // :117,:138
    int pos = v.deck_positions[g];
    int ps = v.player_card_sums[g], ds = v.dealer_card_sums[g];
    bool has = v.has_ace[g] != 0, dhas = v.dealer_has_ace[g] != 0;
    if (hit) {                                                                          // :118-135
        int rank = bj_rank((uint32_t)pos < 52u ? d[pos] : 0);
        const bool ace = rank == 1;
        if (ace && !has) rank = 11;
        const int ci = v.players_card_idx[g];
        if ((uint32_t)ci < 20u) v.players_cards[(size_t)g * 20 + ci] = rank;
        v.players_card_idx[g] = ci + 1;
        has = has || ace;   // (ace & ~already) | already
        ps += rank; pos += 1;
        if (ps > 21 && has) { ps -= 10; has = false; }
    }
    if (stand) {                                                                        // :139-160
        int ci = v.dealer_card_idx[g];
        bool active = ds < 17;
        while (active) {
            int rank = bj_rank((uint32_t)pos < 52u ? d[pos] : 0);
            const bool ace = rank == 1;
            if (ace && !dhas) rank = 11;
            if ((uint32_t)ci < 20u) v.dealer_cards[(size_t)g * 20 + ci] = rank;
            ci += 1;
            dhas = dhas || ace;
            ds += rank;
            if (ds > 21 && dhas) { ds -= 10; dhas = false; }
            pos += 1;
            active = ds < 17 && ds <= 21 && pos < 52;
        }
        v.dealer_card_idx[g] = ci;
    }
    int rew = 0;                                                                        // :183
    if (hit && ps > 21) { rew = -1; term = true; }                                      // :166-168
    if (stand) { rew = (ds > 21 || ps >= ds) ? 1 : -1; term = true; }                   // :171-177
    v.deck_positions[g] = pos; v.player_card_sums[g] = ps; v.dealer_card_sums[g] = ds;
    v.has_ace[g] = has; v.dealer_has_ace[g] = dhas; v.terminated[g] = term; v.rewards[g] = rew;
    v.obs[g * 3 + 0] = ps; v.obs[g * 3 + 1] = has; v.obs[g * 3 + 2] = v.dealer_upcard[g];
}

// ------------------------------------------------------------------------------------ 2048
// (the move itself: tfe_device.h)

// The row table of the packed 4 x 4 move (tfe_device.h): 65,536 entries, built on the device once per device.
__device__ uint32_t g_tfe_row_lut[65536];
__global__ __launch_bounds__(kBlock) void tfe_row_lut_kernel() {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    g_tfe_row_lut[r] = tfe_row_lut_entry(r);
}

// 4 x 4 boards, one lane per board.  The wavefront's 64 boards are one contiguous 4 KB block: it is loaded as four fully
// coalesced 16-byte-per-lane instructions (lane i of load j holds ROW (64 j + i) mod 4 of board (64 j + i) / 4 -- whose board
// that is does not matter to the encoding, which is cell by cell), every row is packed to 16 bits of 4-bit log2 tiles
// (tfe_device.h) and dropped into the wavefront's 512 bytes of LDS, from which each lane reads its own board as one 64-bit
// word; the move is four row-table lookups, spawn and game-over test run on the packed word, and the way out is the way in
// reversed.  (A lane loading its own 64 bytes touches four lines per instruction with sixteen bytes each.)  343 vector
// instructions per wavefront against the cell form's 1,430 (below: still the path of other board sides, of resets, and of a
// wavefront that meets a board the packed form cannot hold or that is the ragged last one).
__global__ __launch_bounds__(kBlock) void tfe_step4_kernel(int32_t* __restrict__ boards, int64_t* __restrict__ total_score,
                                                          const int64_t* __restrict__ actions, int32_t* __restrict__ rewards,
                                                          uint8_t* __restrict__ dones, int n_boards, uint64_t seed,
                                                          uint64_t board_id0, uint64_t step_counter, const uint32_t* __restrict__ lut) {
    __shared__ alignas(16) uint16_t rows_lds[kBlock / 64][256];
    const int g = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, g0 = g - lane;
    if (g0 >= n_boards) return;
    const bool whole = g0 + 64 <= n_boards;                                            // (wavefront-uniform)
    uint16_t* mine = rows_lds[threadIdx.x >> 6];
    int4* blk = reinterpret_cast<int4*>(boards + (size_t)g0 * 16);
    int4 in[4];
    int64_t ts = 0; int k = 0;
    if (whole) {
#pragma unroll
        for (int j = 0; j < 4; ++j) in[j] = blk[64 * j + lane];
        ts = total_score[g];
        k = (int)(actions[g] & 3);                                                      // TFE.py:154
    }
    __builtin_amdgcn_sched_barrier(0);                                                  // every load is in flight before anything waits
    bool fast = whole;
    PackedBoard pb{0u, 0u};
    if (whole) {
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t c[4] = {(uint32_t)in[j].x, (uint32_t)in[j].y, (uint32_t)in[j].z, (uint32_t)in[j].w};
            uint32_t row = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bad |= (c[q] & (c[q] - 1u)) | (c[q] & 0xFFFF8001u);                   // anything but 0 and 2^1 .. 2^14 (tfe_pack4)
                row |= (31u - (uint32_t)__clz((int)(c[q] | 1u))) << (4 * q);
            }
            mine[64 * j + lane] = (uint16_t)row;
        }
        fast = !__any(bad != 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (fast) {
            const uint2 w = *reinterpret_cast<const uint2*>(mine + 4 * lane);           // (a wavefront's LDS operations retire in order)
            pb.lo = w.x; pb.hi = w.y;
        }
    }
    if (fast) {
        const U4 rnd = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
        const int score = tfe_move_packed(pb, k, lut);
        const int empty_before = tfe_spawn_packed(pb, rnd.x, rnd.y);                   // TFE.py:182 (always)
        const bool over = tfe_over_packed(pb, empty_before);                           // TFE.py:48-67
        *reinterpret_cast<uint2*>(mine + 4 * lane) = make_uint2(pb.lo, pb.hi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        total_score[g] = ts + score;                                                    // TFE.py:168
        rewards[g] = score > 0 ? 31 - __clz(score) : 0;                                 // TFE.py:185-187
        dones[g] = over;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t row = mine[64 * j + lane];
            blk[64 * j + lane] = make_int4((int)((1u << (row & 15u)) & ~1u), (int)((1u << ((row >> 4) & 15u)) & ~1u),
                                           (int)((1u << ((row >> 8) & 15u)) & ~1u), (int)((1u << (row >> 12)) & ~1u));
        }
        return;
    }
    // the cell form: this lane's own board
    if (g >= n_boards) return;
    int b[16];
    int4* p4 = reinterpret_cast<int4*>(boards + (size_t)g * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int4 x = p4[q]; b[4 * q] = x.x; b[4 * q + 1] = x.y; b[4 * q + 2] = x.z; b[4 * q + 3] = x.w; }
    const U4 rnd = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
    const int score = tfe_move<4>(b, (int)(actions[g] & 3));
    tfe_spawn<4>(b, rnd.x, rnd.y);
    total_score[g] += score;
    rewards[g] = score > 0 ? 31 - __clz(score) : 0;
    dones[g] = tfe_over<4>(b);
#pragma unroll
    for (int q = 0; q < 4; ++q) p4[q] = make_int4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
}

template <int NB>
__global__ __launch_bounds__(kBlock) void tfe_step_kernel(int32_t* __restrict__ boards, int64_t* __restrict__ total_score,
                                                         const int64_t* __restrict__ actions, int32_t* __restrict__ rewards,
                                                         uint8_t* __restrict__ dones, int n_boards, uint64_t seed,
                                                         uint64_t board_id0, uint64_t step_counter, int is_reset) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) return;
    int b[NB * NB];
    int32_t* bp = boards + (size_t)g * NB * NB;
    const U4 rnd = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
    if (is_reset) {                                                                     // TFE.py:143-149
#pragma unroll
        for (int i = 0; i < NB * NB; ++i) b[i] = 0;
        tfe_spawn<NB>(b, rnd.x, rnd.y);
        tfe_spawn<NB>(b, rnd.z, rnd.w);
        total_score[g] = 0;
    } else {
        if (NB == 4) {
            const int4* p4 = reinterpret_cast<const int4*>(bp);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int4 x = p4[q]; b[4 * q] = x.x; b[4 * q + 1] = x.y; b[4 * q + 2] = x.z; b[4 * q + 3] = x.w; }
        } else {
#pragma unroll
            for (int i = 0; i < NB * NB; ++i) b[i] = bp[i];
        }
        const int k = (int)(actions[g] & 3);                                            // TFE.py:154
        const int score = tfe_move<NB>(b, k);
        total_score[g] += score;                                                        // TFE.py:168
        tfe_spawn<NB>(b, rnd.x, rnd.y);                                                 // TFE.py:182 (always)
        rewards[g] = score > 0 ? 31 - __clz(score) : 0;                                 // TFE.py:185-187
    }
    const bool over = tfe_over<NB>(b);                                                  // TFE.py:48-67
    if (!is_reset) dones[g] = over;
    if (NB == 4) {
        int4* p4 = reinterpret_cast<int4*>(bp);
#pragma unroll
        for (int q = 0; q < 4; ++q) p4[q] = make_int4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < NB * NB; ++i) bp[i] = b[i];
    }
}

// Any board side from 2 to 8 (the reference takes any board_height, TFE.py:112-131; the sides it is run at have the kernels
// above): one lane per board, the board in a per-lane array addressed at run time -- the reference's loops as they stand.
constexpr int kTfeAnyMax = 8;
__global__ __launch_bounds__(kBlock) void tfe_step_any_kernel(int32_t* __restrict__ boards, int64_t* __restrict__ total_score,
                                                             const int64_t* __restrict__ actions, int32_t* __restrict__ rewards,
                                                             uint8_t* __restrict__ dones, int n_boards, int n, uint64_t seed,
                                                             uint64_t board_id0, uint64_t step_counter, int is_reset) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) return;
    int x[kTfeAnyMax * kTfeAnyMax];
    const int cells = n * n;
    int32_t* bp = boards + (size_t)g * cells;
    const U4 rnd = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
    auto spawn = [&](uint32_t r_cell, uint32_t r_val) {                                 // TFE.py:17-34
        int ne = 0;
        for (int i = 0; i < cells; ++i) ne += x[i] == 0;
        if (!ne) return;
        const int kth = (int)__umulhi(r_cell, (uint32_t)ne);
        const int val = (float)(r_val >> 8) * (1.0f / 16777216.0f) > 0.9f ? 4 : 2;
        int seen = 0;
        for (int i = 0; i < cells; ++i) if (x[i] == 0) { if (seen == kth) x[i] = val; ++seen; }
    };
    int score = 0;
    if (is_reset) {                                                                     // TFE.py:143-149
        for (int i = 0; i < cells; ++i) x[i] = 0;
        spawn(rnd.x, rnd.y); spawn(rnd.z, rnd.w);
        total_score[g] = 0;
    } else {
        for (int i = 0; i < cells; ++i) x[i] = bp[i];
        const int k = (int)(actions[g] & 3);                                            // TFE.py:154
        for (int r = 0; r < n; ++r) {                                                   // row r of the board rotated k times (TFE.py:38-44, 158-178)
            int at[kTfeAnyMax], res[kTfeAnyMax];
            for (int c = 0; c < n; ++c) {
                int rr = r, cc = c;
                for (int q = 0; q < k; ++q) { const int nr = cc, nc = n - 1 - rr; rr = nr; cc = nc; }
                at[c] = rr * n + cc; res[c] = 0;
            }
            int w = 0; bool last_merged = false;
            for (int c = 0; c < n; ++c) {                                               // TFE.py:85-101
                const int val = x[at[c]];
                if (val == 0) continue;
                if (res[w] == 0) res[w] = val;
                else if (res[w] == val && !last_merged) { res[w] = val * 2; score += val * 2; last_merged = true; }
                else { w += 1; res[w] = val; last_merged = false; }
            }
            for (int c = 0; c < n; ++c) x[at[c]] = res[c];
        }
        total_score[g] += score;                                                        // TFE.py:168
        spawn(rnd.x, rnd.y);                                                            // TFE.py:182 (always)
        rewards[g] = score > 0 ? 31 - __clz(score) : 0;                                 // TFE.py:185-187
        bool over = true;                                                               // TFE.py:48-67
        for (int i = 0; i < cells; ++i) over = over && x[i] != 0;
        for (int r = 0; r < n; ++r) for (int c = 0; c + 1 < n; ++c) over = over && x[r * n + c] != x[r * n + c + 1];
        for (int r = 0; r + 1 < n; ++r) for (int c = 0; c < n; ++c) over = over && x[r * n + c] != x[(r + 1) * n + c];
        dones[g] = over;
    }
    for (int i = 0; i < cells; ++i) bp[i] = x[i];
}

// ------------------------------------------------------------------------------------ Particle2D
__global__ __launch_bounds__(kBlock) void particle2d_step_kernel(float4* __restrict__ state, const float2* __restrict__ action,
                                                                int32_t* __restrict__ steps, float4* __restrict__ obs_out,
                                                                float* __restrict__ rewards, uint8_t* __restrict__ terminated,
                                                                int n, float dt, int max_steps) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float2 a = action[i];
    a.x = fminf(fmaxf(a.x, -1.0f), 1.0f); a.y = fminf(fmaxf(a.y, -1.0f), 1.0f);       // :23
    s.z = __fadd_rn(s.z, __fmul_rn(a.x, dt)); s.w = __fadd_rn(s.w, __fmul_rn(a.y, dt)); // :24
    s.x = __fadd_rn(s.x, __fmul_rn(s.z, dt)); s.y = __fadd_rn(s.y, __fmul_rn(s.w, dt)); // :25
    const float dist = sqrtf(__fadd_rn(__fmul_rn(s.x, s.x), __fmul_rn(s.y, s.y)));      // :26 (IEEE sqrt: __fsqrt_rn is the native approximation)
    const float pen = __fmul_rn(0.001f, __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y)));
    const int st = steps[i] + 1;                                                        // :28
    state[i] = s; obs_out[i] = s;                                                       // :30 (clone)
    rewards[i] = __fsub_rn(-dist, pen);                                                 // :27
    steps[i] = st;
    terminated[i] = (dist < 0.1f) || (st >= max_steps);                                 // :29
}

int finish_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pulse::fail_hip((int)e, what);
    return 0;
}
inline dim3 grid1(int n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

int check_bj(const PulseBlackjackView* v) {
    if (!v || v->batch_size < 0) return pulse::fail(PULSE_EINVAL, "PulseBlackjackView: null / negative batch");
    const void* ptrs[] = {v->decks, v->deck_positions, v->players_cards, v->players_card_idx, v->player_card_sums, v->dealer_cards,
                          v->dealer_card_idx, v->dealer_upcard, v->dealer_card_sums, v->terminated, v->has_ace, v->dealer_has_ace,
                          v->rewards, v->obs};
    for (const void* p : ptrs) if (!p) return pulse::fail(PULSE_EINVAL, "PulseBlackjackView: null device pointer");
    return 0;
}

}  // namespace

namespace pulse {
// The current device's row table of the packed 2048 move (tfe_device.h), built at the first call on that device (a launch
// of 256 workgroups + one synchronisation, once).  qtable.hip's fused roll-out step uses it too.
int tfe_row_lut(const uint32_t** out) {
    static const uint32_t* table[64] = {nullptr};
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return fail_hip((int)e, "pulse_tfe: hipGetDevice");
    std::lock_guard<std::mutex> lock(mu);
    if (!table[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_tfe_row_lut));
        if (e != hipSuccess) return fail_hip((int)e, "pulse_tfe: row table symbol");
        hipLaunchKernelGGL(tfe_row_lut_kernel, dim3(65536 / kBlock), dim3(kBlock), 0, (hipStream_t)0);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)0);
        if (e != hipSuccess) return fail_hip((int)e, "pulse_tfe: row table build");
        table[dev] = static_cast<const uint32_t*>(p);
    }
    *out = table[dev];
    return 0;
}
}  // namespace pulse

extern "C" {

int pulse_blackjack_reset(const PulseBlackjackView* v, const int32_t* decks_src, int32_t* decks_out, uint64_t seed,
                          uint64_t episode, void* stream) {
    if (int rc = check_bj(v)) return rc;
    if (!decks_out) return pulse::fail(PULSE_EINVAL, "pulse_blackjack_reset: decks_out is null");
    if (v->batch_size == 0) return 0;
    hipLaunchKernelGGL(blackjack_reset_kernel, grid1(v->batch_size), dim3(kBlock), 0, (hipStream_t)stream, *v, decks_src, decks_out,
                       seed, episode);
    return finish_launch("pulse_blackjack_reset");
}

int pulse_blackjack_step(const PulseBlackjackView* v, const int64_t* actions, void* stream) {
    if (int rc = check_bj(v)) return rc;
    if (!actions) return pulse::fail(PULSE_EINVAL, "pulse_blackjack_step: actions is null");
    if (v->batch_size == 0) return 0;
    hipLaunchKernelGGL(blackjack_step_kernel, grid1(v->batch_size), dim3(kBlock), 0, (hipStream_t)stream, *v, actions);
    return finish_launch("pulse_blackjack_step");
}

static int tfe_launch(int32_t* boards, int64_t* total_score, const int64_t* actions, int32_t* rewards, uint8_t* dones,
                      int32_t n_boards, int32_t n, uint64_t seed, uint64_t board_id0, uint64_t step_counter, int is_reset,
                      void* stream) {
    if (!boards || !total_score || n_boards < 0) return pulse::fail(PULSE_EINVAL, "pulse_tfe: null boards/total_score");
    if (!is_reset && (!actions || !rewards || !dones)) return pulse::fail(PULSE_EINVAL, "pulse_tfe_step: null argument");
    if (n_boards == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (n == 4 && !is_reset && ((uintptr_t)boards & 15u) == 0) {
        const uint32_t* lut = nullptr;
        if (int rc = pulse::tfe_row_lut(&lut)) return rc;
        hipLaunchKernelGGL(tfe_step4_kernel, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed,
                           board_id0, step_counter, lut);
        return finish_launch("pulse_tfe_step");
    }
    switch (n) {
    case 3: hipLaunchKernelGGL(tfe_step_kernel<3>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    case 4: hipLaunchKernelGGL(tfe_step_kernel<4>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    case 5: hipLaunchKernelGGL(tfe_step_kernel<5>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    default:
        if (n < 2 || n > kTfeAnyMax) return pulse::fail(PULSE_EINVAL, "pulse_tfe: board side must be 2..8");
        hipLaunchKernelGGL(tfe_step_any_kernel, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, n, seed, board_id0, step_counter, is_reset);
    }
    return finish_launch("pulse_tfe");
}

int pulse_tfe_reset(int32_t* boards, int64_t* total_score, int32_t n_boards, int32_t n, uint64_t seed, uint64_t board_id0,
                    void* stream) {
    return tfe_launch(boards, total_score, nullptr, nullptr, nullptr, n_boards, n, seed, board_id0, 0, 1, stream);
}

int pulse_tfe_step(int32_t* boards, int64_t* total_score, const int64_t* actions, int32_t* rewards, uint8_t* dones,
                   int32_t n_boards, int32_t n, uint64_t seed, uint64_t board_id0, uint64_t step_counter, void* stream) {
    if (step_counter == 0) return pulse::fail(PULSE_EINVAL, "pulse_tfe_step: step_counter must be >= 1 (0 is the reset draw)");
    return tfe_launch(boards, total_score, actions, rewards, dones, n_boards, n, seed, board_id0, step_counter, 0, stream);
}

int pulse_particle2d_step(float* state, const float* action, int32_t* steps, float* obs_out, float* rewards,
                          uint8_t* terminated, int32_t n, float dt, int32_t max_steps, void* stream) {
    if (!state || !action || !steps || !obs_out || !rewards || !terminated || n < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_particle2d_step: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(particle2d_step_kernel, grid1(n), dim3(kBlock), 0, (hipStream_t)stream, reinterpret_cast<float4*>(state),
                       reinterpret_cast<const float2*>(action), steps, reinterpret_cast<float4*>(obs_out), rewards, terminated, n,
                       dt, max_steps);
    return finish_launch("pulse_particle2d_step");
}

}  // extern "C"
