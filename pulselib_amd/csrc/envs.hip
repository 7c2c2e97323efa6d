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
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
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
    switch (n) {
    case 3: hipLaunchKernelGGL(tfe_step_kernel<3>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    case 4: hipLaunchKernelGGL(tfe_step_kernel<4>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    case 5: hipLaunchKernelGGL(tfe_step_kernel<5>, grid1(n_boards), dim3(kBlock), 0, st, boards, total_score, actions, rewards, dones, n_boards, seed, board_id0, step_counter, is_reset); break;
    default: return pulse::fail(PULSE_EINVAL, "pulse_tfe: board side must be 3, 4 or 5");
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
