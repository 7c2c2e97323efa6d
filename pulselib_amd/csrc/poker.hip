// poker.hip -- the hold'em kernels around the fused step (poker_step.hip): episode reset with the device shuffle
// and the evaluation cache, the stand-alone scripted-opponent policy (build_actions), the stand-alone evaluator,
// episode statistics and the hand-metrics side-channel.  gfx950 only.
#include <cstdlib>

#include "poker_device.h"
#include "hand_eval_device.h"

using namespace pulse_dev;

namespace {

constexpr int kLanes = 16;    // reset: lanes per table (one lane per seat, one DPP row)

__device__ __forceinline__ int grp_bcast(int v, int src) { return __shfl(v, src & 15, kLanes); }

// ---------------------------------------------------------------- standalone policy (build_actions)
__global__ __launch_bounds__(kBlock) void poker_policy_kernel(const float* __restrict__ obs, int obs_stride,
                                                             const int32_t* __restrict__ seat_idx, int n,
                                                             uint64_t types_packed, uint64_t seed, uint64_t step_counter,
                                                             uint64_t table_id0, int64_t* __restrict__ actions) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const int seat = seat_idx[t];
    const int type = (int)((types_packed >> (4 * (seat & 15))) & 15u);
    if (type == PULSE_AGENT_EXTERNAL) return;
    const float* o = obs + (size_t)t * obs_stride;
    const PolicyDraw draw = policy_draw(philox4x32(seed, table_id0 + (uint64_t)t, step_counter >> 1), step_counter);
    actions[t] = scripted_action(type, (int)o[5], (int)o[6], (int)o[9], draw);
}
// ---------------------------------------------------------------- standalone evaluator (tests / micro-bench)
__global__ __launch_bounds__(kBlock) void poker_eval_kernel(const int32_t* __restrict__ hr, uint32_t len,
                                                           const int32_t* __restrict__ cards, int n_hands, int n_cards,
                                                           int flop_double, int32_t* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_hands) return;
    const int32_t* c = cards + (size_t)i * n_cards;
    const int c5 = n_cards > 5 ? c[5] : 0, c6 = n_cards > 6 ? c[6] : 0;
    int p;
    if (n_cards == 5 && !flop_double) {
        p = 53;
        for (int k = 0; k < 5; ++k) p = hr_at(hr, len, p + c[k]);
        p = hr_at(hr, len, p);
    } else {
        p = walk7(hr, len, c[0], c[1], c[2], c[3], c[4], c5, c6);
    }
    out[i] = p;
}

// closed-form evaluator alone (tests): value of n_cards (5..7) distinct valid cards per hand
__global__ __launch_bounds__(kBlock) void poker_eval_closed_form_kernel(const int32_t* __restrict__ cards, int n_hands, int n_cards,
                                                                       int32_t* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_hands) return;
    HandAcc acc;
    for (int k = 0; k < n_cards; ++k) hand_add(acc, cards[(size_t)i * n_cards + k]);
    out[i] = hand_value(acc);
}

// ---------------------------------------------------------------- reset (PokerGPU.py:73-157)
// shuffle keys keep their top `shuffle_key_bits` bits (1..26; anything else: 26)
__device__ __forceinline__ int key_shift(const PulsePokerResetOpts& o) {
    return 32 - ((o.shuffle_key_bits > 0 && o.shuffle_key_bits <= 26) ? o.shuffle_key_bits : 26);
}

// ---- bitonic sort of 64 words over the 16 lanes of a table (one DPP row), four words per lane: element i = 4 * lane + q.
// Every merge starts with the mirror exchange i <-> i ^ (k - 1) and continues with i <-> i ^ j for j = k/4 ... 1, so
// that every compare-exchange is "the lower index keeps the smaller word" -- no direction flags.  Exchanges with
// j < 4 stay inside a lane (two min/max pairs); the others pair lanes l <-> l ^ m, each a single DPP pattern of a row:
// m = 1, 2, 3 quad_perm, 7 row_half_mirror, 15 row_mirror, 4 two rotations.  Keys are unique (the card index sits in
// their low six bits), so the sorted order is the stable order of the keys alone.
constexpr int kQuadMirror = 0x1B, kRowHalfMirror = 0x141, kRowMirror = 0x140, kRowRor12 = 0x12C;    // quad_perm:[3,2,1,0]
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v) { return (uint32_t)dpp_mov<CTRL>((int)v); }
__device__ __forceinline__ void cx(uint32_t& lo, uint32_t& hi) { const uint32_t a = min(lo, hi), b = max(lo, hi); lo = a; hi = b; }
// partner values (register q of the partner lane for plain exchanges, register 3 - q for mirrors) -> keep min or max
template <int CTRL, bool MIRROR> __device__ __forceinline__ void cx_lanes(uint32_t (&a)[4], bool lower) {
    uint32_t p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = dpp_u<CTRL>(a[MIRROR ? 3 - q : q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = lower ? min(a[q], p[q]) : max(a[q], p[q]);
}
__device__ __forceinline__ void cx_lanes_xor4(uint32_t (&a)[4], int lane) {          // l <-> l ^ 4: no single DPP pattern
    const bool lower = (lane & 4) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t up = dpp_u<kRowRor12>(a[q]), dn = dpp_u<kRowRor4>(a[q]);    // ror:n -- lane l reads lane (l - n) mod 16
        const uint32_t p = lower ? up : dn;
        a[q] = lower ? min(a[q], p) : max(a[q], p);
    }
}
__device__ __forceinline__ void sort64_in_row(uint32_t (&a)[4], int lane) {
    const bool b0 = (lane & 1) == 0, b1 = (lane & 2) == 0, b2 = (lane & 4) == 0, b3 = (lane & 8) == 0;
    cx(a[0], a[1]); cx(a[2], a[3]);                                                   // k = 2
    cx(a[0], a[3]); cx(a[1], a[2]); cx(a[0], a[1]); cx(a[2], a[3]);                   // k = 4
    cx_lanes<kQuadXor1, true>(a, b0);                                                 // k = 8: mirror (l ^ 1)
    cx(a[0], a[2]); cx(a[1], a[3]); cx(a[0], a[1]); cx(a[2], a[3]);
    cx_lanes<kQuadMirror, true>(a, b1);                                               // k = 16: mirror (l ^ 3)
    cx_lanes<kQuadXor1, false>(a, b0);
    cx(a[0], a[2]); cx(a[1], a[3]); cx(a[0], a[1]); cx(a[2], a[3]);
    cx_lanes<kRowHalfMirror, true>(a, b2);                                            // k = 32: mirror (l ^ 7)
    cx_lanes<kQuadXor2, false>(a, b1);
    cx_lanes<kQuadXor1, false>(a, b0);
    cx(a[0], a[2]); cx(a[1], a[3]); cx(a[0], a[1]); cx(a[2], a[3]);
    cx_lanes<kRowMirror, true>(a, b3);                                                // k = 64: mirror (l ^ 15)
    cx_lanes_xor4(a, lane);
    cx_lanes<kQuadXor2, false>(a, b1);
    cx_lanes<kQuadXor1, false>(a, b0);
    cx(a[0], a[2]); cx(a[1], a[3]); cx(a[0], a[1]); cx(a[2], a[3]);
}

// x mod m for the small operands of the reset (seat and button arithmetic, m <= 16): a multiply and a shift with a reciprocal
// from a table instead of the ~25 vector instructions of a 32-bit division by a run-time divisor -- this kernel is bound by
// vector-instruction issue.  Anything outside 0..255 (a poked button) takes the general path.
struct Rcp16Table {
    uint32_t v[PULSE_MAX_SEATS + 1];
    constexpr Rcp16Table() : v{} { for (int m = 1; m <= PULSE_MAX_SEATS; ++m) v[m] = 65536u / (uint32_t)m + 1u; }
};
__device__ const Rcp16Table kRcp16{};
__device__ __forceinline__ int mod_small(int x, int m, uint32_t rcp) {
    if ((uint32_t)x < 256u) return x - (int)(((uint32_t)x * rcp) >> 16) * m;       // exact: x < 256, m <= 16
    return pymod(x, m);
}
__device__ __forceinline__ int wrap_up(int x, int m) { return x >= m ? x - m : x; }     // x mod m for 0 <= x < 2 m
__device__ __forceinline__ int wrap_down(int x, int m) { return x < 0 ? x + m : x; }    // x mod m for -m <= x < m

#ifndef PULSE_RESET_WIDE
#define PULSE_RESET_WIDE 1        // 0: the per-wavefront narrow stores of rounds 1-3 (`make reset-narrow`: the A/B twin, tools/reset_ab.sh)
#endif
template <bool SHUFFLE>
__global__ __launch_bounds__(kBlock) void poker_reset_kernel(const PulsePokerView v, const PulsePokerResetOpts o) {
    __shared__ int32_t deck_s[kBlock / kLanes][52];
    __shared__ uint8_t valid_s[kBlock / kLanes];          // the table's 2A+5 cards are distinct and in 1..52 (its cache entry can be made)
    const int gt = blockIdx.x * kBlock + threadIdx.x;
    const int t = gt >> 4, s = gt & 15, g = threadIdx.x >> 4;
    if (t >= v.n_games) return;
    const int P = v.n_players, A = v.active_players;
    const bool seat = s < P, inA = s < A;
    const size_t row = (size_t)t * P + s;
    int32_t* dk = o.decks_out + (size_t)t * 52;

    // decks: prefixed copy (:88-92) or rank-of-random-key shuffle == rand().argsort()+1 (:86).
    // The table's 16 lanes sit in one wavefront, LDS ops of a wavefront retire in order, so the
    // wavefront-scope fences below are all the synchronisation the staging needs.
    if (SHUFFLE) {
        // card c = 4 s + q carries the word (key << 6 | c), key = the top bits of word q of lane s's Philox call; lanes
        // 13..15 carry twelve fillers above every card.  Sorted, element i sits in register i & 3 of lane i >> 2, and
        // the deck is the cards in that order: rand().argsort() + 1 (PokerGPU.py:86) with Philox keys.  (Round 2 counted,
        // for every card, the keys below its own -- 52 x 4 compare + add-with-carry per lane, 470 vector instructions
        // per wavefront against ~190 for the network.)
        uint32_t a[4];
        const int ks = key_shift(o);
        if (s < 13) {
            const U4 r = philox4x32(o.seed, o.table_id0 + (uint64_t)t, o.episode * 16 + (uint64_t)s);
            a[0] = (r.x >> ks) << 6 | (uint32_t)(4 * s); a[1] = (r.y >> ks) << 6 | (uint32_t)(4 * s + 1);
            a[2] = (r.z >> ks) << 6 | (uint32_t)(4 * s + 2); a[3] = (r.w >> ks) << 6 | (uint32_t)(4 * s + 3);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = 0xFFFFFFC0u | (uint32_t)(4 * s + q);
        }
        sort64_in_row(a, s);
        if (s < 13) {
#pragma unroll
            for (int q = 0; q < 4; ++q) deck_s[g][4 * s + q] = (int32_t)(a[q] & 63u) + 1;
        }
    } else {
        const int32_t* src = o.prefixed_decks + (size_t)t * 52;
        for (int c = s; c < 52; c += kLanes) deck_s[g][c] = src[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (measured and dropped: the decks of the workgroup's 16 tables leaving as one block after the barrier below -- fewer store
    // instructions, +1 us at every size: these stores overlap the rest of the kernel here, there they wait for the barrier)
    for (int c = s; c < 52; c += kLanes) dk[c] = deck_s[g][c];

    // statistics of the episode that ends here (the caller's per-episode sums, trainGPU.py:96,104), before its flags go
    if (o.stats_out) {
        float rf = 0.0f; bool dn = false;
        if (s == 0) { rf = o.stats_rewards[t]; dn = v.is_done[t] != 0; }
        const int d = __popcll(__ballot(dn));                       // (rows of tables past the end have left: they do not vote)
        const int t_row0 = t - ((threadIdx.x & 63) >> 4);           // first table of this wavefront
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // lane 16 k holds table k's reward
            const float x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rf), 16 * k));
            if (t_row0 + k < v.n_games) r += (double)x;
        }
        if ((threadIdx.x & 63) == 0) {
            // one of PULSE_STATS_SLOTS accumulators, a cache line each: atomics of all wavefronts onto ONE address take
            // ~5 ns apiece one after the other (16,384 wavefronts x 2: this kernel ran 360 us instead of 25)
            double* slot = o.stats_out + (size_t)((gt >> 6) & (PULSE_STATS_SLOTS - 1)) * PULSE_STATS_STRIDE;
            if (r != 0.0) atomicAdd(slot, r);
            if (d) atomicAdd(slot + 1, (double)d);
        }
    }
    // stacks: refill busted / over-max, then torch.roll by `rotation` (:101-110)
    int st = o.starting_bbs;
    if (seat && !o.first) {
        const int src = wrap_down(s - mod_small(o.rotation, P, kRcp16.v[P]), P);          // pymod(s - rotation, P), s < P
        st = v.stacks[(size_t)t * P + src];
        if (st == 0 || st > o.max_bbs) st = o.starting_bbs;
    }
    // the store of this seat's stack below depends on the load above, and a wavefront's load
    // instruction has returned for all its lanes by then: no lane overwrites a seat unread.
    int h0 = -1, h1 = -1;
    if (inA) { h0 = deck_s[g][2 * s]; h1 = deck_s[g][2 * s + 1]; }                     // :112-114
    // evaluation cache for the episode (include/pulse_env.h: pre_*): walk the table board-first once
    if (v.pre_board) {
        const int f0 = deck_s[g][2 * A + 1], f1 = deck_s[g][2 * A + 2], f2 = deck_s[g][2 * A + 3];   // burn, flop
        const int f3 = deck_s[g][2 * A + 5], f4 = deck_s[g][2 * A + 7];                               // burn, turn, burn, river
        unsigned long long m = 0;
        uint32_t bad = 0;
        if (inA) { bad = !(card_ok(h0) && card_ok(h1)); m = (1ull << (h0 & 63)) | (1ull << (h1 & 63)); }
        m = (unsigned long long)row_or((uint32_t)m) | (unsigned long long)row_or((uint32_t)(m >> 32)) << 32;   // over the table's 16 lanes: DPP
        bad = row_or(bad);
        const bool board_ok = card_ok(f0) && card_ok(f1) && card_ok(f2) && card_ok(f3) && card_ok(f4);
        m |= (1ull << (f0 & 63)) | (1ull << (f1 & 63)) | (1ull << (f2 & 63)) | (1ull << (f3 & 63)) | (1ull << (f4 & 63));
        const bool valid = !bad && board_ok && __popcll(m) == 2 * A + 5;    // 2A+5 distinct cards in 1..52
        // the per-seat part of the cache (three hand values, tag, class) is computed at the end of the kernel, by as many
        // threads of the workgroup as there are seats in its hands; seats outside the hand get an empty tag here
        if (s == 0) valid_s[g] = valid ? 1 : 0;
        if (seat && !(valid && inA)) v.pre_hands[row] = 0;
        if (s == 0) v.pre_board[t] = valid ? (int32_t)(pack_board(f0, f1, f2, f3, f4) | kPreBoardValid) : 0;
    }
    const int button = o.first ? 0 : mod_small(v.button[t] + 1, A, kRcp16.v[A]);        // :121
    int sb, bb, idx;
    if (A == 2) { sb = button; bb = wrap_up(button + 1, A); idx = button; }             // :123-125,:131  (button < A, A >= 2)
    else { sb = wrap_up(button + 1, A); bb = wrap_up(button + 2, A); idx = wrap_up(bb + 1, A); }
    int bet = 0, inv = 0, status = inA ? PULSE_ACTIVE : PULSE_SITOUT;
    if (s == bb) { st -= 1; bet = 1; inv = 1; status = st == 0 ? PULSE_ALLIN : PULSE_ACTIVE; }   // :188-199
    if (seat) {
        v.stacks[row] = st; v.current_round_bet[row] = bet; v.total_invested[row] = inv; v.status[row] = status;
        *reinterpret_cast<int2*>(v.hands + row * 2) = make_int2(h0, h1);
    }
    if (inA) v.equities[(size_t)t * A + s] = 0.5f;                                      // :144
#if PULSE_RESET_WIDE
    // The [N] scalars and the board leave from an LDS image of the workgroup's 16 tables: one 64-byte store instruction per
    // array by sixteen lanes of wavefront 0 instead of four 16-byte ones per array (-1.5 % .. -7 % at
    // 1,048,576 tables, -2 % at 131,072, +-0 at 65,536 -- the kernel is bound by vector-instruction issue at every size)
    __shared__ int32_t scal_s[4][kBlock / kLanes];
    if (s == 0) { scal_s[0][g] = button; scal_s[1][g] = sb; scal_s[2][g] = bb; scal_s[3][g] = idx; }
#else
    if (s < 5) v.board[t * 5 + s] = -1;                                                 // :95
    if (s == 0) {
        v.last_raise_size[t] = 1; v.deck_positions[t] = 2 * A; v.pots[t] = 1; v.stages[t] = 0;
        v.button[t] = button; v.sb[t] = sb; v.bb[t] = bb; v.idx[t] = idx;
        v.highest[t] = 1; v.agg[t] = bb; v.acted[t] = 0; v.is_done[t] = 0; v.is_done_out[t] = 0;
        v.equity_dirty[t] = 1; v.prev_stacks[t] = 0; v.prev_invested[t] = 0;
    }
#endif
    // first observation (:157 -> :159-179)
    float* __restrict__ ob = v.obs + (size_t)t * v.obs_size;
    const int a_h0 = grp_bcast(h0, idx), a_h1 = grp_bcast(h1, idx);
    const int a_stack = grp_bcast(st, idx), a_status = grp_bcast(status, idx), a_bet = grp_bcast(bet, idx);
    if (s < 13) {
        int hv;
        switch (s) {
        case 0: case 1: case 2: case 3: case 4: hv = -1; break;
        case 5: hv = a_h0; break; case 6: hv = a_h1; break; case 7: hv = 0; break;
        case 8: hv = wrap_down(idx - button, A); break; case 9: hv = 1; break; case 10: hv = 1 - a_bet; break;
        case 11: hv = a_stack; break; default: hv = a_status; break;
        }
        ob[s] = (float)hv;
    }
    if (s < v.max_players && s != idx) {
        int k; float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
        if (inA) { k = s - idx - 1; if (k < 0) k += A; f0 = (float)st; f1 = (float)status; f2 = (float)bet; }
        else k = s - 1;
        float* dst = ob + 13 + 3 * k;
        dst[0] = f0; dst[1] = f1; dst[2] = f2;
    }

    // ---- evaluation cache, per seat.  In the mapping above (16 lanes per table) only A of a table's 16 lanes hold a seat
    // of the hand, yet every wavefront would run the evaluator (~450 vector instructions) whatever A is.  The decks of the
    // workgroup's tables sit in LDS, so the (table, seat) pairs are dealt to the workgroup's threads in order instead:
    // 16 A pairs -- one wavefront for A <= 4, two for A <= 8, ... -- and the wavefronts beyond skip all of it.
    // (Wavefronts of tables past the end of the batch have left; the barrier counts only those still running.)
#if PULSE_RESET_WIDE
    {
        __syncthreads();                      // every lane has read its table's old button; the image is complete
        const int t0w = blockIdx.x * (kBlock / kLanes), nw = min(kBlock / kLanes, v.n_games - t0w), e = (int)threadIdx.x;
        if (e < nw) {
            const int tt = t0w + e;
            v.last_raise_size[tt] = 1; v.deck_positions[tt] = 2 * A; v.pots[tt] = 1; v.stages[tt] = 0;
            v.button[tt] = scal_s[0][e]; v.sb[tt] = scal_s[1][e]; v.bb[tt] = scal_s[2][e]; v.idx[tt] = scal_s[3][e];
            v.highest[tt] = 1; v.agg[tt] = scal_s[2][e]; v.acted[tt] = 0; v.is_done[tt] = 0; v.is_done_out[tt] = 0;
            v.equity_dirty[tt] = 1; v.prev_stacks[tt] = 0; v.prev_invested[tt] = 0;
        }
        for (int i = e; i < nw * 5; i += kBlock) v.board[t0w * 5 + i] = -1;            // :95
    }
#endif
    if (!v.pre_board) return;
#ifdef PULSE_FIVE_INDEX_LDS
    // experiment (profiles/README.md, r03): the evaluator's 16 KB index table staged in LDS by every workgroup
    __shared__ uint16_t five_s[8192];
    for (int w = threadIdx.x; w < 8192 / 8; w += kBlock) reinterpret_cast<int4*>(five_s)[w] = reinterpret_cast<const int4*>(kFiveIndex.v)[w];
    const uint16_t* five_index = five_s;
#else
    const uint16_t* five_index = kFiveIndex.v;
#endif
    __syncthreads();
    const int t_first = blockIdx.x * (kBlock / kLanes);
    const int tables_here = min(kBlock / kLanes, v.n_games - t_first);
    const int e = (int)threadIdx.x;
    if ((e & ~63) >= tables_here * A) return;                                   // the whole wavefront has no pair
    const int tb = (e * (65536 / A + 1)) >> 16, se = e - tb * A;                 // e / A, e % A (exact for e < 256, A <= 16)
    if (tb >= tables_here || !valid_s[tb]) return;
    {
        const int c0 = deck_s[tb][2 * se], c1 = deck_s[tb][2 * se + 1];
        const int f0 = deck_s[tb][2 * A + 1], f1 = deck_s[tb][2 * A + 2], f2 = deck_s[tb][2 * A + 3];
        const int f3 = deck_s[tb][2 * A + 5], f4 = deck_s[tb][2 * A + 7];
        // For distinct valid cards the table walk returns the closed-form hand value (hand_eval_device.h -- the
        // evaluator the table is generated from), so the three streets are computed instead of gathered: flop =
        // HR[value of the 5 cards] (:521 reads the table once more at the value itself: its hot first 150 KB),
        // turn = value of the 6 cards (:500), river / showdown = value of the 7 (:437-444).
        HandAcc acc;
        hand_add(acc, c0); hand_add(acc, c1); hand_add(acc, f0); hand_add(acc, f1); hand_add(acc, f2);
        const int v5 = hand_value(acc, five_index);
        hand_add(acc, f3);
        const int v6 = hand_value(acc, five_index);
        hand_add(acc, f4);
        const int r7 = hand_value(acc, five_index);
        const float vf = (float)hr_at_nb(v.hand_ranks, (uint32_t)v.hand_ranks_len, v5);   // :521
        const float vt = (float)v6;                                                     // :500
        float ef = __fdiv_rn(__fsub_rn(vf, 74359.0f), 749420.0f);                       // :523
        float et = __fdiv_rn(__fsub_rn(vt, 4109.0f), 32765.0f);                         // :502
        float er = __fdiv_rn(__fsub_rn((float)r7, 4109.0f), 32765.0f);                  // :481
        ef = fminf(fmaxf(ef, 0.0f), 1.0f); et = fminf(fmaxf(et, 0.0f), 1.0f); er = fminf(fmaxf(er, 0.0f), 1.0f);
        const size_t tt = (size_t)(t_first + tb);
        float* pe = v.pre_eq + tt * 3 * P;
        pe[se] = ef; pe[P + se] = et; pe[2 * P + se] = er;
        v.pre_rank[tt * P + se] = r7;
        // hole-card tag + the scripted players' class of the hand (poker_device.h: hand_class), both fixed for the episode
        v.pre_hands[tt * P + se] = (int32_t)(pack_hand(c0, c1) | hand_class(c0, c1) << kClsShift);
    }
}

// ---------------------------------------------------------------- episode statistics
__global__ __launch_bounds__(kBlock) void poker_stats_kernel(const uint8_t* __restrict__ is_done,
                                                            const float* __restrict__ rewards,
                                                            const uint8_t* __restrict__ mask, int n,
                                                            unsigned long long* stats, double* fstats) {
    int cnt = 0; double sum = 0.0;
    const int stride = gridDim.x * kBlock;
    int i = blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {          // four elements per pass: their loads are in flight together
        uint8_t d[4] = {0, 0, 0, 0}, mk[4] = {1, 1, 1, 1}; float r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (is_done) d[u] = is_done[i + u * stride];
            if (rewards) r[u] = rewards[i + u * stride];
            if (rewards && mask) mk[u] = mask[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { cnt += d[u] != 0; if (rewards && mk[u]) sum += (double)r[u]; }
    }
    for (; i < n; i += stride) {
        cnt += is_done ? (is_done[i] != 0) : 0;
        if (rewards && (!mask || mask[i])) sum += (double)rewards[i];
    }
    for (int m = 32; m >= 1; m >>= 1) { cnt += __shfl_xor(cnt, m); sum += __shfl_xor(sum, m); }
    __shared__ int scnt[kBlock / 64]; __shared__ double ssum[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { scnt[threadIdx.x >> 6] = cnt; ssum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0; double sm = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) { c += scnt[w]; sm += ssum[w]; }
        if (c) { if (stats) atomicAdd(stats, (unsigned long long)c); else atomicAdd(fstats + 1, (double)c); }
        if (rewards && sm != 0.0) atomicAdd(fstats, sm);
    }
}

// ---------------------------------------------------------------- hand metrics side-channel
// What scripts/Poker/trainGPU_performance.py:198-206 gathers per step with boolean indexing (a device->host sync each:
// `newly_done.any()`, stacks[newly_done, q_seat], stages[newly_done], positions[newly_done]) as sufficient statistics
// kept on the device: for every hand that finished in this step (done and not terminated before), its chip delta for
// the learner's seat is added to acc[position][street bucket] = {hands, wins (delta > 0), sum delta, sum delta^2}
// (int64: deltas are whole chips, the sums are exact).  position = (q_seat - button) mod A (utils/performance.py:55-59),
// bucket = min(stage, 4) with negatives clamped to 0 (:170-173).  LDS histogram per workgroup, then one atomic per
// touched cell.
__global__ __launch_bounds__(kBlock) void poker_hand_metrics_kernel(const uint8_t* __restrict__ dones, const uint8_t* __restrict__ terminated_before,
                                                                   const int32_t* __restrict__ stacks, int n_players,
                                                                   const int32_t* __restrict__ initial_q_stacks, const int32_t* __restrict__ stages,
                                                                   const int32_t* __restrict__ button, int q_seat, int active_players, int n,
                                                                   unsigned long long* __restrict__ acc, int step_index,
                                                                   int32_t* __restrict__ finish_step, int32_t* __restrict__ hand_delta) {
    __shared__ unsigned long long h[PULSE_MAX_SEATS * 5 * 4];
    for (int i = threadIdx.x; i < PULSE_MAX_SEATS * 5 * 4; i += kBlock) h[i] = 0ull;
    __syncthreads();
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < n; t += gridDim.x * kBlock) {
        if (!dones[t] || (terminated_before && terminated_before[t])) continue;
        const long long delta = (long long)stacks[(size_t)t * n_players + q_seat] - (long long)initial_q_stacks[t];
        // the ordered hand log (the rolling window of utils/performance.py:128-135 needs the hands in the order the
        // reference's per-step boolean pulls produce: by step, then by table): two words per table, written once
        if (finish_step) { finish_step[t] = step_index; hand_delta[t] = (int32_t)delta; }
        const int pos = pymod(q_seat - button[t], active_players);
        const int bucket = min(max(stages[t], 0), 4);                         // 0..3 the street, 4 = showdown (stage codes 4 and 5)
        unsigned long long* cell = h + ((pos & (PULSE_MAX_SEATS - 1)) * 5 + bucket) * 4;
        atomicAdd(cell + 0, 1ull);
        if (delta > 0) atomicAdd(cell + 1, 1ull);
        atomicAdd(cell + 2, (unsigned long long)delta);                       // two's complement: signed sums wrap correctly
        atomicAdd(cell + 3, (unsigned long long)(delta * delta));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PULSE_MAX_SEATS * 5 * 4; i += kBlock)
        if (h[i] != 0ull) atomicAdd(acc + i, h[i]);
}

// ---------------------------------------------------------------- PMC calibration (diagnostic)
// Streams `n_words` dwords with this library's access shape -- one dword per lane, lanes on consecutive
// addresses -- so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count
// (MI355X_MICROARCH.md, HBM section: widths other than 16 B/lane are uncalibrated on gfx950).
__global__ __launch_bounds__(kBlock) void calib_read_kernel(const int32_t* __restrict__ src, size_t n_words, int32_t* __restrict__ out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_words; i += (size_t)gridDim.x * kBlock) acc += src[i];
    if (acc == 0x7fffffff) out[0] = acc;      // keeps the loads alive without a store in the common case
}
__global__ __launch_bounds__(kBlock) void calib_write_kernel(int32_t* __restrict__ dst, size_t n_words, int32_t value) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_words; i += (size_t)gridDim.x * kBlock) dst[i] = value;
}
}  // namespace

// ---------------------------------------------------------------- host side
namespace pulse {

int check_view(const PulsePokerView* v, const char* who) {
    if (!v) return fail(PULSE_EINVAL, "null PulsePokerView");
    if (v->n_games < 0 || v->n_games > (1 << 24) || v->n_players < 2 || v->n_players > PULSE_MAX_SEATS || v->max_players > PULSE_MAX_SEATS ||
        v->max_players < v->n_players || v->active_players < 2 || v->active_players > v->n_players ||
        v->obs_size != 13 + 3 * (v->max_players - 1))
        return fail(PULSE_EINVAL, "PulsePokerView: unsupported shape (need n_games <= 2^24 per view, 2 <= active <= n_players <= max_players <= 16, obs_size = 13+3*(max_players-1))");
    const void* ptrs[] = {v->hand_ranks, v->pots, v->stages, v->deck_positions, v->button, v->sb, v->bb, v->idx, v->highest,
                          v->agg, v->acted, v->last_raise_size, v->prev_stacks, v->prev_invested, v->is_done, v->is_done_out,
                          v->equity_dirty, v->stacks, v->current_round_bet, v->total_invested, v->status, v->hands, v->board,
                          v->decks, v->equities, v->obs, v->w1, v->w2, v->K, v->alpha};
    for (const void* p : ptrs)
        if (!p) return fail(PULSE_EINVAL, "PulsePokerView: null device pointer");
    if (v->hand_ranks_len <= 53) return fail(PULSE_EINVAL, "PulsePokerView: hand_ranks_len too small");
    (void)who;
    return 0;
}

int finish_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip((int)e, what);
    return 0;
}

uint64_t pack_types(const uint8_t* agent_types, int n_players) {
    uint64_t packed = 0;
    for (int i = 0; i < n_players && i < 16; ++i) packed |= (uint64_t)(agent_types[i] & 15u) << (4 * i);
    return packed;
}

}  // namespace pulse

namespace {
inline int grid_for_tables(int n) { return (int)(((long long)n * kLanes + kBlock - 1) / kBlock); }
}

extern "C" {

int pulse_calib_stream(int32_t* buf, uint64_t n_words, int32_t write, void* stream) {
    if (!buf || n_words == 0) return pulse::fail(PULSE_EINVAL, "pulse_calib_stream: bad argument");
    const dim3 grid(2048), block(kBlock);
    if (write) hipLaunchKernelGGL(calib_write_kernel, grid, block, 0, (hipStream_t)stream, buf, (size_t)n_words, 7);
    else hipLaunchKernelGGL(calib_read_kernel, grid, block, 0, (hipStream_t)stream, buf, (size_t)n_words, buf);
    return pulse::finish_launch("pulse_calib_stream");
}

int pulse_poker_reset(const PulsePokerView* v, const PulsePokerResetOpts* o, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_reset")) return rc;
    if (!o || !o->decks_out) return pulse::fail(PULSE_EINVAL, "pulse_poker_reset: null options / decks_out");
    if (v->n_games == 0) return 0;
    const dim3 grid(grid_for_tables(v->n_games)), block(kBlock);
    if (o->prefixed_decks) hipLaunchKernelGGL((poker_reset_kernel<false>), grid, block, 0, (hipStream_t)stream, *v, *o);
    else hipLaunchKernelGGL((poker_reset_kernel<true>), grid, block, 0, (hipStream_t)stream, *v, *o);
    return pulse::finish_launch("pulse_poker_reset");
}

int pulse_poker_policy(const float* obs, int32_t obs_stride, const int32_t* seat_idx, int32_t n, const uint8_t* agent_types,
                       int32_t n_players, uint64_t seed, uint64_t step_counter, uint64_t table_id0, int64_t* actions,
                       void* stream) {
    if (!obs || !seat_idx || !agent_types || !actions || n < 0 || obs_stride < 10 || n_players < 1 || n_players > 16)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_policy: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(poker_policy_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, obs,
                       obs_stride, seat_idx, n, pulse::pack_types(agent_types, n_players), seed, step_counter, table_id0, actions);
    return pulse::finish_launch("pulse_poker_policy");
}

int pulse_poker_eval_hands(const int32_t* hand_ranks, int32_t hand_ranks_len, const int32_t* cards, int32_t n_hands,
                           int32_t n_cards, int32_t flop_double, int32_t* out, void* stream) {
    if (!hand_ranks || !cards || !out || n_hands < 0 || n_cards < 5 || n_cards > 7 || hand_ranks_len <= 53)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_eval_hands: bad argument");
    if (n_hands == 0) return 0;
    hipLaunchKernelGGL(poker_eval_kernel, dim3((n_hands + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       hand_ranks, (uint32_t)hand_ranks_len, cards, n_hands, n_cards, flop_double, out);
    return pulse::finish_launch("pulse_poker_eval_hands");
}

int pulse_poker_eval_closed_form(const int32_t* cards, int32_t n_hands, int32_t n_cards, int32_t* out, void* stream) {
    if (!cards || !out || n_hands < 0 || n_cards < 5 || n_cards > 7) return pulse::fail(PULSE_EINVAL, "pulse_poker_eval_closed_form: bad argument");
    if (n_hands == 0) return 0;
    hipLaunchKernelGGL(poker_eval_closed_form_kernel, dim3((n_hands + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       cards, n_hands, n_cards, out);
    return pulse::finish_launch("pulse_poker_eval_closed_form");
}

int pulse_poker_stats(const uint8_t* is_done, const float* rewards, const uint8_t* mask, int32_t n, int64_t* stats,
                      double* fstats, void* stream) {
    if ((!stats && !fstats) || (rewards && !fstats) || n < 0) return pulse::fail(PULSE_EINVAL, "pulse_poker_stats: bad argument");
    if (n == 0) return 0;
    // few, fat workgroups: every workgroup ends in atomics onto the same two or three words, and atomics from all XCDs
    // onto one address cost more than the reads (8 us per 65,536 tables with one workgroup per 256 tables)
    const int grid = max(1, min(256, n / (4 * kBlock)));
    hipLaunchKernelGGL(poker_stats_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, is_done, rewards, mask, n,
                       reinterpret_cast<unsigned long long*>(stats), fstats);
    return pulse::finish_launch("pulse_poker_stats");
}

int pulse_poker_hand_metrics(const uint8_t* dones, const uint8_t* terminated_before, const int32_t* stacks, int32_t n_players,
                             const int32_t* initial_q_stacks, const int32_t* stages, const int32_t* button, int32_t q_seat,
                             int32_t active_players, int32_t n, int64_t* acc, int32_t step_index, int32_t* finish_step,
                             int32_t* hand_delta, void* stream) {
    if (!dones || !stacks || !initial_q_stacks || !stages || !button || !acc || n < 0 || (finish_step != nullptr) != (hand_delta != nullptr))
        return pulse::fail(PULSE_EINVAL, "pulse_poker_hand_metrics: null argument (finish_step and hand_delta come together)");
    if (n_players < 2 || n_players > PULSE_MAX_SEATS || active_players < 2 || active_players > n_players || q_seat < 0 || q_seat >= n_players)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_hand_metrics: need 2 <= active_players <= n_players <= 16 and 0 <= q_seat < n_players");
    if (n == 0) return 0;
    const int grid = min(256, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(poker_hand_metrics_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, dones, terminated_before, stacks,
                       n_players, initial_q_stacks, stages, button, q_seat, active_players, n, reinterpret_cast<unsigned long long*>(acc),
                       step_index, finish_step, hand_delta);
    return pulse::finish_launch("pulse_poker_hand_metrics");
}

}  // extern "C"

