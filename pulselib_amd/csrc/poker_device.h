// poker_device.h -- device helpers shared by the hold'em kernels (poker_step.hip, poker.hip).
// Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "pulse_internal.h"

namespace pulse_dev {

constexpr int kBlock = 256;   // 4 wavefronts per workgroup in every kernel of the path

__device__ __forceinline__ int pymod(int x, int m) { int r = x % m; return r < 0 ? r + m : r; }

// ---------------------------------------------------------------- hand-rank walk
__device__ __forceinline__ int hr_at(const int32_t* __restrict__ hr, uint32_t len, int i) {
    return (uint32_t)i < len ? hr[(uint32_t)i] : 0;
}
// same value, but the load is unconditional (index clamped to slot 0, result masked): independent lookups
// written back to back stay back to back in the instruction stream instead of becoming branches
__device__ __forceinline__ int hr_at_nb(const int32_t* __restrict__ hr, uint32_t len, int i) {
    const bool ok = (uint32_t)i < len;
    const int x = hr[ok ? (uint32_t)i : 0u];
    return ok ? x : 0;
}
// seven dependent gathers: p = HR[p + c_i], p0 = 53 (PokerGPU.py:437-444).  A card of 0 re-reads
// slot 0 of the state, which is exactly the extra HR[p] / HR[HR[p]] lookups of the turn / flop
// equities (PokerGPU.py:500, :521), so all three streets share one 7-step chain.
__device__ __forceinline__ int walk7(const int32_t* __restrict__ hr, uint32_t len, int c0, int c1, int c2, int c3,
                                     int c4, int c5, int c6) {
    int p = 53;
    p = hr_at(hr, len, p + c0); p = hr_at(hr, len, p + c1); p = hr_at(hr, len, p + c2);
    p = hr_at(hr, len, p + c3); p = hr_at(hr, len, p + c4); p = hr_at(hr, len, p + c5);
    p = hr_at(hr, len, p + c6);
    return p;
}

// ---------------------------------------------------------------- evaluation cache tags
// pre_board: five 6-bit cards (the board this episode will deal) | valid bit 30.
// pre_hands: two 6-bit hole cards | valid bit 12.  A cached value is used only when the cards found
// in state at that moment are in 1..52 and equal the cached ones, so it is the value the reference's
// literal chain would produce (the table state after a set of distinct cards does not depend on order).
constexpr uint32_t kPreBoardValid = 1u << 30, kPreHandsValid = 1u << 12;
__device__ __forceinline__ bool card_ok(int c) { return (uint32_t)(c - 1) < 52u; }
__device__ __forceinline__ uint32_t pack_board(int b0, int b1, int b2, int b3, int b4) {
    return (uint32_t)(b0 & 63) | (uint32_t)(b1 & 63) << 6 | (uint32_t)(b2 & 63) << 12 | (uint32_t)(b3 & 63) << 18 | (uint32_t)(b4 & 63) << 24;
}
__device__ __forceinline__ uint32_t pack_hand(int h0, int h1) { return (uint32_t)(h0 & 63) | (uint32_t)(h1 & 63) << 6 | kPreHandsValid; }
// n_cards of the current board (3, 4 or 5) are dealt, valid and equal to the cached ones
__device__ __forceinline__ bool board_matches(uint32_t tag, int n_cards, int b0, int b1, int b2, int b3, int b4) {
    const uint32_t mask = (1u << (6 * n_cards)) - 1u;
    const bool in_range = card_ok(b0) && card_ok(b1) && card_ok(b2) && (n_cards < 4 || card_ok(b3)) && (n_cards < 5 || card_ok(b4));
    return (tag & kPreBoardValid) && in_range && ((pack_board(b0, b1, b2, b3, b4) ^ tag) & mask) == 0;
}

// ---------------------------------------------------------------- Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32 -> 64 multiply per word pair (v_mad_u64_u32): 32-bit integer multiplies are the slow VALU ops here
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ int rand_below(uint32_t r, int n) { return (int)__umulhi(r, (uint32_t)n); }
__device__ __forceinline__ float rand_unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// The scripted opponents' draw for (table, step): one Philox call serves two consecutive steps --
// call = Philox4x32-10(seed, table id, step >> 1); an even step takes words (x, y), an odd step (z, w).
// `pick` feeds the randint of the action, `coin` loose_passive's rand() (Player.py:146).
struct PolicyDraw { uint32_t pick, coin; };
__device__ __forceinline__ PolicyDraw policy_draw(const U4& call, uint64_t step) {
    // mask arithmetic, not `odd ? z : x`: the compiler turns that select into an indexed read of the call,
    // which then lives in scratch memory
    const uint32_t m = 0u - (uint32_t)(step & 1u);
    PolicyDraw d;
    d.pick = (call.x & ~m) | (call.z & m);
    d.coin = (call.y & ~m) | (call.w & m);
    return d;
}

// ---------------------------------------------------------------- scripted opponents
// environments/Poker/Player.py:79-176 + utils.py:121; c1,c2 = hole cards 1..52, pot = obs col 9.
// Everything the four hand-written players derive from the two hole cards alone is a handful of predicates -- the
// hand's CLASS, fixed for the episode like the cards: computed once (by the reset kernel into the spare bits of the
// evaluation cache's hole-card tag, or at a chunk's start) instead of at every step; what is left per step is the
// type's row of the table below, small_ball's pot thresholds and the two draws.
constexpr uint32_t kClsHhRaise = 1u << 0;     // heuristic_hands raises: pair / a king or ace, not both cards low   (Player.py:85-102)
constexpr uint32_t kClsTaRaise = 1u << 1;     // tight_aggressive raises                                              (:112-124)
constexpr uint32_t kClsLpCall = 1u << 2;      // loose_passive calls (and raises when its coin says so)              (:134-149)
constexpr uint32_t kClsSbStrong = 1u << 3;    // small_ball raises unless the pot rule folds                          (:159-174)
constexpr uint32_t kClsSbLow6 = 1u << 4;      // both cards below rank 6 (small_ball folds into pots > 30)
constexpr uint32_t kClsSbLow9 = 1u << 5;      // both cards below rank 9 (small_ball folds into pots > 80)
constexpr uint32_t kClsTaCall = 1u << 6;      // tight_aggressive does not fold
constexpr int kClsShift = 13;                 // where the class sits in a pre_hands word (above the 13-bit hole-card tag)
__device__ __forceinline__ uint32_t hand_class(int c1, int c2) {
    const int r1 = pymod(c1, 13), r2 = pymod(c2, 13);
    const int d = r1 > r2 ? r1 - r2 : r2 - r1;
    const bool pair = r1 == r2;
    const bool strong = pair || (r1 >= 10 && r2 > 5) || (r2 >= 10 && r1 > 5);          // tight_aggressive / small_ball raise hands
    const bool hh_fold = r1 < 8 && r2 < 8;
    const bool hh_raise = (pair || r1 >= 10 || r2 >= 10) && !hh_fold;
    const bool ta_fold = r1 < 7 && r2 < 7 && d > 5;
    const bool lp_fold = r1 <= 4 && r2 <= 4 && d > 9;
    const bool lp_call = ((pair && r1 > 8) || (r1 >= 11 && r2 > 9) || (r2 >= 11 && r1 > 9)) && !lp_fold;
    return (hh_raise ? kClsHhRaise : 0u) | (strong && !ta_fold ? kClsTaRaise : 0u) | (lp_call ? kClsLpCall : 0u) |
           (strong ? kClsSbStrong : 0u) | (r1 < 6 && r2 < 6 ? kClsSbLow6 : 0u) | (r1 < 9 && r2 < 9 ? kClsSbLow9 : 0u) |
           (!ta_fold ? kClsTaCall : 0u);
}
// Branch-free: the tables of a wavefront are played by different types, so a switch would run every case anyway
// (and pay a branch for each).  Every type reduces to `raise ? base + randint(n) : otherwise`, one multiply for the draw.
__device__ __forceinline__ int scripted_action_cls(int type, uint32_t cls, int pot, const PolicyDraw& rnd) {
    const bool t_rand = type == PULSE_AGENT_RANDOM, t_hh = type == PULSE_AGENT_HEURISTIC_HANDS, t_ta = type == PULSE_AGENT_TIGHT_AGGRESSIVE,
               t_lp = type == PULSE_AGENT_LOOSE_PASSIVE, t_sb = type == PULSE_AGENT_SMALL_BALL;
    // types 2..5 raise on class bits 0..3 (in that order), subject to loose_passive's coin and small_ball's pot rule
    const uint32_t slot = (uint32_t)(type - PULSE_AGENT_HEURISTIC_HANDS);
    const bool cand = slot < 4u && ((cls >> (slot & 3u)) & 1u) != 0u;
    const bool sb_fold = ((cls & kClsSbLow6) && pot > 30) || ((cls & kClsSbLow9) && pot > 80);
    const bool lp_raise = rand_unit(rnd.coin) > 0.9f;
    const bool draws = t_rand || (cand && (!t_lp || lp_raise) && (!t_sb || !sb_fold));
    const int n = t_rand ? 13 : t_hh ? 9 : t_sb ? 3 : 4;                                 // randint range (utils.py:121: randint(0, 13))
    const int base = t_rand ? 0 : t_ta ? 7 : 2;
    const int otherwise = (t_ta && (cls & kClsTaCall)) || (t_lp && (cls & kClsLpCall)) ? 1 : 0;   // the type's action when it does not raise
    return draws ? base + rand_below(rnd.pick, n) : otherwise;
}
__device__ __forceinline__ int scripted_action(int type, int c1, int c2, int pot, const PolicyDraw& rnd) {
    return scripted_action_cls(type, hand_class(c1, c2), pot, rnd);
}

// 32-bit byte-offset addressing: base pointers are wave-uniform (SGPR pair) and every array of a view is
// far below 4 GiB, so an access is `global_load v, v_off, s[base]` with no 64-bit VALU address arithmetic.
template <class T> __device__ __forceinline__ T ldo(const T* __restrict__ base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T> __device__ __forceinline__ void sto(T* __restrict__ base, uint32_t byte_off, T val) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = val;
}

// Store through a pointer that is known to be global memory (a pointer re-read from the kernel-argument segment is
// generic to the compiler: it would emit flat_store, which also counts against lgkmcnt and is waited for by LDS reads).
template <class T> __device__ __forceinline__ void stg(T* base, uint32_t byte_off, T val) {
    typedef __attribute__((address_space(1))) T GlobalT;
    *(GlobalT*)((__attribute__((address_space(1))) char*)base + byte_off) = val;
}

// The same store from inside a long loop: the byte offset passes through an empty asm, so the address cannot be
// hoisted out of the loop as a 64-bit per-lane pointer (two VGPRs alive across the whole loop per store site, which
// the register allocator then spills -- and a spill reload is a vector-memory load that waits behind every store in
// flight); it is re-formed where it is used as scalar base + 32-bit lane offset.
template <class T> __device__ __forceinline__ void sto_in_loop(T* base, uint32_t byte_off, T val) {
    asm volatile("" : "+v"(byte_off));
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = val;
}

// ---------------------------------------------------------------- DPP cross-lane steps (no LDS traffic)
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;                               // quad_perm:[1,0,3,2] / [2,3,0,1]
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

// reductions over the 4 lanes of a quad (two quad-permute steps) and over a 16-lane DPP row (four rotations)
#define PULSE_REDUCE(NAME, TYPE, OP)                                                                  \
    __device__ __forceinline__ TYPE quad_##NAME(TYPE v) {                                              \
        TYPE o = (TYPE)dpp_mov<kQuadXor1>((int)v); v = OP(v, o);                                       \
        o = (TYPE)dpp_mov<kQuadXor2>((int)v); v = OP(v, o);                                            \
        return v;                                                                                      \
    }                                                                                                  \
    __device__ __forceinline__ TYPE row_##NAME(TYPE v) {                                               \
        TYPE o = (TYPE)dpp_mov<kRowRor1>((int)v); v = OP(v, o);                                        \
        o = (TYPE)dpp_mov<kRowRor2>((int)v); v = OP(v, o);                                             \
        o = (TYPE)dpp_mov<kRowRor4>((int)v); v = OP(v, o);                                             \
        o = (TYPE)dpp_mov<kRowRor8>((int)v); v = OP(v, o);                                             \
        return v;                                                                                      \
    }
#define PULSE_OP_OR(a, b) ((a) | (b))
#define PULSE_OP_MIN(a, b) min((a), (b))
#define PULSE_OP_MAX(a, b) max((a), (b))
#define PULSE_OP_ADD(a, b) ((a) + (b))
PULSE_REDUCE(or, uint32_t, PULSE_OP_OR)
PULSE_REDUCE(imin, int, PULSE_OP_MIN)
PULSE_REDUCE(imax, int, PULSE_OP_MAX)
PULSE_REDUCE(sum, int, PULSE_OP_ADD)
#undef PULSE_REDUCE
// reduction over the LPT adjacent lanes that own one table (LPT = 2: one quad-permute step, 4: two)
template <int LPT> __device__ __forceinline__ uint32_t grp_or(uint32_t v) {
    static_assert(LPT == 2 || LPT == 4, "lanes per table");
    v |= (uint32_t)dpp_mov<kQuadXor1>((int)v);
    if (LPT == 4) v |= (uint32_t)dpp_mov<kQuadXor2>((int)v);
    return v;
}
template <int LPT> __device__ __forceinline__ int grp_imin(int v) {
    v = min(v, dpp_mov<kQuadXor1>(v));
    if (LPT == 4) v = min(v, dpp_mov<kQuadXor2>(v));
    return v;
}
template <int LPT> __device__ __forceinline__ int grp_imax(int v) {
    v = max(v, dpp_mov<kQuadXor1>(v));
    if (LPT == 4) v = max(v, dpp_mov<kQuadXor2>(v));
    return v;
}
template <int LPT> __device__ __forceinline__ int grp_isum(int v) {
    v += dpp_mov<kQuadXor1>(v);
    if (LPT == 4) v += dpp_mov<kQuadXor2>(v);
    return v;
}

// x mod A for x that is almost always within one period of [0, A): two conditional corrections,
// integer division only on the (poked-state) slow path.
__device__ __forceinline__ int mod_near(int x, int A) {
    if ((uint32_t)(x + A) < (uint32_t)(3 * A)) { x += x < 0 ? A : 0; x -= x >= A ? A : 0; return x; }
    return pymod(x, A);
}
// first seat (x+1 .. x+A) % A whose bit is set in `bits` (bits limited to seats < A); -1 if none.
__device__ __forceinline__ int first_after_near(uint32_t bits, int x, int A) {
    const int xm = mod_near(x, A);
    const uint32_t maskA = (1u << A) - 1u;
    const uint32_t rot = ((bits >> (xm + 1)) | (bits << (A - 1 - xm))) & maskA;   // bit k <-> seat (xm+1+k)%A
    if (!rot) return -1;
    const int seat = xm + 1 + (__ffs((int)rot) - 1);
    return seat >= A ? seat - A : seat;
}

// Sum a check point's partial done-counts (one workgroup) and publish {local, global = local, sequence number} into
// coherent pinned host memory, the sequence number last behind a system-scope fence: the host polls it.  `host` =
// nullptr: the device pair only.
template <int BLOCK = kBlock>
__device__ __forceinline__ void sum_and_publish(const uint32_t* __restrict__ partials, int n, long long* __restrict__ pair, long long* host, long long seq) {
    long long s = 0;
    int i = threadIdx.x;
    for (; i + 7 * BLOCK < n; i += 8 * BLOCK) {        // eight independent loads in flight (1 M tables: 65,536 words)
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = partials[i + u * BLOCK];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
    }
    for (; i < n; i += BLOCK) s += partials[i];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    __shared__ long long w_[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) w_[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int k = 0; k < BLOCK / 64; ++k) t += w_[k];
        if (pair) { pair[0] = t; pair[1] = t; }
        if (host) {
            host[0] = t; host[1] = t;
            __threadfence_system();
            *reinterpret_cast<volatile long long*>(host + 2) = seq;
            __threadfence_system();
        }
    }
}

}  // namespace pulse_dev

// host-side helpers shared by the translation units
namespace pulse {
int check_view(const PulsePokerView* v, const char* who);
int finish_launch(const char* what);
uint64_t pack_types(const uint8_t* agent_types, int n_players);
}  // namespace pulse
