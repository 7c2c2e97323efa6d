// poker.hip -- hand-written gfx950 kernels for the batched no-limit hold'em step.
//
// Replaces the ~1,700 eager torch dispatches of one PokerGPU.step
// (environments/Poker/PokerGPU.py:527-633) by ONE launch.
//
// Mapping: 16 lanes per table (one lane per seat, P <= 16), 4 tables per 64-wide wavefront,
// 16 tables per 256-thread workgroup.  Per-table scalars are held replicated in the 16 lanes of
// the table's DPP row; per-seat rows ([N,P] int32, exactly the reference's layout) are one
// coalesced dword per lane.  Counting seats, "first ACTIVE seat after x" and winner selection are
// wavefront ballots + bit tricks on the table's 16-bit slice; actor values move with ds_bpermute;
// side-pot layers use 16-lane min/max butterflies.  All integer arithmetic is the reference's,
// the fp32 reward keeps torch's op order (no contraction; tanh rounded once from double).
//
// Memory: state is read once and written once per step in the reference's own SoA tensors, so
// the drop-in class can expose them unchanged.  The 130 MB hand-rank table is only touched by
// tables that need an evaluation (dirty equity or showdown); its top levels live in L2, the rest
// in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "pulse_internal.h"

namespace {

constexpr int kLanes = 16;    // lanes per table
constexpr int kBlock = 256;   // 4 wavefronts, 16 tables

// ---------------------------------------------------------------- 16-lane group primitives
__device__ __forceinline__ uint32_t grp_ballot(bool p) {
    const unsigned long long b = __ballot(p);
    return (uint32_t)(b >> (threadIdx.x & 48)) & 0xFFFFu;
}
__device__ __forceinline__ int grp_bcast(int v, int src) { return __shfl(v, src & 15, kLanes); }
__device__ __forceinline__ float grp_bcastf(float v, int src) { return __shfl(v, src & 15, kLanes); }
__device__ __forceinline__ int grp_min(int v) {
    v = min(v, __shfl_xor(v, 1, kLanes)); v = min(v, __shfl_xor(v, 2, kLanes));
    v = min(v, __shfl_xor(v, 4, kLanes)); v = min(v, __shfl_xor(v, 8, kLanes));
    return v;
}
__device__ __forceinline__ int grp_max(int v) {
    v = max(v, __shfl_xor(v, 1, kLanes)); v = max(v, __shfl_xor(v, 2, kLanes));
    v = max(v, __shfl_xor(v, 4, kLanes)); v = max(v, __shfl_xor(v, 8, kLanes));
    return v;
}
__device__ __forceinline__ int pymod(int x, int m) { int r = x % m; return r < 0 ? r + m : r; }

// first seat (x+1 .. x+A) % A whose bit is set in `bits` (bits limited to seats < A); -1 if none.
__device__ __forceinline__ int first_after(uint32_t bits, int x, int A) {
    const int xm = pymod(x, A);
    const uint32_t maskA = (1u << A) - 1u;
    const uint32_t rot = ((bits >> (xm + 1)) | (bits << (A - 1 - xm))) & maskA;   // bit k <-> seat (xm+1+k)%A
    if (!rot) return -1;
    int seat = xm + 1 + (__ffs((int)rot) - 1);
    return seat >= A ? seat - A : seat;
}

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

// ---------------------------------------------------------------- scripted opponents
// environments/Poker/Player.py:79-176 + utils.py:121; c1,c2 = hole cards 1..52, pot = obs col 9.
__device__ __forceinline__ int scripted_action(int type, int c1, int c2, int pot, const U4& rnd) {
    const int r1 = pymod(c1, 13), r2 = pymod(c2, 13);
    const int d = r1 > r2 ? r1 - r2 : r2 - r1;
    const bool pair = r1 == r2;
    int a = 0;
    switch (type) {
    case PULSE_AGENT_RANDOM:                                              // utils.py:121
        a = rand_below(rnd.x, 13); break;
    case PULSE_AGENT_HEURISTIC_HANDS: {                                   // Player.py:85-102
        const bool fold = r1 < 8 && r2 < 8;
        const bool raise = (pair || r1 >= 10 || r2 >= 10) && !fold;
        a = raise ? 2 + rand_below(rnd.x, 9) : 0; break; }
    case PULSE_AGENT_TIGHT_AGGRESSIVE: {                                  // Player.py:112-124
        const bool fold = r1 < 7 && r2 < 7 && d > 5;
        const bool raise = (pair || (r1 >= 10 && r2 > 5) || (r2 >= 10 && r1 > 5)) && !fold;
        a = fold ? 0 : 1;
        if (raise) a = 2 + 5 + rand_below(rnd.x, 4);
        break; }
    case PULSE_AGENT_LOOSE_PASSIVE: {                                     // Player.py:134-149
        const bool fold = r1 <= 4 && r2 <= 4 && d > 9;
        const bool call = ((pair && r1 > 8) || (r1 >= 11 && r2 > 9) || (r2 >= 11 && r1 > 9)) && !fold;
        const bool raise = rand_unit(rnd.y) > 0.9f && call;
        a = call ? 1 : 0;
        if (raise) a = 2 + rand_below(rnd.x, 4);
        break; }
    case PULSE_AGENT_SMALL_BALL: {                                        // Player.py:159-174
        const bool fold = (r1 < 6 && r2 < 6 && pot > 30) || (r1 < 9 && r2 < 9 && pot > 80);
        const bool raise = (pair || (r1 >= 10 && r2 > 5) || (r2 >= 10 && r1 > 5)) && !fold;
        a = raise ? 2 + rand_below(rnd.x, 3) : 0; break; }
    default: break;
    }
    return a;
}

// 32-bit byte-offset addressing: base pointers are wave-uniform (SGPR pair) and every array of a view is
// far below 4 GiB, so an access is `global_load v, v_off, s[base]` with no 64-bit VALU address arithmetic.
template <class T> __device__ __forceinline__ T ldo(const T* __restrict__ base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T> __device__ __forceinline__ void sto(T* __restrict__ base, uint32_t byte_off, T val) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = val;
}

// ---------------------------------------------------------------- the fused step
struct PolicyArgs {
    uint64_t types_packed, seed, step_counter, table_id0;
    uint32_t* wave_done;                   // nullptr, or one word per wavefront of the launch: tables done after this step
};
// In-kernel timeline (diagnostic build only, -DPULSE_STAMPS=1 -> libpulse_hip_stamps.so; no stamp executes
// in the product): lane 0 of every wavefront stores s_memtime at phase boundaries into a buffer of its own.
#ifndef PULSE_STAMPS
#define PULSE_STAMPS 0
#endif
#if PULSE_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(i) do { if ((threadIdx.x & 63) == 0 && g_stamp_buf) { __builtin_amdgcn_sched_barrier(0); \
    g_stamp_buf[((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef PULSE_TANH_F32
#define PULSE_TANH_F32 0
#endif
#ifndef PULSE_KERNARG_EARLY
#define PULSE_KERNARG_EARLY 1
#endif
#ifndef PULSE_PREFETCH_DECK
#define PULSE_PREFETCH_DECK 0
#endif
constexpr bool kPrefetchDeck = PULSE_PREFETCH_DECK != 0;   // 1: read the next street's cards up front (latency) ; 0: only on a deal (bytes)

// A table is owned by LPT adjacent lanes (LPT = 1, 2, 4, 8 or 16: a DPP quad/row fraction, never
// straddling a wavefront); lane j of the group owns seats j, j+LPT, j+2*LPT, ... (SPL of them).
// Per-table scalars are replicated in the group's lanes, so the scalar part of the state machine
// costs 1/LPT wave-instructions per table; per-seat work is SPL unrolled iterations.  LPT = 4 is
// the default: cross-seat reductions are two quad-permute DPP steps (no LDS), 16 tables share a
// wavefront, and 65,536 tables still give 4 waves per SIMD to hide the gather latency.
// Cross-lane steps are DPP modifiers on the VALU op (no LDS traffic): quad_perm inside a quad,
// row rotations inside a 16-lane row.  Every lane of a table is active whenever one is (all branches
// around these calls are table-uniform), so no step reads a disabled lane.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;                               // quad_perm:[1,0,3,2] / [2,3,0,1]
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

#define PULSE_GRP_REDUCE(NAME, TYPE, OP)                                                            \
    template <int LPT> __device__ __forceinline__ TYPE NAME(TYPE v) {                                \
        static_assert(LPT == 1 || LPT == 2 || LPT == 4 || LPT == 16, "unsupported lanes per table"); \
        if (LPT == 2 || LPT == 4) { const TYPE o = (TYPE)dpp_mov<kQuadXor1>((int)v); v = OP(v, o); } \
        if (LPT == 4) { const TYPE o = (TYPE)dpp_mov<kQuadXor2>((int)v); v = OP(v, o); }             \
        if (LPT == 16) {                                                                             \
            TYPE o = (TYPE)dpp_mov<kRowRor1>((int)v); v = OP(v, o);                                  \
            o = (TYPE)dpp_mov<kRowRor2>((int)v); v = OP(v, o);                                       \
            o = (TYPE)dpp_mov<kRowRor4>((int)v); v = OP(v, o);                                       \
            o = (TYPE)dpp_mov<kRowRor8>((int)v); v = OP(v, o);                                       \
        }                                                                                            \
        return v;                                                                                    \
    }
#define PULSE_OP_OR(a, b) ((a) | (b))
#define PULSE_OP_MIN(a, b) min((a), (b))
#define PULSE_OP_MAX(a, b) max((a), (b))
PULSE_GRP_REDUCE(grp_or, uint32_t, PULSE_OP_OR)
PULSE_GRP_REDUCE(grp_imin, int, PULSE_OP_MIN)
PULSE_GRP_REDUCE(grp_imax, int, PULSE_OP_MAX)
#define PULSE_OP_ADD(a, b) ((a) + (b))
PULSE_GRP_REDUCE(grp_sum, int, PULSE_OP_ADD)
// x mod A for x that is almost always within one period of [0, A): two conditional corrections,
// integer division only on the (poked-state) slow path.
__device__ __forceinline__ int mod_near(int x, int A) {
    if ((uint32_t)(x + A) < (uint32_t)(3 * A)) { x += x < 0 ? A : 0; x -= x >= A ? A : 0; return x; }
    return pymod(x, A);
}
__device__ __forceinline__ int first_after_near(uint32_t bits, int x, int A) {
    const int xm = mod_near(x, A);
    const uint32_t maskA = (1u << A) - 1u;
    const uint32_t rot = ((bits >> (xm + 1)) | (bits << (A - 1 - xm))) & maskA;   // bit k <-> seat (xm+1+k)%A
    if (!rot) return -1;
    const int seat = xm + 1 + (__ffs((int)rot) - 1);
    return seat >= A ? seat - A : seat;
}
// tanh rounded once from double: 1 - 2/(exp(2x)+1) (abs. error ~1e-16, far below the fp32 ulp)
__device__ __forceinline__ float tanh_rn(float x) {
#if PULSE_TANH_F32
    return tanhf(x);
#else
    const double e2 = exp(2.0 * (double)x);
    return (float)(1.0 - 2.0 / (e2 + 1.0));
#endif
}

// WOBS (LPT = 4, n_games % 16 == 0): the observation rows of a wavefront's 16 tables are one contiguous
// 16 x obs_size x 4 B block in HBM.  Written column by column they cost thirteen store instructions that each
// touch sixteen cache lines; with WOBS the lanes drop their values into the wavefront's LDS slice and the
// block leaves as three 1-KiB bursts (16 B per lane).  A wavefront's LDS operations retire in order, so no
// workgroup barrier is involved.
template <uint32_t PH, bool POLICY, int LPT, int SPL, bool WOBS>
__global__ __launch_bounds__(kBlock) void poker_step_kernel(const PulsePokerView v, int64_t* __restrict__ actions,
                                                           const int32_t* __restrict__ actor_idx_in,
                                                           float* __restrict__ rewards, const PolicyArgs pa) {
    static_assert(LPT * SPL >= 1 && LPT * SPL <= 16 * 16 && (LPT & (LPT - 1)) == 0, "bad table mapping");
    static_assert(!WOBS || LPT == 4, "observation staging is written for 4 lanes per table");
    extern __shared__ int4 smem4[];
    const int gt = blockIdx.x * kBlock + threadIdx.x;
    const int t = gt / LPT;
    const int j = gt % LPT;
#if PULSE_KERNARG_EARLY
    // Fetch the whole argument block before the first wait: left alone, the compiler sinks the scalar loads
    // of the ~40 view pointers next to their uses and the prologue pays five dependent scalar-cache trips.
    asm volatile("" :: "s"(v.pots), "s"(v.stages), "s"(v.deck_positions), "s"(v.button), "s"(v.idx), "s"(v.highest), "s"(v.agg),
                 "s"(v.acted), "s"(v.last_raise_size), "s"(v.is_done), "s"(v.equity_dirty), "s"(v.board));
    asm volatile("" :: "s"(v.stacks), "s"(v.current_round_bet), "s"(v.total_invested), "s"(v.status), "s"(v.hands), "s"(v.equities),
                 "s"(v.pre_board), "s"(v.w1), "s"(v.w2), "s"(v.K), "s"(v.alpha), "s"(actions));
#endif
    if (t >= v.n_games) return;   // whole lane groups leave together
    STAMP(0);
    const int P = v.n_players, A = v.active_players;
    const int32_t* __restrict__ hr = v.hand_ranks;
    const uint32_t hr_len = (uint32_t)v.hand_ranks_len;

    // ---- load (every load is independent: all in flight at once)
    const uint32_t ut = (uint32_t)t, so = ut * 4u, bo = ut * 20u;      // byte offsets into [N] int32 / board
    int idx = ldo(v.idx, so), button = ldo(v.button, so), pot = ldo(v.pots, so), stage = ldo(v.stages, so), dpos = ldo(v.deck_positions, so);
    int highest = ldo(v.highest, so), agg = ldo(v.agg, so), acted = ldo(v.acted, so), lrs = ldo(v.last_raise_size, so);
    bool done = ldo(v.is_done, ut) != 0;
    bool dirty = ldo(v.equity_dirty, ut) != 0;
    int b0 = ldo(v.board, bo), b1 = ldo(v.board, bo + 4), b2 = ldo(v.board, bo + 8), b3 = ldo(v.board, bo + 12), b4 = ldo(v.board, bo + 16);
    int stack[SPL], bet[SPL], inv[SPL], status[SPL], h0[SPL], h1[SPL];
    float eq[SPL];
    uint32_t ro[SPL];                                                  // byte offset of (table, seat) in an [N,P] int32 array
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int seat = j + LPT * k;
        ro[k] = (ut * (uint32_t)P + (uint32_t)seat) * 4u;
        stack[k] = 0; bet[k] = 0; inv[k] = 0; status[k] = PULSE_SITOUT; h0[k] = -1; h1[k] = -1; eq[k] = 0.5f;
        if (seat < P) {
            stack[k] = ldo(v.stacks, ro[k]); bet[k] = ldo(v.current_round_bet, ro[k]);
            inv[k] = ldo(v.total_invested, ro[k]); status[k] = ldo(v.status, ro[k]);
            const int2 h = ldo(reinterpret_cast<const int2*>(v.hands), ro[k] * 2u);
            h0[k] = h.x; h1[k] = h.y;
        }
        if (seat < A) eq[k] = ldo(v.equities, (ut * (uint32_t)A + (uint32_t)seat) * 4u);
    }
    long long act64 = 0;
    if ((PH & (PULSE_PH_EXECUTE | PULSE_PH_REWARD)) && actions) act64 = ldo(actions, ut * 8u);
    // the Philox draw of this (table, step) depends on no load: issue it now, it executes under the load latency
    U4 rnd{0, 0, 0, 0};
    if (POLICY) rnd = philox4x32(pa.seed, pa.table_id0 + (uint64_t)t, pa.step_counter);
    const float w1 = *v.w1, w2 = *v.w2;
    const int Kdiv = *v.K, alpha = *v.alpha;
    uint32_t pre_tag = 0;
    if ((PH & (PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN)) && v.pre_board) pre_tag = (uint32_t)ldo(v.pre_board, so);
    // next street's cards, fetched now so the dependent deck read overlaps the policy / betting logic
    int nx0 = 0, nx1 = 0, nx2 = 0;
    if ((PH & PULSE_PH_ADVANCE) && kPrefetchDeck) {
        const int32_t* dk = v.decks + (size_t)t * 52;
        nx0 = (uint32_t)(dpos + 1) < 52u ? dk[dpos + 1] : 0;
        nx1 = (uint32_t)(dpos + 2) < 52u ? dk[dpos + 2] : 0;
        nx2 = (uint32_t)(dpos + 3) < 52u ? dk[dpos + 3] : 0;
    }

    // values as loaded: only what a step changes is written back (a table's seat rows change in one or
    // two cells per step; writing all of them back doubles the store traffic)
    const int idx_in = idx, pot_in = pot, stage_in = stage, dpos_in = dpos, highest_in = highest, agg_in = agg,
              acted_in = acted, lrs_in = lrs;
    const bool dirty_in = dirty;
    int stack_in[SPL], bet_in[SPL], inv_in[SPL], status_in[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) { stack_in[k] = stack[k]; bet_in[k] = bet[k]; inv_in[k] = inv[k]; status_in[k] = status[k]; }
    const int b0_in = b0, b1_in = b1, b2_in = b2, b3_in = b3, b4_in = b4;

    // seat-set bitmask of a per-seat predicate / value of one seat, visible to every lane of the table
#define SEAT_BITS(expr) ([&]() { uint32_t m_ = 0; _Pragma("unroll") for (int k = 0; k < SPL; ++k) m_ |= (uint32_t)((expr) ? 1u : 0u) << (j + LPT * k); return grp_or<LPT>(m_); }())
#define SEAT_PICK(arr, seat_) ([&]() { uint32_t r_ = 0; _Pragma("unroll") for (int k = 0; k < SPL; ++k) r_ |= (j + LPT * k) == (seat_) ? (uint32_t)(arr)[k] : 0u; return (int)grp_or<LPT>(r_); }())

    STAMP(1);   // loads issued
    // ---- capture (PokerGPU.py:530-539)
    const bool prev_done = done;
    const int actor = ((PH & PULSE_PH_CAPTURE) || !actor_idx_in ? idx : actor_idx_in[t]) & 15;
    const int a_status = SEAT_PICK(status, actor), a_stack = SEAT_PICK(stack, actor), a_bet = SEAT_PICK(bet, actor);
    const bool has_legal_actor = a_status != PULSE_FOLDED && a_status != PULSE_ALLIN && a_status != PULSE_SITOUT && !prev_done;
    int prev_invested = a_bet;
    if (!(PH & PULSE_PH_CAPTURE)) prev_invested = ldo(v.prev_invested, so);
    const int prev_stack = a_stack;

    STAMP(2);   // first loads have arrived (actor values picked)
    // ---- scripted opponents (environments/Poker/utils.py:108-123), fused in front of the step
    if (POLICY) {
        const int type = (int)((pa.types_packed >> (4 * (idx & 15))) & 15u);
        if (type != PULSE_AGENT_EXTERNAL) {
            const int seat_i = idx & 15;
            act64 = scripted_action(type, SEAT_PICK(h0, seat_i), SEAT_PICK(h1, seat_i), pot, rnd);
            if (j == 0) sto(actions, ut * 8u, (int64_t)act64);
        }
    }
    const int action = act64 < -1 ? -1 : (act64 > 13 ? 13 : (int)act64);   // masks only test ==0, ==1, >=2, 2, 3..11, 12

    STAMP(3);   // policy done
    // ---- 1) equities of dirty tables (PokerGPU.py:455-525)
    if (PH & PULSE_PH_EQUITY) {
        if (dirty) {
            const int c5 = stage >= 2 ? b3 : 0, c6 = stage == 3 ? b4 : 0;
            const bool street = stage >= 1 && stage <= 3;
            const bool cached_board = street && board_matches(pre_tag, stage + 2, b0, b1, b2, b3, b4);
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int seat = j + LPT * k;
                float e = 0.5f;
                if (seat < A && street) {
                    bool hit = false;
                    if (cached_board) {       // both cache reads are independent single hops
                        const uint32_t ph = (uint32_t)ldo(v.pre_hands, ro[k]);
                        const float pe = ldo(v.pre_eq, ((ut * 3u + (uint32_t)(stage - 1)) * (uint32_t)P + (uint32_t)seat) * 4u);
                        hit = ph == pack_hand(h0[k], h1[k]) && card_ok(h0[k]) && card_ok(h1[k]);
                        e = pe;
                    }
                    if (!hit) {               // the reference's literal seven-gather chain
                        const float r = (float)walk7(hr, hr_len, h0[k], h1[k], b0, b1, b2, c5, c6);
                        e = stage == 1 ? __fdiv_rn(__fsub_rn(r, 74359.0f), 749420.0f) : __fdiv_rn(__fsub_rn(r, 4109.0f), 32765.0f);
                        e = fminf(fmaxf(e, 0.0f), 1.0f);
                    }
                }
                eq[k] = e;
                if (seat < A) sto(v.equities, (ut * (uint32_t)A + (uint32_t)seat) * 4u, e);
            }
            dirty = false;
        }
    }
    float e_actor;
    {
        uint32_t r_ = 0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) r_ |= (j + LPT * k) == actor ? __float_as_uint(eq[k]) : 0u;
        e_actor = __uint_as_float(grp_or<LPT>(r_));
        if (actor >= LPT * SPL) e_actor = 0.5f;
    }

    STAMP(4);   // equities done
    // ---- 2) execute the action of the seat to act (PokerGPU.py:230-303)
    if (PH & PULSE_PH_EXECUTE) {
        const int call_cost = highest - a_bet;
        const bool active = a_status != PULSE_FOLDED && a_status != PULSE_ALLIN && a_status != PULSE_SITOUT && !done;
        if (active && action >= 0) {
            int n_stack = a_stack, n_bet = a_bet, n_inv_add = 0, n_status = a_status;
            if (action == 0) {
                n_status = PULSE_FOLDED;
            } else {
                int raise_amt = 0;
                if (action == 2) raise_amt = lrs;
                else if (action == 12) raise_amt = a_stack;
                else if (action >= 3 && action <= 11) {
                    const float fr = action == 3 ? 0.25f : action == 4 ? 0.33f : action == 5 ? 0.50f : action == 6 ? 0.75f
                                   : action == 7 ? 1.00f : action == 8 ? 1.50f : action == 9 ? 2.00f : action == 10 ? 3.00f : 4.00f;
                    raise_amt = (int)__fmul_rn((float)pot, fr);
                }
                const int total = action == 1 ? call_cost : call_cost + raise_amt;
                const int amt = min(total, a_stack);
                const bool is_raise = action >= 2 && amt > call_cost;
                n_stack = a_stack - amt; n_bet = a_bet + amt; n_inv_add = amt; pot += amt;
                if (n_stack == 0) n_status = PULSE_ALLIN;
                if (is_raise) {
                    const int raise_size = n_bet - highest;
                    highest = n_bet;
                    if (raise_size >= lrs) { agg = idx; acted = 0; lrs = raise_size; }
                }
            }
            acted += 1;
#pragma unroll
            for (int k = 0; k < SPL; ++k)
                if (j + LPT * k == (idx & 15)) { stack[k] = n_stack; bet[k] = n_bet; inv[k] += n_inv_add; status[k] = n_status; }
        }
    }

    const uint32_t act_bits = SEAT_BITS(status[k] == PULSE_ACTIVE);
    const uint32_t cont_bits = SEAT_BITS(status[k] == PULSE_ACTIVE || status[k] == PULSE_ALLIN);
    const int contenders = __popc(cont_bits);

    STAMP(5);   // action executed, seat masks built
    // ---- 3) next actor, round close, street transition (PokerGPU.py:547-616)
    if (PH & PULSE_PH_ADVANCE) {
        const int truly_active = __popc(act_bits);
        const bool all_acted = acted >= truly_active;
        bool round_over = done || truly_active == 0;
        const uint32_t maskA = (1u << A) - 1u;
        const int next_seat = first_after_near(act_bits & maskA, idx, A);
        const bool has_next = next_seat >= 0;
        const bool closes = all_acted && (idx == agg || (has_next && next_seat == agg));
        round_over = round_over || !has_next || closes;
        if (!round_over && has_next) idx = next_seat;
        const bool early_term = contenders <= 1 && round_over;
        if (early_term) done = true;
        if (round_over && !early_term && !done) {
            lrs = 1; stage += 1; highest = 0; agg = mod_near(button + 1, A); acted = 0;
#pragma unroll
            for (int k = 0; k < SPL; ++k) bet[k] = 0;
            const int first = first_after_near(act_bits & maskA, button, A);
            if (first >= 0) idx = first;
            if (stage > 3) { done = true; stage = 4; }
            else {
                if (!kPrefetchDeck) {
                    const int32_t* dk = v.decks + (size_t)t * 52;
                    nx0 = (uint32_t)(dpos + 1) < 52u ? dk[dpos + 1] : 0;
                    if (stage == 1) {
                        nx1 = (uint32_t)(dpos + 2) < 52u ? dk[dpos + 2] : 0;
                        nx2 = (uint32_t)(dpos + 3) < 52u ? dk[dpos + 3] : 0;
                    }
                }
                if (stage == 1) { b0 = nx0; b1 = nx1; b2 = nx2; dpos += 4; }     // burn + flop (:601-604)
                else if (stage == 2) { b3 = nx0; dpos += 2; }                    // burn + turn (:607-610)
                else { b4 = nx0; dpos += 2; }                                    // burn + river (:613-616)
                dirty = true;
            }
        }
    }

    STAMP(6);   // advance / deal done
    // ---- 4) payouts on newly finished tables (PokerGPU.py:619-623)
    const bool newly_done = (PH & PULSE_PH_CAPTURE) ? (done && !prev_done) : done;
    if (PH & PULSE_PH_FOLDWIN) {                                            // :331-338
        if (newly_done && contenders == 1) {
            const int survivor = __ffs((int)cont_bits) - 1;
#pragma unroll
            for (int k = 0; k < SPL; ++k) if (j + LPT * k == survivor) stack[k] += pot;
            pot = 0;
        }
    }
    if (PH & PULSE_PH_SHOWDOWN) {                                           // :380-453
        if (newly_done && stage < 5 && contenders > 1) {
            const int32_t* dk = v.decks + (size_t)t * 52;
            if (stage == 0) {
                b0 = (uint32_t)(dpos + 1) < 52u ? dk[dpos + 1] : 0;
                b1 = (uint32_t)(dpos + 2) < 52u ? dk[dpos + 2] : 0;
                b2 = (uint32_t)(dpos + 3) < 52u ? dk[dpos + 3] : 0;
                b3 = (uint32_t)(dpos + 5) < 52u ? dk[dpos + 5] : 0;
                b4 = (uint32_t)(dpos + 7) < 52u ? dk[dpos + 7] : 0;
                dpos += 8;
            } else if (stage == 1) {
                b3 = (uint32_t)(dpos + 1) < 52u ? dk[dpos + 1] : 0;
                b4 = (uint32_t)(dpos + 3) < 52u ? dk[dpos + 3] : 0;
                dpos += 4;
            } else if (stage == 2) {
                b4 = (uint32_t)(dpos + 1) < 52u ? dk[dpos + 1] : 0;
                dpos += 2;
            }
            bool eligible[SPL]; int rank[SPL], payout[SPL];
            const bool cached_board = board_matches(pre_tag, 5, b0, b1, b2, b3, b4);
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int seat = j + LPT * k;
                eligible[k] = seat < A && (status[k] == PULSE_ACTIVE || status[k] == PULSE_ALLIN);
                rank[k] = INT_MIN; payout[k] = 0;
                if (eligible[k]) {
                    bool hit = false;
                    if (cached_board) {
                        const uint32_t ph = (uint32_t)ldo(v.pre_hands, ro[k]);
                        const int pr = ldo(v.pre_rank, ro[k]);
                        hit = ph == pack_hand(h0[k], h1[k]) && card_ok(h0[k]) && card_ok(h1[k]);
                        rank[k] = pr;
                    }
                    if (!hit) rank[k] = walk7(hr, hr_len, h0[k], h1[k], b0, b1, b2, b3, b4);
                }
            }
            // side pots, one layer per distinct commitment level (PokerGPU.py:340-378)
            int prev_level = 0;
            for (int l = 0; l < A; ++l) {
                int lv = INT_MAX;
#pragma unroll
                for (int k = 0; k < SPL; ++k) if ((j + LPT * k) < A && inv[k] > prev_level) lv = min(lv, inv[k]);
                const int level = grp_imin<LPT>(lv);
                if (level == INT_MAX) break;
                const int n_contrib = __popc(SEAT_BITS((j + LPT * k) < A && inv[k] >= level));
                int bl = INT_MIN;
#pragma unroll
                for (int k = 0; k < SPL; ++k) if ((j + LPT * k) < A && inv[k] >= level && eligible[k]) bl = max(bl, rank[k]);
                const int best = grp_imax<LPT>(bl);
                const uint32_t win_bits = SEAT_BITS((j + LPT * k) < A && inv[k] >= level && eligible[k] && rank[k] == best);
                const int n_win = __popc(win_bits);
                if (n_win > 0) {
                    const int layer_pot = (level - prev_level) * n_contrib;
                    const int share = layer_pot / n_win, rem = layer_pot - share * n_win;
                    const int first_win = __ffs((int)win_bits) - 1;
#pragma unroll
                    for (int k = 0; k < SPL; ++k)
                        if ((win_bits >> (j + LPT * k)) & 1u) payout[k] += share + ((j + LPT * k) == first_win ? rem : 0);
                }
                prev_level = level;
            }
#pragma unroll
            for (int k = 0; k < SPL; ++k) stack[k] += payout[k];
            pot = 0; stage = 5;
        }
    }
    if (PH & PULSE_PH_CLEARDONE) {                                          // :625-628
        if (done) {
            highest = 0;
#pragma unroll
            for (int k = 0; k < SPL; ++k) { bet[k] = 0; inv[k] = 0; }
        }
    }

    STAMP(7);   // payouts done
    // ---- 5) shaped reward (PokerGPU.py:305-329, :631-632)
    if (PH & PULSE_PH_REWARD) {
        const float cnt = (float)contenders;
        const float fair = __fdiv_rn(1.0f, fmaxf(cnt, 1.0f));
        const int cc = max(0, highest - prev_invested);
        const float potf = (float)pot;
        const float m = __fmul_rn(e_actor, potf);
        const float o = __fdiv_rn((float)cc, __fadd_rn((float)(pot + cc), 1e-6f));
        float sv = 0.0f;
        if (action == 1) sv = __fmul_rn(__fsub_rn(e_actor, o), potf);
        else if (action == 0) sv = __fmul_rn(__fsub_rn(o, e_actor), potf);
        else if (action >= 2) sv = __fmul_rn(__fsub_rn(e_actor, fair), potf);
        const float x = __fdiv_rn(__fadd_rn(__fmul_rn(w1, m), __fmul_rn(w2, sv)), (float)Kdiv);
        float r = __fmul_rn((float)alpha, tanh_rn(x));
        if ((PH & PULSE_PH_CAPTURE) && (!has_legal_actor || prev_done)) r = 0.0f;
        if (j == 0) sto(rewards, so, r);
    }

    STAMP(8);   // reward done
    // ---- 6) observation for the next seat to act (PokerGPU.py:159-179)
    if (PH & PULSE_PH_OBS) {
        const int wlane = threadIdx.x & 63;
        float* const l_obs = reinterpret_cast<float*>(smem4) + (threadIdx.x >> 6) * 16 * v.obs_size;
        float* __restrict__ o = WOBS ? l_obs + (wlane >> 2) * v.obs_size
                                     : reinterpret_cast<float*>(reinterpret_cast<char*>(v.obs) + ut * (uint32_t)v.obs_size * 4u);
        const int seat_i = idx & 15;
        const int n_h0 = SEAT_PICK(h0, seat_i), n_h1 = SEAT_PICK(h1, seat_i);
        const int n_stack = SEAT_PICK(stack, seat_i), n_status = SEAT_PICK(status, seat_i), n_bet = SEAT_PICK(bet, seat_i);
        const int idxm = mod_near(idx, A);
        const int pos = mod_near(idx - button, A);
        if (LPT == 4) {
            // columns 0..12, four per lane-quad pass: lane j writes column 4*pass + j (a 4-way select per pass)
            const int h0v = j == 0 ? b0 : j == 1 ? b1 : j == 2 ? b2 : b3;
            const int h1v = j == 0 ? b4 : j == 1 ? n_h0 : j == 2 ? n_h1 : stage;
            const int h2v = j == 0 ? pos : j == 1 ? pot : j == 2 ? highest - n_bet : n_stack;
            o[j] = (float)h0v; o[4 + j] = (float)h1v; o[8 + j] = (float)h2v;
            if (j == 0) o[12] = (float)n_status;
        } else {
#pragma unroll
            for (int c0 = 0; c0 < 13; c0 += LPT) {
                const int c = c0 + j;
                if (c < 13) {
                    int hv;
                    switch (c) {
                    case 0: hv = b0; break; case 1: hv = b1; break; case 2: hv = b2; break; case 3: hv = b3; break;
                    case 4: hv = b4; break; case 5: hv = n_h0; break; case 6: hv = n_h1; break; case 7: hv = stage; break;
                    case 8: hv = pos; break; case 9: hv = pot; break; case 10: hv = highest - n_bet; break;
                    case 11: hv = n_stack; break; default: hv = n_status; break;
                    }
                    o[c] = (float)hv;
                }
            }
        }
        // opponents: seat (idx+1+k)%A -> columns 13+3k..; seats >= A zero-fill the padding slots
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int seat = j + LPT * k;
            if (seat < v.max_players && seat != idxm) {
                int slot; float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
                if (seat < A) { slot = seat - idxm - 1; if (slot < 0) slot += A; f0 = (float)stack[k]; f1 = (float)status[k]; f2 = (float)bet[k]; }
                else slot = seat - 1;
                float* dst = o + 13 + 3 * slot;
                dst[0] = f0; dst[1] = f1; dst[2] = f2;
            }
        }
    }

    if (WOBS && (PH & PULSE_PH_OBS)) {
        const int wlane = threadIdx.x & 63;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n4 = 4 * v.obs_size;                                   // int4 per wavefront block
        const int tw0 = (int)((blockIdx.x * kBlock + threadIdx.x) >> 6) << 4;   // first table of this wavefront
        int4* dst = reinterpret_cast<int4*>(v.obs + (size_t)tw0 * v.obs_size);
        const int4* src = reinterpret_cast<const int4*>(smem4) + (threadIdx.x >> 6) * n4;
        for (int e = wlane; e < n4; e += 64) dst[e] = src[e];
    }
    STAMP(9);   // observation stores issued
    // ---- store (changed records only: a seat's four cells and the per-table scalars change together, so
    //      one test per seat / per table guards each group of stores instead of one branch per word)
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int seat = j + LPT * k;
        const bool changed = (stack[k] != stack_in[k]) | (bet[k] != bet_in[k]) | (inv[k] != inv_in[k]) | (status[k] != status_in[k]);
        if (seat < P && changed) {
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN)) sto(v.stacks, ro[k], stack[k]);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE | PULSE_PH_CLEARDONE)) sto(v.current_round_bet, ro[k], bet[k]);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_CLEARDONE)) sto(v.total_invested, ro[k], inv[k]);
            if (PH & PULSE_PH_EXECUTE) sto(v.status, ro[k], status[k]);
        }
    }
    if (PH & (PULSE_PH_ADVANCE | PULSE_PH_SHOWDOWN)) {
        const bool board_changed = (b0 != b0_in) | (b1 != b1_in) | (b2 != b2_in) | (b3 != b3_in) | (b4 != b4_in);
        if (board_changed) {
#pragma unroll
            for (int c0 = 0; c0 < 5; c0 += LPT) {
                const int c = c0 + j;
                if (c < 5) sto(v.board, bo + (uint32_t)c * 4u, c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : c == 3 ? b3 : b4);
            }
        }
    }
    if (j == 0) {
        if (PH & PULSE_PH_CAPTURE) { sto(v.prev_stacks, so, prev_stack); sto(v.prev_invested, so, prev_invested); }
        const bool betting_changed = (pot != pot_in) | (highest != highest_in) | (agg != agg_in) | (acted != acted_in) |
                                     (lrs != lrs_in) | (idx != idx_in);
        if (betting_changed) {
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN)) sto(v.pots, so, pot);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE | PULSE_PH_CLEARDONE)) sto(v.highest, so, highest);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE)) { sto(v.agg, so, agg); sto(v.acted, so, acted); sto(v.last_raise_size, so, lrs); }
            if (PH & PULSE_PH_ADVANCE) sto(v.idx, so, idx);
        }
        const bool street_changed = (stage != stage_in) | (dpos != dpos_in) | (dirty != dirty_in);
        if (street_changed) {
            if (PH & (PULSE_PH_ADVANCE | PULSE_PH_SHOWDOWN)) { sto(v.stages, so, stage); sto(v.deck_positions, so, dpos); }
            if (PH & (PULSE_PH_EQUITY | PULSE_PH_ADVANCE)) sto(v.equity_dirty, ut, (uint8_t)(dirty ? 1 : 0));
        }
        if (PH & PULSE_PH_ADVANCE) sto(v.is_done_out, ut, (uint8_t)(done ? 1 : 0));      // ping-pong buffer: always written
    }
    if (POLICY && pa.wave_done) {
        // the roll-out's stop rule (trainGPU.py:27-33) rides on the chunk's last launch: every wavefront stores how many
        // of its tables are done -- a plain store, summed on the host after an asynchronous copy (atomics onto shared
        // counters cost this launch as much as the separate counting kernel they would replace)
        const int c = __popcll(__ballot(done && j == 0));
        if ((threadIdx.x & 63) == 0) pa.wave_done[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = (uint32_t)c;
    }
    STAMP(10);  // state stores issued
#if PULSE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);                      // vmcnt(0): all stores acknowledged
    STAMP(11);
#endif
#undef SEAT_BITS
#undef SEAT_PICK
}

// ---------------------------------------------------------------- standalone policy (build_actions)
__global__ __launch_bounds__(kBlock) void poker_policy_kernel(const float* __restrict__ obs, int obs_stride,
                                                             const int32_t* __restrict__ seat_idx, int n,
                                                             PolicyArgs pa, int64_t* __restrict__ actions) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const int seat = seat_idx[t];
    const int type = (int)((pa.types_packed >> (4 * (seat & 15))) & 15u);
    if (type == PULSE_AGENT_EXTERNAL) return;
    const float* o = obs + (size_t)t * obs_stride;
    const U4 rnd = philox4x32(pa.seed, pa.table_id0 + (uint64_t)t, pa.step_counter);
    actions[t] = scripted_action(type, (int)o[5], (int)o[6], (int)o[9], rnd);
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

// ---------------------------------------------------------------- reset (PokerGPU.py:73-157)
// shuffle keys keep their top `shuffle_key_bits` bits (0 = all 32; fewer bits force ties, for the tests)
__device__ __forceinline__ int key_shift(const PulsePokerResetOpts& o) {
    return (o.shuffle_key_bits > 0 && o.shuffle_key_bits < 32) ? 32 - o.shuffle_key_bits : 0;
}

template <bool SHUFFLE>
__global__ __launch_bounds__(kBlock) void poker_reset_kernel(const PulsePokerView v, const PulsePokerResetOpts o) {
    __shared__ uint32_t keys[kBlock / kLanes][52];
    __shared__ int32_t deck_s[kBlock / kLanes][52];
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
        if (s < 13) {
            const U4 r = philox4x32(o.seed, o.table_id0 + (uint64_t)t, o.episode * 16 + (uint64_t)s);
            const int ks = key_shift(o);
            keys[g][4 * s + 0] = r.x >> ks; keys[g][4 * s + 1] = r.y >> ks; keys[g][4 * s + 2] = r.z >> ks; keys[g][4 * s + 3] = r.w >> ks;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // card c (0..51) lands at position #{keys < key[c]} (ties by index): deck[pos] = c + 1.  Without a tie
        // the strict count alone is the position (one compare + one add-with-carry per pair); with one, the
        // table's strict counts sum to less than 0 + 1 + ... + 51 and the wavefront (rarely: ~3e-7 per table)
        // recounts with the index tie-break.
        uint32_t kc[4] = {0, 0, 0, 0}; int pos[4] = {0, 0, 0, 0};
        if (s < 13) {
#pragma unroll
            for (int q = 0; q < 4; ++q) kc[q] = keys[g][4 * s + q];
            for (int j = 0; j < 52; ++j) {
                const uint32_t kj = keys[g][j];
#pragma unroll
                for (int q = 0; q < 4; ++q) pos[q] += kj < kc[q];
            }
        }
        const int psum = grp_sum<kLanes>(pos[0] + pos[1] + pos[2] + pos[3]);
        if (__any(psum != 1326)) {
            if (s < 13) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pos[q] = 0;
                for (int j = 0; j < 52; ++j) {
                    const uint32_t kj = keys[g][j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pos[q] += (kj < kc[q]) || (kj == kc[q] && j < 4 * s + q);
                }
            }
        }
        if (s < 13) {
#pragma unroll
            for (int q = 0; q < 4; ++q) deck_s[g][pos[q]] = 4 * s + q + 1;
        }
    } else {
        const int32_t* src = o.prefixed_decks + (size_t)t * 52;
        for (int c = s; c < 52; c += kLanes) deck_s[g][c] = src[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int c = s; c < 52; c += kLanes) dk[c] = deck_s[g][c];

    // stacks: refill busted / over-max, then torch.roll by `rotation` (:101-110)
    int st = o.starting_bbs;
    if (seat && !o.first) {
        const int src = pymod(s - o.rotation, P);
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
        for (int x = 1; x < kLanes; x <<= 1) { m |= __shfl_xor(m, x, kLanes); bad |= (uint32_t)__shfl_xor((int)bad, x, kLanes); }
        const bool board_ok = card_ok(f0) && card_ok(f1) && card_ok(f2) && card_ok(f3) && card_ok(f4);
        m |= (1ull << (f0 & 63)) | (1ull << (f1 & 63)) | (1ull << (f2 & 63)) | (1ull << (f3 & 63)) | (1ull << (f4 & 63));
        const bool valid = !bad && board_ok && __popcll(m) == 2 * A + 5;    // 2A+5 distinct cards in 1..52
        const int32_t* __restrict__ hr = v.hand_ranks;
        const uint32_t hr_len = (uint32_t)v.hand_ranks_len;
        if (valid && inA) {
            int p3 = 53;
            p3 = hr_at(hr, hr_len, p3 + f0); p3 = hr_at(hr, hr_len, p3 + f1); p3 = hr_at(hr, hr_len, p3 + f2);
            const int p4 = hr_at(hr, hr_len, p3 + f3);
            const int p5 = hr_at(hr, hr_len, p4 + f4);
            // three independent chains, issued level by level (unconditional loads: no branch between them)
            const int a1 = hr_at_nb(hr, hr_len, p3 + h0), b1 = hr_at_nb(hr, hr_len, p4 + h0), c1 = hr_at_nb(hr, hr_len, p5 + h0);
            const int q5 = hr_at_nb(hr, hr_len, a1 + h1), q6 = hr_at_nb(hr, hr_len, b1 + h1), r7 = hr_at_nb(hr, hr_len, c1 + h1);
            const int a3 = hr_at_nb(hr, hr_len, q5);
            const float vt = (float)hr_at_nb(hr, hr_len, q6);                               // :500
            const float vf = (float)hr_at_nb(hr, hr_len, a3);                               // :521
            float ef = __fdiv_rn(__fsub_rn(vf, 74359.0f), 749420.0f);                       // :523
            float et = __fdiv_rn(__fsub_rn(vt, 4109.0f), 32765.0f);                         // :502
            float er = __fdiv_rn(__fsub_rn((float)r7, 4109.0f), 32765.0f);                  // :481
            ef = fminf(fmaxf(ef, 0.0f), 1.0f); et = fminf(fmaxf(et, 0.0f), 1.0f); er = fminf(fmaxf(er, 0.0f), 1.0f);
            float* pe = v.pre_eq + (size_t)t * 3 * P;
            pe[s] = ef; pe[P + s] = et; pe[2 * P + s] = er;
            v.pre_rank[row] = r7;
        }
        if (seat) v.pre_hands[row] = (valid && inA) ? (int32_t)pack_hand(h0, h1) : 0;
        if (s == 0) v.pre_board[t] = valid ? (int32_t)(pack_board(f0, f1, f2, f3, f4) | kPreBoardValid) : 0;
    }
    const int button = o.first ? 0 : pymod(v.button[t] + 1, A);                         // :121
    int sb, bb, idx;
    if (A == 2) { sb = button; bb = pymod(button + 1, A); idx = button; }               // :123-125,:131
    else { sb = pymod(button + 1, A); bb = pymod(button + 2, A); idx = pymod(bb + 1, A); }
    int bet = 0, inv = 0, status = inA ? PULSE_ACTIVE : PULSE_SITOUT;
    if (s == bb) { st -= 1; bet = 1; inv = 1; status = st == 0 ? PULSE_ALLIN : PULSE_ACTIVE; }   // :188-199
    if (seat) {
        v.stacks[row] = st; v.current_round_bet[row] = bet; v.total_invested[row] = inv; v.status[row] = status;
        *reinterpret_cast<int2*>(v.hands + row * 2) = make_int2(h0, h1);
    }
    if (inA) v.equities[(size_t)t * A + s] = 0.5f;                                      // :144
    if (s < 5) v.board[t * 5 + s] = -1;                                                 // :95
    if (s == 0) {
        v.last_raise_size[t] = 1; v.deck_positions[t] = 2 * A; v.pots[t] = 1; v.stages[t] = 0;
        v.button[t] = button; v.sb[t] = sb; v.bb[t] = bb; v.idx[t] = idx;
        v.highest[t] = 1; v.agg[t] = bb; v.acted[t] = 0; v.is_done[t] = 0; v.is_done_out[t] = 0;
        v.equity_dirty[t] = 1; v.prev_stacks[t] = 0; v.prev_invested[t] = 0;
    }
    // first observation (:157 -> :159-179)
    float* __restrict__ ob = v.obs + (size_t)t * v.obs_size;
    const int a_h0 = grp_bcast(h0, idx), a_h1 = grp_bcast(h1, idx);
    const int a_stack = grp_bcast(st, idx), a_status = grp_bcast(status, idx), a_bet = grp_bcast(bet, idx);
    if (s < 13) {
        int hv;
        switch (s) {
        case 0: case 1: case 2: case 3: case 4: hv = -1; break;
        case 5: hv = a_h0; break; case 6: hv = a_h1; break; case 7: hv = 0; break;
        case 8: hv = pymod(idx - button, A); break; case 9: hv = 1; break; case 10: hv = 1 - a_bet; break;
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
}

// ---------------------------------------------------------------- episode statistics
__global__ __launch_bounds__(kBlock) void poker_stats_kernel(const uint8_t* __restrict__ is_done,
                                                            const float* __restrict__ rewards,
                                                            const uint8_t* __restrict__ mask, int n,
                                                            unsigned long long* stats, double* fstats) {
    int cnt = 0; double sum = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
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
                                                                   unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long h[PULSE_MAX_SEATS * 5 * 4];
    for (int i = threadIdx.x; i < PULSE_MAX_SEATS * 5 * 4; i += kBlock) h[i] = 0ull;
    __syncthreads();
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < n; t += gridDim.x * kBlock) {
        if (!dones[t] || (terminated_before && terminated_before[t])) continue;
        const long long delta = (long long)stacks[(size_t)t * n_players + q_seat] - (long long)initial_q_stacks[t];
        const int pos = pymod(q_seat - button[t], active_players);
        const int st = stages[t], bucket = st >= 4 ? 4 : max(st, 0);
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

// ---------------------------------------------------------------- host side
int check_view(const PulsePokerView* v, const char* who) {
    if (!v) return pulse::fail(PULSE_EINVAL, "null PulsePokerView");
    if (v->n_games < 0 || v->n_players < 2 || v->n_players > PULSE_MAX_SEATS || v->max_players > PULSE_MAX_SEATS ||
        v->max_players < v->n_players || v->active_players < 2 || v->active_players > v->n_players ||
        v->obs_size != 13 + 3 * (v->max_players - 1))
        return pulse::fail(PULSE_EINVAL, "PulsePokerView: unsupported shape (need 2 <= active <= n_players <= max_players <= 16, obs_size = 13+3*(max_players-1))");
    const void* ptrs[] = {v->hand_ranks, v->pots, v->stages, v->deck_positions, v->button, v->sb, v->bb, v->idx, v->highest,
                          v->agg, v->acted, v->last_raise_size, v->prev_stacks, v->prev_invested, v->is_done, v->is_done_out,
                          v->equity_dirty, v->stacks, v->current_round_bet, v->total_invested, v->status, v->hands, v->board,
                          v->decks, v->equities, v->obs, v->w1, v->w2, v->K, v->alpha};
    for (const void* p : ptrs)
        if (!p) return pulse::fail(PULSE_EINVAL, "PulsePokerView: null device pointer");
    if (v->hand_ranks_len <= 53) return pulse::fail(PULSE_EINVAL, "PulsePokerView: hand_ranks_len too small");
    (void)who;
    return 0;
}

inline int grid_for_tables(int n) { return (int)(((long long)n * kLanes + kBlock - 1) / kBlock); }

int finish_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pulse::fail_hip((int)e, what);
    return 0;
}

uint64_t pack_types(const uint8_t* agent_types, int n_players) {
    uint64_t packed = 0;
    for (int i = 0; i < n_players && i < 16; ++i) packed |= (uint64_t)(agent_types[i] & 15u) << (4 * i);
    return packed;
}

// Table -> lane mapping chosen per launch: PULSE_LPT env (1, 4, 16) overrides; default 4 lanes per table.
int g_lpt = 0;
int lanes_per_table() {
    if (!g_lpt) {
        const char* e = getenv("PULSE_LPT");
        int x = e ? atoi(e) : 4;
        g_lpt = (x == 1 || x == 2 || x == 4 || x == 16) ? x : 4;
    }
    return g_lpt;
}

int g_wobs = -1;
bool obs_staging_enabled() {
    if (g_wobs < 0) { const char* e = getenv("PULSE_WOBS"); g_wobs = (e && atoi(e) == 0) ? 0 : 1; }
    return g_wobs != 0;
}

template <uint32_t PH, bool POLICY>
void launch_step(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, const PolicyArgs& pa,
                 hipStream_t st) {
    const int lpt = lanes_per_table();
    const dim3 grid((unsigned)(((long long)v.n_games * lpt + kBlock - 1) / kBlock)), block(kBlock);
    const int spl = (v.max_players + lpt - 1) / lpt;   // seats per lane needed to cover max_players (obs padding too)
    if (lpt == 16) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 16, 1, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
    else if (lpt == 1) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 1, 16, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
    else if (lpt == 2 && spl <= 5) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 2, 5, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
    else if (lpt == 2) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 2, 8, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
    else if (PH == PULSE_PH_STEP && obs_staging_enabled() && (v.n_games & 15) == 0 && ((uintptr_t)v.obs & 15u) == 0) {
        const size_t lds = sizeof(float) * (size_t)(kBlock / 64) * 16 * (size_t)v.obs_size;
        if (spl <= 3) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 4, 3, PH == PULSE_PH_STEP>), grid, block, lds, st, v, actions, actor_idx, rewards, pa);
        else hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 4, 4, PH == PULSE_PH_STEP>), grid, block, lds, st, v, actions, actor_idx, rewards, pa);
    }
    else if (spl <= 3) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 4, 3, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
    else hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, 4, 4, false>), grid, block, 0, st, v, actions, actor_idx, rewards, pa);
}

template <uint32_t PH>
void launch_phase(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, hipStream_t st) {
    launch_step<PH, false>(v, actions, actor_idx, rewards, PolicyArgs{0, 0, 0, 0}, st);
}

}  // namespace

extern "C" {

int pulse_poker_step(const PulsePokerView* v, const int64_t* actions, float* rewards, void* stream) {
    if (int rc = check_view(v, "pulse_poker_step")) return rc;
    if (!actions || !rewards) return pulse::fail(PULSE_EINVAL, "pulse_poker_step: null actions/rewards");
    if (v->n_games == 0) return 0;
    launch_phase<PULSE_PH_STEP>(*v, const_cast<int64_t*>(actions), nullptr, rewards, (hipStream_t)stream);
    return finish_launch("pulse_poker_step");
}

int pulse_poker_policy_step(const PulsePokerView* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                            uint64_t table_id0, int64_t* actions, float* rewards, void* stream) {
    if (int rc = check_view(v, "pulse_poker_policy_step")) return rc;
    if (!actions || !rewards || !agent_types) return pulse::fail(PULSE_EINVAL, "pulse_poker_policy_step: null argument");
    if (v->n_games == 0) return 0;
    const PolicyArgs pa{pack_types(agent_types, v->n_players), seed, step_counter, table_id0};
    launch_step<PULSE_PH_STEP, true>(*v, actions, nullptr, rewards, pa, (hipStream_t)stream);
    return finish_launch("pulse_poker_policy_step");
}

int pulse_poker_phases(const PulsePokerView* v, uint32_t phases, const int64_t* actions, const int32_t* actor_idx,
                       float* rewards, void* stream) {
    if (int rc = check_view(v, "pulse_poker_phases")) return rc;
    if ((phases & (PULSE_PH_EXECUTE | PULSE_PH_REWARD)) && !actions)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_phases: actions required for EXECUTE/REWARD");
    if ((phases & PULSE_PH_REWARD) && !rewards) return pulse::fail(PULSE_EINVAL, "pulse_poker_phases: rewards required for REWARD");
    if (v->n_games == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int64_t* a = const_cast<int64_t*>(actions);
    switch (phases) {
    case PULSE_PH_STEP: launch_phase<PULSE_PH_STEP>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_STEP & ~PULSE_PH_EQUITY: launch_phase<(PULSE_PH_STEP & ~PULSE_PH_EQUITY)>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_EQUITY: launch_phase<PULSE_PH_EQUITY>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_EXECUTE: launch_phase<PULSE_PH_EXECUTE>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_ADVANCE: launch_phase<PULSE_PH_ADVANCE>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_FOLDWIN: launch_phase<PULSE_PH_FOLDWIN>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_SHOWDOWN: launch_phase<PULSE_PH_SHOWDOWN>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN: launch_phase<(PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN)>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_CLEARDONE: launch_phase<PULSE_PH_CLEARDONE>(*v, a, nullptr, rewards, st); break;
    case PULSE_PH_REWARD: launch_phase<PULSE_PH_REWARD>(*v, a, actor_idx, rewards, st); break;
    case PULSE_PH_OBS: launch_phase<PULSE_PH_OBS>(*v, a, nullptr, rewards, st); break;
    default: return pulse::fail(PULSE_EINVAL, "pulse_poker_phases: unsupported phase combination");
    }
    return finish_launch("pulse_poker_phases");
}

/* Diagnostic (tools/ablate_step.py): the fused policy+step with some phases compiled out, 4 lanes per
 * table.  Results are NOT a valid transition; used only to price phases. */
int pulse_poker_ablate(const PulsePokerView* v, uint32_t phases, int64_t* actions, float* rewards, uint64_t types_packed,
                       uint64_t step_counter, void* stream) {
    if (int rc = check_view(v, "pulse_poker_ablate")) return rc;
    const dim3 grid((unsigned)(((long long)v->n_games * 4 + kBlock - 1) / kBlock)), block(kBlock);
    const PolicyArgs pa{types_packed, 1, step_counter, 0};
    hipStream_t st = (hipStream_t)stream;
#define PULSE_ABL(MASK) case (MASK): hipLaunchKernelGGL((poker_step_kernel<(MASK), true, 4, 3, false>), grid, block, 0, st, *v, actions, (const int32_t*)nullptr, rewards, pa); break;
    switch (phases) {
    PULSE_ABL(PULSE_PH_STEP)
    PULSE_ABL(PULSE_PH_STEP & ~PULSE_PH_EQUITY)
    PULSE_ABL(PULSE_PH_STEP & ~PULSE_PH_SHOWDOWN)
    PULSE_ABL(PULSE_PH_STEP & ~(PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN))
    PULSE_ABL(PULSE_PH_STEP & ~PULSE_PH_REWARD)
    PULSE_ABL(PULSE_PH_STEP & ~PULSE_PH_OBS)
    PULSE_ABL(PULSE_PH_STEP & ~(PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN | PULSE_PH_REWARD))
    PULSE_ABL(PULSE_PH_STEP & ~(PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN | PULSE_PH_REWARD | PULSE_PH_OBS))
    PULSE_ABL(PULSE_PH_CAPTURE)
    default: return pulse::fail(PULSE_EINVAL, "pulse_poker_ablate: mask not instantiated");
    }
#undef PULSE_ABL
    return finish_launch("pulse_poker_ablate");
}

int pulse_calib_stream(int32_t* buf, uint64_t n_words, int32_t write, void* stream) {
    if (!buf || n_words == 0) return pulse::fail(PULSE_EINVAL, "pulse_calib_stream: bad argument");
    const dim3 grid(2048), block(kBlock);
    if (write) hipLaunchKernelGGL(calib_write_kernel, grid, block, 0, (hipStream_t)stream, buf, (size_t)n_words, 7);
    else hipLaunchKernelGGL(calib_read_kernel, grid, block, 0, (hipStream_t)stream, buf, (size_t)n_words, buf);
    return finish_launch("pulse_calib_stream");
}

#if PULSE_STAMPS
int pulse_debug_set_stamp_buffer(unsigned long long* buf) {
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_debug_set_stamp_buffer");
}
#endif

int pulse_poker_reset(const PulsePokerView* v, const PulsePokerResetOpts* o, void* stream) {
    if (int rc = check_view(v, "pulse_poker_reset")) return rc;
    if (!o || !o->decks_out) return pulse::fail(PULSE_EINVAL, "pulse_poker_reset: null options / decks_out");
    if (v->n_games == 0) return 0;
    const dim3 grid(grid_for_tables(v->n_games)), block(kBlock);
    if (o->prefixed_decks) hipLaunchKernelGGL((poker_reset_kernel<false>), grid, block, 0, (hipStream_t)stream, *v, *o);
    else hipLaunchKernelGGL((poker_reset_kernel<true>), grid, block, 0, (hipStream_t)stream, *v, *o);
    return finish_launch("pulse_poker_reset");
}

int pulse_poker_policy(const float* obs, int32_t obs_stride, const int32_t* seat_idx, int32_t n, const uint8_t* agent_types,
                       int32_t n_players, uint64_t seed, uint64_t step_counter, uint64_t table_id0, int64_t* actions,
                       void* stream) {
    if (!obs || !seat_idx || !agent_types || !actions || n < 0 || obs_stride < 10 || n_players < 1 || n_players > 16)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_policy: bad argument");
    if (n == 0) return 0;
    const PolicyArgs pa{pack_types(agent_types, n_players), seed, step_counter, table_id0};
    hipLaunchKernelGGL(poker_policy_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, obs,
                       obs_stride, seat_idx, n, pa, actions);
    return finish_launch("pulse_poker_policy");
}

int pulse_poker_eval_hands(const int32_t* hand_ranks, int32_t hand_ranks_len, const int32_t* cards, int32_t n_hands,
                           int32_t n_cards, int32_t flop_double, int32_t* out, void* stream) {
    if (!hand_ranks || !cards || !out || n_hands < 0 || n_cards < 5 || n_cards > 7 || hand_ranks_len <= 53)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_eval_hands: bad argument");
    if (n_hands == 0) return 0;
    hipLaunchKernelGGL(poker_eval_kernel, dim3((n_hands + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       hand_ranks, (uint32_t)hand_ranks_len, cards, n_hands, n_cards, flop_double, out);
    return finish_launch("pulse_poker_eval_hands");
}

/* ---- roll-out: n_steps fused policy+step launches enqueued back to back from native code ---------- */
namespace {
constexpr int kMaxTimed = 4096;
hipEvent_t g_ev_start[kMaxTimed], g_ev_stop[kMaxTimed];
int g_ev_created = 0, g_ev_used = 0;
int g_ev_launches[kMaxTimed];
long long g_rollout_calls = 0;
}

// ---- stop rule without a host sync (scripts/Poker/trainGPU.py:27-33), native side -------------------------
// After a chunk of steps: count the finished tables on the device (cumulative counter: no memset in the loop), copy
// the count to pinned host memory on a side stream, decide on the newest count that has ALREADY arrived.  Done
// here rather than in Python because the chunk boundary -- a kernel launch, two event records, a stream wait and
// an async copy -- cost more host time through torch (~50 us) than the five step launches of the chunk take on
// the GPU, which left the GPU idle a quarter of the time.
struct PulseStopRule {
    hipStream_t side;
    hipEvent_t ready[2], copied[2];
    unsigned long long* counts_dev;        // [2] cumulative per slot (counting kernel, pulse_stoprule_submit)
    unsigned long long* counts_host;       // [2] pinned
    uint32_t* waves_dev;                   // [2][max_waves] per-wavefront done counts written by the roll-out's last launch
    uint32_t* waves_host;                  // [2][max_waves] pinned
    int max_waves; int waves_used[2];      // waves_used[slot] > 0: the slot's verdict comes from the wave counts
    unsigned long long seen[2];
    long long pending[2]; int n_pending;
    long long chunk;
    int n; double threshold; bool late_over;
};

namespace {
bool stoprule_pop(PulseStopRule* h) {
    const long long c = h->pending[0];
    h->pending[0] = h->pending[1]; --h->n_pending;
    const int slot = (int)(c & 1);
    unsigned long long n_done = 0;
    if (h->waves_used[slot] > 0) {
        const uint32_t* w = h->waves_host + (size_t)slot * h->max_waves;
        for (int i = 0; i < h->waves_used[slot]; ++i) n_done += w[i];
    } else {
        const unsigned long long total = h->counts_host[slot];
        n_done = total - h->seen[slot];
        h->seen[slot] = total;
    }
    return (double)n_done > h->threshold * (double)h->n;
}
}  // namespace

int pulse_stoprule_create(int32_t n_tables, double threshold, void** out) {
    if (!out || n_tables < 0) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_create: bad argument");
    PulseStopRule* h = new PulseStopRule();
    h->n = n_tables; h->threshold = threshold;
    hipError_t e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&h->ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->copied[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->counts_dev), 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->counts_dev, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->counts_host), 2 * sizeof(unsigned long long), hipHostMallocDefault);
    h->max_waves = (int)(((long long)n_tables * 16 + 63) / 64) + 4;       // 16 lanes per table at most
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->waves_dev), 2 * (size_t)h->max_waves * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(h->waves_dev, 0, 2 * (size_t)h->max_waves * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->waves_host), 2 * (size_t)h->max_waves * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) { delete h; return pulse::fail_hip((int)e, "pulse_stoprule_create"); }
    h->counts_host[0] = h->counts_host[1] = 0;
    *out = h;
    return 0;
}

int pulse_stoprule_destroy(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return 0;
    (void)hipStreamSynchronize(h->side);
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(h->ready[i]); (void)hipEventDestroy(h->copied[i]); }
    (void)hipFree(h->counts_dev); (void)hipHostFree(h->counts_host); (void)hipFree(h->waves_dev); (void)hipHostFree(h->waves_host);
    (void)hipStreamDestroy(h->side);
    delete h;
    return 0;
}

namespace {
void stoprule_make_room(PulseStopRule* h) {                // bounded run-ahead: never reuse a slot still in flight
    while (h->n_pending >= 2) {
        (void)hipEventSynchronize(h->copied[h->pending[0] & 1]);
        h->late_over = stoprule_pop(h) || h->late_over;
    }
}
// n_waves > 0: the roll-out's last launch has stored its per-wavefront counts in the slot; 0: count with the kernel
int stoprule_submit(PulseStopRule* h, const uint8_t* is_done, hipStream_t st, int n_waves) {
    stoprule_make_room(h);
    const int slot = (int)(h->chunk & 1);
    h->waves_used[slot] = n_waves;
    if (h->n > 0 && n_waves == 0) {
        const int grid = min(1024, (h->n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(poker_stats_kernel, dim3(grid), dim3(kBlock), 0, st, is_done, (const float*)nullptr, (const uint8_t*)nullptr, h->n,
                           h->counts_dev + slot, (double*)nullptr);
    }
    hipError_t e = hipEventRecord(h->ready[slot], st);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->ready[slot], 0);
    if (e == hipSuccess) {
        if (n_waves > 0) e = hipMemcpyAsync(h->waves_host + (size_t)slot * h->max_waves, h->waves_dev + (size_t)slot * h->max_waves,
                                            (size_t)n_waves * sizeof(uint32_t), hipMemcpyDeviceToHost, h->side);
        else e = hipMemcpyAsync(h->counts_host + slot, h->counts_dev + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->side);
    }
    if (e == hipSuccess) e = hipEventRecord(h->copied[slot], h->side);
    if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_stoprule_submit");
    h->pending[h->n_pending++] = h->chunk;
    ++h->chunk;
    return 0;
}
}  // namespace

int pulse_stoprule_submit(void* handle, const uint8_t* is_done, void* stream) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !is_done) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_submit: null argument");
    return stoprule_submit(h, is_done, (hipStream_t)stream, 0);
}

int pulse_stoprule_over(void* handle, int32_t blocking, int32_t* over) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !over) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_over: null argument");
    bool o = h->late_over; h->late_over = false;
    while (h->n_pending > 0) {
        hipEvent_t ev = h->copied[h->pending[0] & 1];
        if (blocking) (void)hipEventSynchronize(ev);
        else if (hipEventQuery(ev) != hipSuccess) break;
        o = stoprule_pop(h) || o;
    }
    *over = o ? 1 : 0;
    return 0;
}

int pulse_stoprule_drain(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_drain: null argument");
    while (h->n_pending > 0) { (void)hipEventSynchronize(h->copied[h->pending[0] & 1]); (void)stoprule_pop(h); }
    h->late_over = false;
    return 0;
}

int pulse_poker_rollout(const PulsePokerView* v_even, const PulsePokerView* v_odd, const uint8_t* agent_types,
                        uint64_t seed, uint64_t step_counter0, uint64_t table_id0, int64_t* actions, float* rewards_even,
                        float* rewards_odd, int32_t n_steps, int32_t time_every, void* stoprule, void* stream) {
    if (int rc = check_view(v_even, "pulse_poker_rollout")) return rc;
    if (int rc = check_view(v_odd, "pulse_poker_rollout")) return rc;
    if (!actions || !rewards_even || !rewards_odd || !agent_types || n_steps < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_rollout: bad argument");
    if (v_even->n_games == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t packed = pack_types(agent_types, v_even->n_players);
    // time_every > 0: bracket the whole chunk of launches with one HIP event pair on the launch stream
    // (every time_every-th call only: the two timing events cost about as much queue time as a tenth of a chunk)
    const bool timed = time_every > 0 && n_steps > 0 && g_ev_used < kMaxTimed && (g_rollout_calls++ % time_every) == 0;
    if (timed) {
        if (g_ev_used >= g_ev_created) {
            if (hipEventCreate(&g_ev_start[g_ev_created]) != hipSuccess || hipEventCreate(&g_ev_stop[g_ev_created]) != hipSuccess)
                return pulse::fail(PULSE_ENODEVICE, "pulse_poker_rollout: hipEventCreate failed");
            ++g_ev_created;
        }
        (void)hipEventRecord(g_ev_start[g_ev_used], st);
    }
    PulseStopRule* rule = static_cast<PulseStopRule*>(stoprule);
    const int n_waves = (int)(((long long)v_even->n_games * lanes_per_table() + 63) / 64);
    const bool ride = rule && n_steps > 0 && n_waves <= rule->max_waves && rule->n == v_even->n_games;
    if (ride) stoprule_make_room(rule);                    // the slot must be free before the last launch writes into it
    for (int i = 0; i < n_steps; ++i) {
        const PulsePokerView& v = (i & 1) ? *v_odd : *v_even;
        float* rw = (i & 1) ? rewards_odd : rewards_even;
        PolicyArgs pa{packed, seed, step_counter0 + (uint64_t)i, table_id0, nullptr};
        if (ride && i == n_steps - 1) pa.wave_done = rule->waves_dev + (size_t)(rule->chunk & 1) * rule->max_waves;
        launch_step<PULSE_PH_STEP, true>(v, actions, nullptr, rw, pa, st);
    }
    if (timed) { (void)hipEventRecord(g_ev_stop[g_ev_used], st); g_ev_launches[g_ev_used] = n_steps; ++g_ev_used; }
    if (stoprule && n_steps > 0) {                         // the done flags of the state the last launch produced
        const PulsePokerView& last = ((n_steps - 1) & 1) ? *v_odd : *v_even;
        if (int rc = stoprule_submit(rule, last.is_done_out, st, ride ? n_waves : 0)) return rc;
    }
    return finish_launch("pulse_poker_rollout");
}

int pulse_rollout_timing_collect(float* sum_ms, int32_t* n_timed) {
    if (!sum_ms || !n_timed) return pulse::fail(PULSE_EINVAL, "pulse_rollout_timing_collect: null argument");
    float total = 0.0f;
    for (int i = 0; i < g_ev_used; ++i) {
        float ms = 0.0f;
        const hipError_t e = hipEventElapsedTime(&ms, g_ev_start[i], g_ev_stop[i]);
        if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_rollout_timing_collect (call it after a stream sync)");
        total += ms;
    }
    int launches = 0;
    for (int i = 0; i < g_ev_used; ++i) launches += g_ev_launches[i];
    *sum_ms = total; *n_timed = launches;
    g_ev_used = 0;
    return 0;
}

int pulse_poker_stats(const uint8_t* is_done, const float* rewards, const uint8_t* mask, int32_t n, int64_t* stats,
                      double* fstats, void* stream) {
    if ((!stats && !fstats) || (rewards && !fstats) || n < 0) return pulse::fail(PULSE_EINVAL, "pulse_poker_stats: bad argument");
    if (n == 0) return 0;
    const int grid = min(1024, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(poker_stats_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, is_done, rewards, mask, n,
                       reinterpret_cast<unsigned long long*>(stats), fstats);
    return finish_launch("pulse_poker_stats");
}

int pulse_poker_hand_metrics(const uint8_t* dones, const uint8_t* terminated_before, const int32_t* stacks, int32_t n_players,
                             const int32_t* initial_q_stacks, const int32_t* stages, const int32_t* button, int32_t q_seat,
                             int32_t active_players, int32_t n, int64_t* acc, void* stream) {
    if (!dones || !stacks || !initial_q_stacks || !stages || !button || !acc || n < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_hand_metrics: null argument");
    if (n_players < 2 || n_players > PULSE_MAX_SEATS || active_players < 2 || active_players > n_players || q_seat < 0 || q_seat >= n_players)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_hand_metrics: need 2 <= active_players <= n_players <= 16 and 0 <= q_seat < n_players");
    if (n == 0) return 0;
    const int grid = min(256, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(poker_hand_metrics_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, dones, terminated_before, stacks,
                       n_players, initial_q_stacks, stages, button, q_seat, active_players, n, reinterpret_cast<unsigned long long*>(acc));
    return finish_launch("pulse_poker_hand_metrics");
}

}  // extern "C"
