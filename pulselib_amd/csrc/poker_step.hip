// poker_step.hip -- the fused hold'em step for gfx950: one launch per PokerGPU.step, or ONE launch for a whole
// chunk of roll-out steps with the tables' state held in registers in between.
//
// Replaces the ~1,700 eager torch dispatches of one PokerGPU.step (environments/Poker/PokerGPU.py:527-633).
//
// Mapping: a table is owned by LPT adjacent lanes (4 = one DPP quad; the code also compiles for 2), lane j owns the SPL
// consecutive seats SPL*j .. SPL*j+SPL-1: 64/LPT tables per 64-wide wavefront.  Per-table scalars are replicated in the
// table's lanes, so the scalar part of the state machine costs 1/LPT wave-instruction per table; a lane's cells of a
// per-seat row ([N,P] int32, the reference's own layout) are one 12- or 16-byte load.  Seat sets (ACTIVE seats,
// contenders, winners) are bitmasks OR-reduced across the table's lanes with DPP quad_perm modifiers (no LDS); "first
// ACTIVE seat after x" is a rotate + ffs on the mask; side-pot layers are min/max butterflies.  Single steps run with
// LPT = 4, chunks with LPT = 2 up to ten seats (see lanes_for below for the measurements).  All integer arithmetic is the reference's; the fp32 reward keeps torch's
// op order (no contraction; tanh rounded once from double).
//
// Memory: single step -- state is read once and only the words that changed are written back, in the reference's
// own SoA tensors, so the drop-in class exposes them unchanged.  Chunk (pulse_poker_rollout) -- state is read once
// per chunk, every step still stores its observation / reward / done flag / action (ping-pong buffers, exactly what
// n single launches leave behind), and the changed state words are written once at the end.  The 130 MB hand-rank
// table is only touched when a poked state misses the evaluation cache the reset kernel fills.
#include <algorithm>
#include <type_traits>

#include "poker_device.h"
#include "qnet_device.h"

using namespace pulse_dev;

namespace {

// threads per workgroup of the step kernel.  Its wavefronts do not cooperate (only the stop rule's sum in workgroup 0
// does), so smaller workgroups only change how finely the dispatcher refills a CU: measured per 5-step chunk at 65,536 /
// 1,048,576 tables, 256 threads: 39.5 / 421 us, 128: 39.6 / 404, 64: 40.5 / 400.
#ifndef PULSE_STEP_BLOCK
#define PULSE_STEP_BLOCK 128
#endif
constexpr int kStepBlock = PULSE_STEP_BLOCK;

// dwords of LDS per wavefront of a chunk launch: the observation staging block + the staged read-only rows
// (hole cards [T][P_][2], 7-card rank | hand class << 16 [T][P_], flop and turn equities [T][2][P_], deck window [T][8]);
// 16-byte aligned so that the observation block of the next wavefront is.
// SLIM (MULTI == 2): the river's equities are not staged (they follow from the 7-card rank) and the hand class shares
// a word with the rank: 5 instead of 7 dwords per seat, which is what lets three two-lane wavefronts per SIMD fit the LDS at large batches.
__host__ __device__ constexpr int chunk_lds_dwords(int obs_size, int seats, int tables, bool slim) { return (tables * (obs_size + seats * (slim ? 5 : 7) + 8) + 3) & ~3; }

struct PolicyArgs {
    uint64_t types_packed, seed, step_counter, table_id0;
    uint32_t* wave_done;                   // nullptr, or one word per wavefront of the launch: tables done after the last step
    // the PREVIOUS check point's wavefront counts, summed and published to the host by workgroup 0 of this launch, before its own tables
    const uint32_t* carry_partials; int carry_n; long long* carry_host; long long carry_seq;
    // paired launches (pulse_internal.h: StopRulePair): a second carried check point, the counts after the launch's first
    // chunk, and the verdict word {launch id << 8 | skip_all | stop_mid << 1 | error << 7} the host answers the carries with
    const uint32_t* carry2_partials; int carry2_n; long long* carry2_host; long long carry2_seq;
    uint32_t* wave_done_mid; int mid_step;
    const long long* verdict_host; long long* verdict_dev; long long verdict_id;
    long long* verdict_err;                // pinned: set by a launch that gave up waiting for its verdict (the host then runs its steps unpaired)
    long long verdict_ticks;               // ticks of the 100 MHz wall clock thread 0 waits for the word: a launch never waits for a dead host for ever
};
constexpr int kVerdictStride = 16;                          // long longs between the 64 copies of a relayed verdict word (one 128-byte line each)
struct ChunkArgs {                         // MULTI only: the odd steps' output buffers and the number of steps
    float* obs_odd;
    float* rewards_odd;
    int n_steps;
};

// The kernel-argument segment as the kernel below receives it.  Values needed once inside (or after) the long step loop
// are re-read from it through a constant-address-space pointer (scalar loads) instead of being kept in scalar
// registers across the loop, where they were spilled to vector lanes.
struct StepKernargs { PulsePokerView v; int64_t* actions; const int32_t* actor_idx_in; float* rewards; PolicyArgs pa; ChunkArgs ca; };
typedef const __attribute__((address_space(4))) StepKernargs* StepKernargsPtr;

// In-kernel timeline (diagnostic build only, -DPULSE_STAMPS=1 -> libpulse_hip_stamps.so; no stamp executes
// in the product): lane 0 of every wavefront stores s_memtime at phase boundaries into a buffer of its own.
#ifndef PULSE_STAMPS
#define PULSE_STAMPS 0
#endif
#if PULSE_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(i) do { if ((threadIdx.x & 63) == 0 && g_stamp_buf) { __builtin_amdgcn_sched_barrier(0); \
    g_stamp_buf[((size_t)blockIdx.x * (kStepBlock / 64) + (threadIdx.x >> 6)) * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// tanh rounded once from double: 1 - 2/(exp(2x)+1), everything in double (abs. error ~1e-16, far below the fp32 ulp:
// the result is the correctly rounded fp32 tanh except in ~1e-8 of cases).  The exponential is written out -- the
// library's exp + IEEE division carry special-case handling this argument range never needs, and double-precision
// vector instructions issue at half rate, so they were a sixth of the kernel's vector work: k = rint(y log2 e),
// r = y - k ln 2 (two-part constant), e^r by its Taylor series to r^13 (|r| <= 0.347: remainder < 5e-18), 2^k by
// ldexp; the reciprocal is v_rcp_f64 + two Newton steps.  Below |x| = 1e-3 the difference 1 - 2/(..) cancels, and
// x (1 - x^2/3) is exact to double rounding instead.  Against libm's tanh rounded to fp32: 2 of 4e7 arguments differ.
__device__ __forceinline__ float tanh_rn(float x) {
    const double xd = (double)x;
    double y = 2.0 * xd;
    y = fmin(fmax(y, -40.0), 40.0);                       // tanh is +-1 in fp32 long before; keeps 2^k finite
    const double k = rint(y * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, y);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                    // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                  // 1/12!
    p = fma(p, r, 2.505210838544172e-08);                 // 1/11!
    p = fma(p, r, 2.755731922398589e-07);                 // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                  // 1/8!
    p = fma(p, r, 1.984126984126984e-04);                 // 1/7!
    p = fma(p, r, 1.388888888888889e-03);                 // 1/6!
    p = fma(p, r, 8.333333333333333e-03);                 // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);                // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double d = ldexp(p, (int)k) + 1.0;              // exp(2x) + 1
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    q = fma(fma(-d, q, 1.0), q, q);
    const double big = fma(-2.0, q, 1.0), small = xd * fma(-xd * xd, 0.3333333333333333, 1.0);
    return (float)(fabs(xd) < 1e-3 ? small : big);
}

// SPL consecutive int32 cells of a row, loaded as one 4-byte-aligned vector (global_load_dwordx3 / x4 take any dword address)
template <int N_> struct alignas(4) SeatCells { int32_t v[N_]; };

// WOBS (n_games % 16 == 0): the observation rows of a wavefront's 16 tables are one contiguous
// 16 x obs_size x 4 B block in HBM.  Written column by column they cost thirteen store instructions that each
// touch sixteen cache lines; with WOBS the lanes drop their values into the wavefront's LDS slice and the
// block leaves as three 1-KiB bursts (16 B per lane).  A wavefront's LDS operations retire in order, so no
// workgroup barrier is involved.
// MULTI: ca.n_steps steps in one launch (fused policy only); step i writes observation / done flag / reward into
// the even (i even) or odd buffers, as n single launches on the two ping-pong views would.
// ACT (the learner in the loop): the workgroup -- 256 threads, 128 tables at two lanes each -- first picks
// the learner's actions for its own window of tables (pulse_qnet::act_window, the body of the stand-alone act launch: masks,
// row lists for the training launch, forward on the matrix cores, argmax / epsilon draw) and then steps those tables: act ->
// env step has no grid-wide dependence, so the two launches of the trainer's step are one, and the tables' state is already
// on its way from HBM while the forward runs.  `qa` = the act launch's arguments (ACT only).
template <uint32_t PH, bool POLICY, int LPT, int SPL, bool WOBS, int MULTI, bool ACT>
__device__ __forceinline__ void poker_step_body(const PulsePokerView& v, int64_t* __restrict__ actions, const int32_t* __restrict__ actor_idx_in,
                                                float* __restrict__ rewards, const PolicyArgs& pa, const ChunkArgs& ca, const pulse_qnet::QNetArgs* qa) {
    static_assert((LPT == 4 && (SPL == 3 || SPL == 4)) || (LPT == 2 && (SPL == 5 || SPL == 8)), "lanes per table x seats per lane: 4x3, 4x4, 2x5, 2x8");
    constexpr int TPW = 64 / LPT;                                        // tables per wavefront
    constexpr int BLK = ACT ? 256 : kStepBlock;                          // threads per workgroup
    static_assert(!MULTI || (POLICY && PH == PULSE_PH_STEP), "a chunk is fused policy + full step");
    static_assert(!ACT || (MULTI == 1 && LPT == 2), "the act + step launch is a one-step chunk at two lanes per table");
    extern __shared__ int4 smem4[];
    if (POLICY && pa.carry_n > 0 && blockIdx.x == 0) {
        // The stop rule's previous check point: workgroup 0 sums its per-wavefront counts and publishes the total to
        // the host BEFORE its own tables.  (A separate last workgroup did this in the first version; at 65,536 tables
        // the grid fills the chip exactly, so that workgroup only got a slot when the first one retired and the host
        // learned the count ~35 us later than it could -- too late to keep the queue fed at an episode boundary.)
        sum_and_publish<BLK>(pa.carry_partials, pa.carry_n, nullptr, pa.carry_host, pa.carry_seq);
        if (pa.carry2_n > 0) { __syncthreads(); sum_and_publish<BLK>(pa.carry2_partials, pa.carry2_n, nullptr, pa.carry2_host, pa.carry2_seq); }
    }
    if (ACT) {
        // the learner's actions of this workgroup's 128 tables first (the act window owns the workgroup's whole LDS block; the
        // step's staging reuses it behind the barrier).  Measured: with the step's state loads issued BEFORE the forward -- in
        // flight underneath it -- the kernel needs 256 registers and spills (42.6 us against 22.7 + 15.2 for the two launches).
        pulse_qnet::act_window<true, 128, 5>(*qa, reinterpret_cast<float*>(smem4), (int)blockIdx.x);
        __syncthreads();
    }
    const int gt = blockIdx.x * BLK + threadIdx.x;
    const int t = gt / LPT;
    const int j_lane = gt % LPT;
    const int j = j_lane;
    if (!ACT && t >= v.n_games) return;   // the lanes of a table leave together (ACT: the host launches whole windows only -- every thread meets the barriers)
    STAMP(0);
    const int P = v.n_players, A = v.active_players;
    const int32_t* __restrict__ hr = v.hand_ranks;
    const uint32_t hr_len = (uint32_t)v.hand_ranks_len;

    // ---- load (every load is independent: all in flight at once).  n_games <= 2^24 (check_view): 24-bit multiplies.
    const uint32_t ut = (uint32_t)t, so = ut * 4u, bo = ut * 20u;      // byte offsets into [N] int32 / board
    int idx = ldo(v.idx, so), button = ldo(v.button, so), pot = ldo(v.pots, so), stage = ldo(v.stages, so), dpos = ldo(v.deck_positions, so);
    int highest = ldo(v.highest, so), agg = ldo(v.agg, so), acted = ldo(v.acted, so), lrs = ldo(v.last_raise_size, so);
    bool done = ldo(v.is_done, ut) != 0;
    bool dirty = ldo(v.equity_dirty, ut) != 0;
    const SeatCells<5> bd_ = ldo(reinterpret_cast<const SeatCells<5>*>(v.board), bo);          // 16 + 4 bytes
    int b0 = bd_.v[0], b1 = bd_.v[1], b2 = bd_.v[2], b3 = bd_.v[3], b4 = bd_.v[4];
    int stack[SPL], bet[SPL], inv[SPL], status[SPL], h0[SPL], h1[SPL];
    float eq[SPL];
    const uint32_t row0 = __umul24(ut, (uint32_t)P), eq0 = __umul24(ut, (uint32_t)A);
    // Seats are dealt to the lanes of a table in blocks: lane j owns seats SPL*j .. SPL*j + SPL-1, i.e. SPL consecutive
    // cells of every [N,P] row -- one 12- or 16-byte load per array instead of SPL dword loads (the prologue is bound by
    // the number of vector-memory instructions a wavefront issues, not by bytes).  The block of a table's last lane runs
    // past the row (P = 10: seats 10, 11 = the next table's first cells: loaded, masked); where it would run past the
    // ARRAY (the last table) the cells are loaded one by one.
#define SEAT(k) (SPL * j + (k))
#define ROW_OFF(k) ((row0 + (uint32_t)SEAT(k)) * 4u)                   /* byte offset of (table, seat) in an [N,P] int32 array */
    const uint32_t cell0 = row0 + (uint32_t)(SPL * j), ecell0 = eq0 + (uint32_t)(SPL * j);     // this lane's first cell
    // (plain 32-bit multiplies: n_games may be 2^24 itself, which a 24-bit multiply truncates to 0)
    const bool vec_ok = cell0 + (uint32_t)SPL <= (uint32_t)v.n_games * (uint32_t)P &&
                        ecell0 + (uint32_t)SPL <= (uint32_t)v.n_games * (uint32_t)A;   // (implies the same for hands and pre_eq)
    // VEC = std::true_type: one vector load; std::false_type: cell by cell.  `limit` = seats that exist in the row.
    auto load_cells = [&](auto VEC, const int32_t* base, uint32_t first_cell, int limit, int (&out)[SPL], int fill) {
        SeatCells<SPL> c;
        if (decltype(VEC)::value) c = ldo(reinterpret_cast<const SeatCells<SPL>*>(base), first_cell * 4u);
        else {
#pragma unroll
            for (int k = 0; k < SPL; ++k) c.v[k] = SEAT(k) < limit ? ldo(base, (first_cell + (uint32_t)k) * 4u) : fill;
        }
#pragma unroll
        for (int k = 0; k < SPL; ++k) out[k] = SEAT(k) < limit ? c.v[k] : fill;
    };
    auto load_hands = [&](auto VEC, int (&o0)[SPL], int (&o1)[SPL]) {
        SeatCells<2 * SPL> c;
        if (decltype(VEC)::value) c = ldo(reinterpret_cast<const SeatCells<2 * SPL>*>(v.hands), cell0 * 8u);
        else {
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                int2 h = make_int2(-1, -1);
                if (SEAT(k) < P) h = ldo(reinterpret_cast<const int2*>(v.hands), ROW_OFF(k) * 2u);
                c.v[2 * k] = h.x; c.v[2 * k + 1] = h.y;
            }
        }
#pragma unroll
        for (int k = 0; k < SPL; ++k) { o0[k] = SEAT(k) < P ? c.v[2 * k] : -1; o1[k] = SEAT(k) < P ? c.v[2 * k + 1] : -1; }
    };
    const bool cache_on = (PH & (PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN)) && v.pre_board;
    int ph_[SPL], pr_[SPL], e1_[SPL], e2_[SPL], e3_[SPL];            // chunk only: the evaluation cache's rows (staged in LDS below)
    auto load_rows = [&](auto VEC) {
        load_cells(VEC, v.stacks, cell0, P, stack, 0); load_cells(VEC, v.current_round_bet, cell0, P, bet, 0);
        load_cells(VEC, v.total_invested, cell0, P, inv, 0); load_cells(VEC, v.status, cell0, P, status, PULSE_SITOUT);
        int ei[SPL];
        load_cells(VEC, reinterpret_cast<const int32_t*>(v.equities), ecell0, A, ei, 0x3f000000 /* 0.5f */);
#pragma unroll
        for (int k = 0; k < SPL; ++k) eq[k] = __int_as_float(ei[k]);
        load_hands(VEC, h0, h1);          // (a chunk keeps them in LDS: they never change inside an episode)
        if (MULTI) {
#pragma unroll
            for (int k = 0; k < SPL; ++k) { ph_[k] = 0; pr_[k] = 0; e1_[k] = 0; e2_[k] = 0; e3_[k] = 0; }
            if (cache_on) {
                load_cells(VEC, v.pre_hands, cell0, P, ph_, 0); load_cells(VEC, v.pre_rank, cell0, P, pr_, 0);
                const int32_t* pe = reinterpret_cast<const int32_t*>(v.pre_eq);     // [N,3,P]: three rows of P per table
                const uint32_t c0 = __umul24(ut, 3u * (uint32_t)P) + (uint32_t)(SPL * j);     // (ut * 3 leaves 24 bits above 5.59 M tables; 3 P <= 48 does not)
                load_cells(VEC, pe, c0, P, e1_, 0); load_cells(VEC, pe, c0 + (uint32_t)P, P, e2_, 0);
                if (MULTI != 2) load_cells(VEC, pe, c0 + 2u * (uint32_t)P, P, e3_, 0);          // (SLIM: the river row is not read)
            }
        }
    };
    if (vec_ok) load_rows(std::true_type{}); else load_rows(std::false_type{});
    // ---- chunk only: the read-only rows the steps consult (hole cards, the evaluation cache, the next cards of the deck)
    // are staged in the wavefront's LDS slice ONCE.  A global load inside the step loop would have to be waited for
    // with s_waitcnt vmcnt, which counts stores too -- i.e. for every observation / reward / done store of the
    // previous step (the profile showed wavefronts parked there half of their life); LDS reads wait for nothing.
    constexpr int P_ = LPT * SPL;                                       // seats per table as staged (10, 12 or 16)
    const int wave = threadIdx.x >> 6, q = (threadIdx.x & 63) / LPT;     // wavefront of the workgroup, table of the wavefront
    constexpr bool SLIM = MULTI == 2;                                   // chunk for large batches: see chunk_lds_dwords
    constexpr int EQROWS = SLIM ? 2 : 3;
    int32_t* const lw = reinterpret_cast<int32_t*>(smem4) + wave * chunk_lds_dwords(v.obs_size, P_, TPW, SLIM);
    int32_t* const l_hands = lw + TPW * v.obs_size;                      // [TPW][P_][2]
    int32_t* const l_prerank = l_hands + TPW * P_ * 2;                   // [TPW][P_]: 7-card rank (SLIM: its 16 bits | hand class of the seat << 16)
    float* const l_preeq = reinterpret_cast<float*>(l_prerank + TPW * P_);  // [TPW][EQROWS][P_]: flop, turn (and, not SLIM, river) equities
    int32_t* const l_deck = reinterpret_cast<int32_t*>(l_preeq + TPW * EQROWS * P_);  // [TPW][8]: cards at deck position dpos0 + 0..7
    int32_t* const l_class = l_deck + TPW * 8;                            // [TPW][P_] (not SLIM): hand class of every seat (scripted players)
    constexpr int DPL = 8 / LPT;                                         // deck-window entries each lane stages
    const int dpos0 = dpos;
    bool seat_hit[SPL];               // chunk: the cache entry of this lane's seat k was made from the hole cards it holds now
#pragma unroll
    for (int k = 0; k < SPL; ++k) seat_hit[k] = false;
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int seat = SEAT(k);
            *reinterpret_cast<int2*>(l_hands + (q * P_ + seat) * 2) = make_int2(h0[k], h1[k]);
            seat_hit[k] = ((uint32_t)ph_[k] & (kPreHandsValid * 2u - 1u)) == pack_hand(h0[k], h1[k]) && card_ok(h0[k]) && card_ok(h1[k]);   // hole cards are fixed for the episode
            uint32_t cls = (uint32_t)ph_[k] >> kClsShift;                    // the reset kernel classified the cards the tag names
            if (POLICY && !seat_hit[k]) {          // seats outside the hand hold (-1, -1): a constant; anything else (no cache, poked cards) is classified here
                const bool empty = h0[k] == -1 && h1[k] == -1;
                cls = hand_class(-1, -1);
                if (!empty) cls = hand_class(h0[k], h1[k]);
            }
            if (SLIM) l_prerank[q * P_ + seat] = (int32_t)(((uint32_t)pr_[k] & 0xFFFFu) | cls << 16);      // a valid rank is below 2^16 (PokerGPU.py:13-18)
            else { l_prerank[q * P_ + seat] = pr_[k]; l_class[q * P_ + seat] = (int32_t)cls; }
            l_preeq[(q * EQROWS + 0) * P_ + seat] = __int_as_float(e1_[k]); l_preeq[(q * EQROWS + 1) * P_ + seat] = __int_as_float(e2_[k]);
            if (!SLIM) l_preeq[(q * EQROWS + 2) * P_ + seat] = __int_as_float(e3_[k]);
        }
        const int32_t* dk = v.decks + (size_t)t * 52;
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int at = dpos0 + DPL * j + e;
            l_deck[q * 8 + DPL * j + e] = (uint32_t)at < 52u ? dk[at] : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // Paired launch: the host answers the carried counts with this launch's verdict word; thread 0 of the launch relays
    // it from pinned memory into device memory (one poller on PCIe), everybody reads it there before the first step --
    // nothing has been stored yet: if the episode turns out to have ended before this launch, the launch leaves no
    // trace.  (The answer takes the host ~5 us from the launch's start; the load burst above takes the wavefronts longer.)
    bool stop_mid = false;                 // the previous launch's final count ends the episode after this launch's first chunk
    if (MULTI && pa.verdict_dev) {
        // All accesses to the relayed word are RELAXED system-scope atomics: they bypass the (per-XCD, mutually
        // non-coherent) L2s without the cache-wide invalidate / write-back an acquire / release at agent scope costs --
        // 2,048 wavefronts invalidating their L2 in a spin loop tripled the launch time.  The word carries its own
        // launch id, so no other data needs ordering with it.  It is replicated over 64 cache lines (one store
        // instruction of the relaying wavefront); wavefront w polls copy w mod 64.
        if (blockIdx.x == 0 && threadIdx.x < 64) {
            int flags = 0;
            if (threadIdx.x == 0) {
                const long long t0 = wall_clock64();
                for (;;) {
                    const long long w = __hip_atomic_load(pa.verdict_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((w >> 8) == pa.verdict_id) { flags = (int)(w & 0xFF); break; }
                    if (wall_clock64() - t0 > pa.verdict_ticks) {                              // too late: run nothing, and tell the host
                        flags = 0x81;
                        if (pa.verdict_err) __hip_atomic_store(pa.verdict_err, pa.verdict_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            flags = __builtin_amdgcn_readfirstlane(flags);
            __hip_atomic_store(pa.verdict_dev + (threadIdx.x & 63) * kVerdictStride, (pa.verdict_id << 8) | (long long)flags, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const long long* mine = pa.verdict_dev + (((blockIdx.x * BLK + threadIdx.x) >> 6) & 63) * kVerdictStride;
        const long long t0 = wall_clock64();
        long long w;
        for (;;) {
            w = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((w >> 8) == pa.verdict_id) break;
            if (wall_clock64() - t0 > 2 * pa.verdict_ticks + 100000000ll) { w = 0x81; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        const int flags = __builtin_amdgcn_readfirstlane((int)w);
        if (flags & 1) return;
        stop_mid = (flags & 2) != 0;
    }
    // hole cards of one seat, as every lane of the table may ask: what SEAT_PICK(h0/h1, seat) yields (seat in 0..15)
    auto hand_of_seat = [&](int seat) -> int2 {
        if (seat < P_) return *reinterpret_cast<const int2*>(l_hands + (q * P_ + seat) * 2);
        return make_int2(0, 0);
    };
    // card at deck position `at`: from the staged window.  A hand deals at most 8 positions past its first street, so
    // only a poked state can leave the window; it is then refilled from `at` on -- inside this branch, loads waited for
    // here -- so that the ordinary path holds no global load (and no vmcnt wait at the join) at all.
    int dwin0 = dpos0;
    auto deck_card = [&](int at) -> int {
        if (!MULTI) return (uint32_t)at < 52u ? v.decks[(size_t)t * 52 + at] : 0;
        if ((uint32_t)(at - dwin0) >= 8u) {                         // table-uniform: the table's lanes compute the same `at`
            dwin0 = at;
            const int32_t* dk = v.decks + (size_t)t * 52;
            int c[DPL];
#pragma unroll
            for (int e = 0; e < DPL; ++e) { const int a = at + DPL * j + e; c[e] = (uint32_t)a < 52u ? dk[a] : 0; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int e = 0; e < DPL; ++e) l_deck[q * 8 + DPL * j + e] = c[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return l_deck[q * 8 + (at - dwin0)];
    };
    long long act64 = 0;
    if ((PH & (PULSE_PH_EXECUTE | PULSE_PH_REWARD)) && actions) act64 = ldo(actions, ut * 8u);
    // the Philox draws depend on no load: issue them now, they execute under the load latency.  In a chunk lane j
    // of the table computes call (first + j): LPT calls = 2 * LPT steps of draws in the time of one.
    const uint64_t tid = pa.table_id0 + (uint64_t)t;
    uint64_t pool_base = pa.step_counter >> 1;
    U4 pool{0, 0, 0, 0};
    if (POLICY) pool = philox4x32(pa.seed, tid, pool_base + (MULTI ? (uint64_t)j : 0u));
    const float w1 = *v.w1, w2 = *v.w2;
    const int Kdiv = *v.K, alpha = *v.alpha;
    uint32_t pre_tag = 0;
    if ((PH & (PULSE_PH_EQUITY | PULSE_PH_SHOWDOWN)) && v.pre_board) pre_tag = (uint32_t)ldo(v.pre_board, so);

    // What the launch changes is written back once, after the last step; only changed records are stored (a table's
    // seat rows change in one or two cells per step; writing everything back doubles the store traffic).  A flag is
    // raised wherever a group of words is assigned: bit k of cells_dirty = this lane's seat k (its four row cells),
    // bet_dirty = {pot, highest, agg, acted, last_raise_size, idx}, street_dirty = {stage, deck position, dirty flag},
    // board_dirty = the five board cards.
    uint32_t cells_dirty = 0;
    bool bet_dirty = false, street_dirty = false, board_dirty = false, eq_dirty = false, act_dirty = false;
    int prev_stack = 0, prev_invested = 0;

    // seat-set bitmask of a per-seat predicate / value of one seat, visible to every lane of the table
#define SEAT_BITS(expr) ([&]() { uint32_t m_ = 0; _Pragma("unroll") for (int k = 0; k < SPL; ++k) m_ |= (uint32_t)((expr) ? 1u : 0u) << SEAT(k); return grp_or<LPT>(m_); }())
#define SEAT_PICK(arr, seat_) ([&]() { uint32_t r_ = 0; _Pragma("unroll") for (int k = 0; k < SPL; ++k) r_ |= SEAT(k) == (seat_) ? (uint32_t)(arr)[k] : 0u; return (int)grp_or<LPT>(r_); }())

    STAMP(1);   // loads issued
    // The seat to act, as every lane of the table sees it: status / stack / round bet / hole cards.  Nothing changes
    // between the observation a step writes (for the NEXT seat to act) and the following step's capture, so a chunk
    // carries these five values from one step to the next instead of picking them twice.
    int a_status = 0, a_stack = 0, a_bet = 0, a_h0 = 0, a_h1 = 0;
    uint32_t a_cls = 0;                  // chunk: the hand class of the seat to act instead of its two cards
    auto class_of_seat = [&](int seat) -> uint32_t { return seat < P_ ? (SLIM ? (uint32_t)l_prerank[q * P_ + seat] >> 16 : (uint32_t)l_class[q * P_ + seat]) : hand_class(0, 0); };
    if (MULTI) {
        const int seat0 = idx & 15;
        a_status = SEAT_PICK(status, seat0); a_stack = SEAT_PICK(stack, seat0); a_bet = SEAT_PICK(bet, seat0);
        a_cls = class_of_seat(seat0);
    }
    // the caller's action word was loaded with the state; pin its arrival to the prologue's waits: left pending, its one
    // use after the loop (the final store of the action) became an s_waitcnt vmcnt(0) behind every store of the last step
    if (MULTI) asm volatile("" : "+v"(act64));
    const int n_steps = MULTI ? (stop_mid ? pa.mid_step : ca.n_steps) : 1;        // (the verdict is known: no exit from inside the loop)
    // output buffers of this step / of the next one: swapped at the end of every step (scalar moves; a select on the
    // step's parity made the compiler keep both sets of per-lane addresses alive across the loop -- and spill them)
    float* obs_dst = v.obs; float* obs_nxt = MULTI ? ca.obs_odd : v.obs;
    uint8_t* done_dst = v.is_done_out; uint8_t* done_nxt = v.is_done;
    float* rew_dst = rewards; float* rew_nxt = MULTI ? ca.rewards_odd : rewards;
    const int mid_step = MULTI ? pa.mid_step : 0;          // > 0: the launch covers two check intervals, the first one mid_step steps long
    for (int i = 0; i < n_steps; ++i) {
        // In a chunk, everything derived from the lane's position in its table (seat numbers, `seat < A` masks, ...)
        // is re-derived inside the step: hoisted out of the loop those lane masks filled the scalar registers and were
        // spilled (~200 v_readlane per step to fetch them back); a compare is cheaper than its reload.
        int j = j_lane;
        if (MULTI) asm volatile("" : "+v"(j));

        // ---- capture (PokerGPU.py:530-539)
        const bool prev_done = done;
        const int actor = ((PH & PULSE_PH_CAPTURE) || !actor_idx_in ? idx : actor_idx_in[t]) & 15;
        if (!MULTI) {
            a_status = SEAT_PICK(status, actor); a_stack = SEAT_PICK(stack, actor); a_bet = SEAT_PICK(bet, actor);
            if (POLICY) { a_h0 = SEAT_PICK(h0, idx & 15); a_h1 = SEAT_PICK(h1, idx & 15); }
        }
        const bool has_legal_actor = a_status != PULSE_FOLDED && a_status != PULSE_ALLIN && a_status != PULSE_SITOUT && !prev_done;
        prev_invested = a_bet;
        if (!(PH & PULSE_PH_CAPTURE)) prev_invested = ldo(v.prev_invested, so);
        prev_stack = a_stack;

        STAMP(2);   // first loads have arrived (actor values picked)
        // ---- scripted opponents (environments/Poker/utils.py:108-123), fused in front of the step
        if (POLICY) {
            const uint64_t step = pa.step_counter + (uint64_t)i;
            PolicyDraw draw;
            if (MULTI) {
                const uint64_t call = step >> 1;
                if (call - pool_base >= (uint64_t)LPT) {              // wave-uniform: the pool of four calls is used up
                    pool_base = call;
                    pool = philox4x32(pa.seed, tid, call + (uint64_t)j);
                }
                const int rel = (int)(call - pool_base);
                const PolicyDraw mine = policy_draw(pool, step);
                draw.pick = grp_or<LPT>(j == rel ? mine.pick : 0u);
                draw.coin = grp_or<LPT>(j == rel ? mine.coin : 0u);
            } else {
                draw = policy_draw(pool, step);
            }
            const int type = (int)((pa.types_packed >> (4 * (idx & 15))) & 15u);
            if (type != PULSE_AGENT_EXTERNAL) {
                act64 = MULTI ? scripted_action_cls(type, a_cls, pot, draw) : scripted_action(type, a_h0, a_h1, pot, draw);
                act_dirty = true;                  // stored once, after the last step (a later step's action overwrites it anyway)
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
                    const int seat = SEAT(k);
                    float e = 0.5f;
                    if (seat < A && street) {
                        bool hit = false;
                        if (cached_board) {       // the cache reads are independent single hops
                            if (MULTI) {
                                hit = seat_hit[k];
                                if (SLIM && stage == 3) {      // the river's equity is the 7-card rank normalised (PokerGPU.py:481), as the reset kernel stored it
                                    e = __fdiv_rn(__fsub_rn((float)(l_prerank[q * P_ + seat] & 0xFFFF), 4109.0f), 32765.0f);
                                    e = fminf(fmaxf(e, 0.0f), 1.0f);
                                } else e = l_preeq[(q * EQROWS + (stage - 1)) * P_ + seat];
                            } else {
                                const uint32_t ph = (uint32_t)ldo(v.pre_hands, ROW_OFF(k));
                                const float pe = ldo(v.pre_eq, (__umul24(ut, 3u * (uint32_t)P) + __umul24((uint32_t)(stage - 1), (uint32_t)P) + (uint32_t)seat) * 4u);
                                hit = (ph & (kPreHandsValid * 2u - 1u)) == pack_hand(h0[k], h1[k]) && card_ok(h0[k]) && card_ok(h1[k]);
                                e = pe;
                            }
                        }
                        if (!hit) {               // the reference's literal seven-gather chain
                            int hc0 = h0[k], hc1 = h1[k];
                            if (MULTI) { const int2 h = hand_of_seat(seat); hc0 = h.x; hc1 = h.y; }
                            const float r = (float)walk7(hr, hr_len, hc0, hc1, b0, b1, b2, c5, c6);
                            e = stage == 1 ? __fdiv_rn(__fsub_rn(r, 74359.0f), 749420.0f) : __fdiv_rn(__fsub_rn(r, 4109.0f), 32765.0f);
                            e = fminf(fmaxf(e, 0.0f), 1.0f);
                        }
                    }
                    eq[k] = e;
                }
                dirty = false; street_dirty = true; eq_dirty = true;
            }
        }
        float e_actor;
        {
            uint32_t r_ = 0;
#pragma unroll
            for (int k = 0; k < SPL; ++k) r_ |= SEAT(k) == actor ? __float_as_uint(eq[k]) : 0u;
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
                bet_dirty = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k)
                    if (SEAT(k) == (idx & 15)) { stack[k] = n_stack; bet[k] = n_bet; inv[k] += n_inv_add; status[k] = n_status; cells_dirty |= 1u << k; }
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
            if (!round_over && has_next) { idx = next_seat; bet_dirty = true; }
            const bool early_term = contenders <= 1 && round_over;
            if (early_term) done = true;
            if (round_over && !early_term && !done) {
                lrs = 1; stage += 1; highest = 0; agg = mod_near(button + 1, A); acted = 0;
                bet_dirty = true; street_dirty = true;
#pragma unroll
                for (int k = 0; k < SPL; ++k) { cells_dirty |= (bet[k] != 0 ? 1u : 0u) << k; bet[k] = 0; }
                const int first = first_after_near(act_bits & maskA, button, A);
                if (first >= 0) idx = first;
                if (stage > 3) { done = true; stage = 4; }
                else {
                    const int nx0 = deck_card(dpos + 1);
                    if (stage == 1) {                                                // burn + flop (:601-604)
                        b0 = nx0;
                        b1 = deck_card(dpos + 2);
                        b2 = deck_card(dpos + 3);
                        dpos += 4;
                    }
                    else if (stage == 2) { b3 = nx0; dpos += 2; }                    // burn + turn (:607-610)
                    else { b4 = nx0; dpos += 2; }                                    // burn + river (:613-616)
                    dirty = true; board_dirty = true;
                }
            }
        }
        // a chunk fetches the next actor's hole cards now; they are first needed by the observation below
        int2 next_hand = make_int2(0, 0);
        uint32_t next_cls = 0;
        if (MULTI) { next_hand = hand_of_seat(idx & 15); next_cls = class_of_seat(idx & 15); }

        STAMP(6);   // advance / deal done
        // ---- 4) payouts on newly finished tables (PokerGPU.py:619-623)
        const bool newly_done = (PH & PULSE_PH_CAPTURE) ? (done && !prev_done) : done;
        if (PH & PULSE_PH_FOLDWIN) {                                            // :331-338
            if (newly_done && contenders == 1) {
                const int survivor = __ffs((int)cont_bits) - 1;
#pragma unroll
                for (int k = 0; k < SPL; ++k) if (SEAT(k) == survivor) { stack[k] += pot; cells_dirty |= 1u << k; }
                pot = 0; bet_dirty = true;
            }
        }
        if (PH & PULSE_PH_SHOWDOWN) {                                           // :380-453
            if (newly_done && stage < 5 && contenders > 1) {
                if (stage == 0) {
                    b0 = deck_card(dpos + 1); b1 = deck_card(dpos + 2); b2 = deck_card(dpos + 3);
                    b3 = deck_card(dpos + 5); b4 = deck_card(dpos + 7);
                    dpos += 8;
                } else if (stage == 1) {
                    b3 = deck_card(dpos + 1); b4 = deck_card(dpos + 3);
                    dpos += 4;
                } else if (stage == 2) {
                    b4 = deck_card(dpos + 1);
                    dpos += 2;
                }
                bool eligible[SPL]; int rank[SPL], payout[SPL];
                const bool cached_board = board_matches(pre_tag, 5, b0, b1, b2, b3, b4);
#pragma unroll
                for (int k = 0; k < SPL; ++k) {
                    const int seat = SEAT(k);
                    eligible[k] = seat < A && (status[k] == PULSE_ACTIVE || status[k] == PULSE_ALLIN);
                    rank[k] = INT_MIN; payout[k] = 0;
                    if (eligible[k]) {
                        bool hit = false;
                        if (cached_board) {
                            if (MULTI) {
                                hit = seat_hit[k];
                                rank[k] = SLIM ? l_prerank[q * P_ + seat] & 0xFFFF : l_prerank[q * P_ + seat];
                            } else {
                                const uint32_t ph = (uint32_t)ldo(v.pre_hands, ROW_OFF(k));
                                hit = (ph & (kPreHandsValid * 2u - 1u)) == pack_hand(h0[k], h1[k]) && card_ok(h0[k]) && card_ok(h1[k]);
                                rank[k] = ldo(v.pre_rank, ROW_OFF(k));
                            }
                        }
                        if (!hit) {
                            int hc0 = h0[k], hc1 = h1[k];
                            if (MULTI) { const int2 h = hand_of_seat(seat); hc0 = h.x; hc1 = h.y; }
                            rank[k] = walk7(hr, hr_len, hc0, hc1, b0, b1, b2, b3, b4);
                            asm volatile("" : "+v"(rank[k]));      // the chain's last load is waited for HERE, not at the join every table passes
                        }
                    }
                }
                // side pots, one layer per distinct commitment level (PokerGPU.py:340-378)
                int prev_level = 0;
                for (int l = 0; l < A; ++l) {
                    int lv = INT_MAX;
#pragma unroll
                    for (int k = 0; k < SPL; ++k) if (SEAT(k) < A && inv[k] > prev_level) lv = min(lv, inv[k]);
                    const int level = grp_imin<LPT>(lv);
                    if (level == INT_MAX) break;
                    const int n_contrib = __popc(SEAT_BITS(SEAT(k) < A && inv[k] >= level));
                    int bl = INT_MIN;
#pragma unroll
                    for (int k = 0; k < SPL; ++k) if (SEAT(k) < A && inv[k] >= level && eligible[k]) bl = max(bl, rank[k]);
                    const int best = grp_imax<LPT>(bl);
                    const uint32_t win_bits = SEAT_BITS(SEAT(k) < A && inv[k] >= level && eligible[k] && rank[k] == best);
                    const int n_win = __popc(win_bits);
                    if (n_win > 0) {
                        const int layer_pot = (level - prev_level) * n_contrib;
                        const int share = layer_pot / n_win, rem = layer_pot - share * n_win;
                        const int first_win = __ffs((int)win_bits) - 1;
#pragma unroll
                        for (int k = 0; k < SPL; ++k)
                            if ((win_bits >> SEAT(k)) & 1u) payout[k] += share + (SEAT(k) == first_win ? rem : 0);
                    }
                    prev_level = level;
                }
#pragma unroll
                for (int k = 0; k < SPL; ++k) { stack[k] += payout[k]; cells_dirty |= (payout[k] != 0 ? 1u : 0u) << k; }
                pot = 0; stage = 5;
                bet_dirty = true; street_dirty = true; board_dirty = true;
            }
        }
        if (PH & PULSE_PH_CLEARDONE) {                                          // :625-628
            if (done) {
                bet_dirty = bet_dirty || highest != 0;
                highest = 0;
#pragma unroll
                for (int k = 0; k < SPL; ++k) { cells_dirty |= ((bet[k] | inv[k]) != 0 ? 1u : 0u) << k; bet[k] = 0; inv[k] = 0; }
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
            if (j == 0) sto_in_loop(rew_dst, so, r);
        }

        STAMP(8);   // reward done
        // ---- 6) observation for the next seat to act (PokerGPU.py:159-179)
        if (PH & PULSE_PH_OBS) {
            const int wlane = threadIdx.x & 63;
            float* const l_obs = MULTI ? reinterpret_cast<float*>(lw) : reinterpret_cast<float*>(smem4) + (threadIdx.x >> 6) * TPW * v.obs_size;
            float* __restrict__ o = WOBS ? l_obs + q * v.obs_size
                                         : reinterpret_cast<float*>(reinterpret_cast<char*>(obs_dst) + __umul24(ut, (uint32_t)v.obs_size) * 4u);
            const int seat_i = idx & 15;
            const int n_h0 = MULTI ? next_hand.x : SEAT_PICK(h0, seat_i), n_h1 = MULTI ? next_hand.y : SEAT_PICK(h1, seat_i);
            const int n_stack = SEAT_PICK(stack, seat_i), n_status = SEAT_PICK(status, seat_i), n_bet = SEAT_PICK(bet, seat_i);
            if (MULTI) { a_status = n_status; a_stack = n_stack; a_bet = n_bet; a_cls = next_cls; }   // the next step's actor
            const int idxm = mod_near(idx, A);
            const int pos = mod_near(idx - button, A);
            // columns 0..12, LPT per pass: lane j writes column LPT*pass + j (an LPT-way select per pass)
            if (LPT == 4) {
                const int h0v = j == 0 ? b0 : j == 1 ? b1 : j == 2 ? b2 : b3;
                const int h1v = j == 0 ? b4 : j == 1 ? n_h0 : j == 2 ? n_h1 : stage;
                const int h2v = j == 0 ? pos : j == 1 ? pot : j == 2 ? highest - n_bet : n_stack;
                o[j] = (float)h0v; o[4 + j] = (float)h1v; o[8 + j] = (float)h2v;
                if (j == 0) o[12] = (float)n_status;
            } else {
                o[j] = (float)(j == 0 ? b0 : b1); o[2 + j] = (float)(j == 0 ? b2 : b3); o[4 + j] = (float)(j == 0 ? b4 : n_h0);
                o[6 + j] = (float)(j == 0 ? n_h1 : stage); o[8 + j] = (float)(j == 0 ? pos : pot);
                o[10 + j] = (float)(j == 0 ? highest - n_bet : n_stack);
                if (j == 0) o[12] = (float)n_status;
            }
            // opponents: seat (idx+1+k)%A -> columns 13+3k..; seats >= A zero-fill the padding slots
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int seat = SEAT(k);
                if (seat < v.max_players && seat != idxm) {
                    int slot; float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
                    if (seat < A) { slot = seat - idxm - 1; if (slot < 0) slot += A; f0 = (float)stack[k]; f1 = (float)status[k]; f2 = (float)bet[k]; }
                    else slot = seat - 1;
                    float* dst = o + 13 + 3 * slot;
                    dst[0] = f0; dst[1] = f1; dst[2] = f2;
                }
            }
            if (WOBS) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int n4 = TPW / 4 * v.obs_size;                             // int4 per wavefront block
                const int tw0 = (int)((blockIdx.x * BLK + threadIdx.x) >> 6) * TPW;  // first table of this wavefront
                const uint32_t blk0 = __umul24((uint32_t)tw0, (uint32_t)v.obs_size) * 4u;      // byte offset of the wavefront's block
                const int4* src = reinterpret_cast<const int4*>(l_obs);
                constexpr int kBursts = 5;          // two-lane chunk, 40 columns: 32 rows x 160 B = five 1-KiB bursts
                if (n4 == kBursts * 64) {
                    // all reads first, then all stores: as a loop over e every pass waited for its own LDS read
                    // (five exposed LDS latencies per step)
                    int4 x[kBursts];
#pragma unroll
                    for (int u = 0; u < kBursts; ++u) x[u] = src[wlane + 64 * u];
#pragma unroll
                    for (int u = 0; u < kBursts; ++u) sto_in_loop(reinterpret_cast<int4*>(obs_dst), blk0 + (uint32_t)(wlane + 64 * u) * 16u, x[u]);
                } else
                    for (int e = wlane; e < n4; e += 64) sto_in_loop(reinterpret_cast<int4*>(obs_dst), blk0 + (uint32_t)e * 16u, src[e]);
                if (MULTI) {       // the next step's values must not overtake these reads of the slice
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        } else if (MULTI) {
            // (not reachable: a chunk always has PULSE_PH_OBS) keep the carried actor consistent anyway
            const int seat_i = idx & 15;
            a_status = SEAT_PICK(status, seat_i); a_stack = SEAT_PICK(stack, seat_i); a_bet = SEAT_PICK(bet, seat_i);
            a_cls = next_cls;
        }
        STAMP(9);   // observation stores issued
        if ((PH & PULSE_PH_ADVANCE) && j == 0) sto_in_loop(done_dst, ut, (uint8_t)(done ? 1 : 0));      // ping-pong buffer: always written
        if (MULTI && i + 1 == mid_step) {        // the check point in the middle of a paired launch
            StepKernargsPtr ka = (StepKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const int c = __popcll(__ballot(done && j == 0));
            if ((threadIdx.x & 63) == 0) stg(ka->pa.wave_done_mid, (uint32_t)(blockIdx.x * (BLK / 64) + (threadIdx.x >> 6)) * 4u, (uint32_t)c);
        }
        if (MULTI) {
            float* fo = obs_dst; obs_dst = obs_nxt; obs_nxt = fo;
            uint8_t* fd = done_dst; done_dst = done_nxt; done_nxt = fd;
            float* fr = rew_dst; rew_dst = rew_nxt; rew_nxt = fr;
        }
    }

    // ---- store: the groups whose flag was raised (one test per seat / per group instead of one branch per word).
    // In a chunk the ~25 state pointers would have to stay in scalar registers across the whole step loop (they were
    // spilled, 95 of them); the kernel argument block is re-read here through a pointer the compiler cannot see through.
    // The pointer stays in the constant address space (scalar loads of the array pointers, lgkmcnt only -- as a generic
    // pointer they became vector loads waited for with vmcnt(0), i.e. behind every store in flight), and the stores go
    // out as global stores (stg).
    typedef const __attribute__((address_space(4))) PulsePokerView* ViewInKernarg;
    ViewInKernarg vk = (ViewInKernarg)__builtin_amdgcn_kernarg_segment_ptr();       // the view is the first kernel argument
    if (MULTI) asm volatile("" : "+s"(vk));
#define VS(field) (MULTI ? vk->field : v.field)
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int seat = SEAT(k);
        if (seat < P && ((cells_dirty >> k) & 1u)) {
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN)) stg(VS(stacks), ROW_OFF(k), stack[k]);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE | PULSE_PH_CLEARDONE)) stg(VS(current_round_bet), ROW_OFF(k), bet[k]);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_CLEARDONE)) stg(VS(total_invested), ROW_OFF(k), inv[k]);
            if (PH & PULSE_PH_EXECUTE) stg(VS(status), ROW_OFF(k), status[k]);
        }
    }
    if ((PH & PULSE_PH_EQUITY) && eq_dirty) {          // the equities as the last step that recomputed them left them (PokerGPU.py:455-525)
#pragma unroll
        for (int k = 0; k < SPL; ++k) if (SEAT(k) < A) stg(VS(equities), (eq0 + (uint32_t)SEAT(k)) * 4u, eq[k]);
    }
    if (POLICY && act_dirty && j == 0) stg(MULTI ? *(int64_t* const __attribute__((address_space(4)))*)((const __attribute__((address_space(4))) char*)vk + sizeof(PulsePokerView)) : actions, ut * 8u, (int64_t)act64);
    if (PH & (PULSE_PH_ADVANCE | PULSE_PH_SHOWDOWN)) {
        if (board_dirty) {
            static_assert(LPT == 4 || LPT == 2, "board store");
            if (LPT == 4) {
                stg(VS(board), bo + (uint32_t)j * 4u, j == 0 ? b0 : j == 1 ? b1 : j == 2 ? b2 : b3);
                if (j == 0) stg(VS(board), bo + 16u, b4);
            } else {
                stg(VS(board), bo + (uint32_t)j * 4u, j == 0 ? b0 : b1);
                stg(VS(board), bo + 8u + (uint32_t)j * 4u, j == 0 ? b2 : b3);
                if (j == 0) stg(VS(board), bo + 16u, b4);
            }
        }
    }
    if (j == 0) {
        if (PH & PULSE_PH_CAPTURE) { stg(VS(prev_stacks), so, prev_stack); stg(VS(prev_invested), so, prev_invested); }
        if (bet_dirty) {
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_FOLDWIN | PULSE_PH_SHOWDOWN)) stg(VS(pots), so, pot);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE | PULSE_PH_CLEARDONE)) stg(VS(highest), so, highest);
            if (PH & (PULSE_PH_EXECUTE | PULSE_PH_ADVANCE)) { stg(VS(agg), so, agg); stg(VS(acted), so, acted); stg(VS(last_raise_size), so, lrs); }
            if (PH & PULSE_PH_ADVANCE) stg(VS(idx), so, idx);
        }
        if (street_dirty) {
            if (PH & (PULSE_PH_ADVANCE | PULSE_PH_SHOWDOWN)) { stg(VS(stages), so, stage); stg(VS(deck_positions), so, dpos); }
            if (PH & (PULSE_PH_EQUITY | PULSE_PH_ADVANCE)) stg(VS(equity_dirty), ut, (uint8_t)(dirty ? 1 : 0));
        }
    }
    if (POLICY && pa.wave_done) {
        // the roll-out's stop rule (trainGPU.py:27-33) rides on the launch: every wavefront stores how many of its
        // tables are done -- a plain store, summed on a side stream (atomics onto shared counters cost this launch
        // as much as the separate counting kernel they would replace)
        const int c = __popcll(__ballot(done && j == 0));
        if ((threadIdx.x & 63) == 0) pa.wave_done[blockIdx.x * (BLK / 64) + (threadIdx.x >> 6)] = (uint32_t)c;
    }
    STAMP(10);  // state stores issued
#if PULSE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);                      // vmcnt(0): all stores acknowledged
    STAMP(11);
#endif
#undef SEAT_BITS
#undef SEAT_PICK
#undef ROW_OFF
#undef SEAT
#undef VS
}

template <uint32_t PH, bool POLICY, int LPT, int SPL, bool WOBS, int MULTI>
__global__ __launch_bounds__(kStepBlock, LPT == 4 ? 4 : 2) void poker_step_kernel(const PulsePokerView v, int64_t* __restrict__ actions,
                                                           const int32_t* __restrict__ actor_idx_in,
                                                           float* __restrict__ rewards, const PolicyArgs pa, const ChunkArgs ca) {
    poker_step_body<PH, POLICY, LPT, SPL, WOBS, MULTI, false>(v, actions, actor_idx_in, rewards, pa, ca, nullptr);
}

// the learner's action selection + scripted opponents + env step of 128 tables per workgroup (the body's ACT form); the
// leading arguments are the step kernel's (the body re-reads some of them from the kernel-argument segment by position)
template <bool WOBS>
__global__ __launch_bounds__(256, 2) void poker_act_step_kernel(const PulsePokerView v, int64_t* __restrict__ actions,
                                                                const int32_t* __restrict__ actor_idx_in, float* __restrict__ rewards,
                                                                const PolicyArgs pa, const ChunkArgs ca, const pulse_qnet::QNetArgs qa) {
    poker_step_body<PULSE_PH_STEP, true, 2, 5, WOBS, 1, true>(v, actions, actor_idx_in, rewards, pa, ca, &qa);
}

// ---------------------------------------------------------------- host side
// Lanes per table.  Single-step launches: four.  Chunk launches: two where a lane can hold the table's seats in five
// (max_players <= 10; PULSE_VIEW_FOUR_LANES asks for four), four otherwise.  Two lanes per table replicate the table's
// scalar state machine half as often and issue half the wavefronts; measured per 5-step chunk (4 vs 2 lanes) at
// 65,536 / 262,144 / 1,048,576 tables: 39.6 / 112.4 / 404 us vs 37.3 / 114.9 / 395 us.  (With this round's first chunk
// kernel -- per-seat dword loads, library tanh, policy recomputed from the cards every step -- two lanes lost at every
// size, 46.9 vs 43.6 us and 553 vs 463 us: what it saves is exactly the replicated table-level work those changes cut.
// Tables of 12 / 16 seats at six / eight seats per lane were measured too and lose: 42.7 vs 42.5 and 72.6 vs 59.2 us at
// 65,536 tables, 277 vs 233 and 391 vs 317 us at 524,288 -- registers (168 / 187) and LDS per wavefront grow with the seats.)
inline int lanes_for(const PulsePokerView& v, bool chunk) { return chunk && v.max_players <= 10 && !(v.flags & PULSE_VIEW_FOUR_LANES) ? 2 : 4; }
inline dim3 step_grid(const PulsePokerView& v, int lpt) { return dim3((unsigned)(((long long)v.n_games * lpt + kStepBlock - 1) / kStepBlock)); }
inline bool obs_staging(const PulsePokerView& v, const float* obs_odd, int lpt) {
    return !(v.flags & PULSE_VIEW_NO_OBS_STAGING) && (v.n_games % (64 / lpt)) == 0 && ((uintptr_t)v.obs & 15u) == 0 && ((uintptr_t)obs_odd & 15u) == 0;
}

template <uint32_t PH, bool POLICY, int LPT, int SPL, int MULTI>
void launch_one(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, const PolicyArgs& pa, const ChunkArgs& ca,
                hipStream_t st) {
    dim3 grid = step_grid(v, LPT);
    const dim3 block(kStepBlock);
    constexpr int TPW = 64 / LPT;
    const bool wobs = (PH & PULSE_PH_OBS) && (MULTI || PH == PULSE_PH_STEP) && obs_staging(v, MULTI ? ca.obs_odd : nullptr, LPT);
    const size_t lds = MULTI ? sizeof(int32_t) * (size_t)(kStepBlock / 64) * (size_t)chunk_lds_dwords(v.obs_size, LPT * SPL, TPW, MULTI == 2)
                             : (wobs ? sizeof(float) * (size_t)(kStepBlock / 64) * TPW * (size_t)v.obs_size : 0);
    constexpr bool W = PH == PULSE_PH_STEP;          // only the full step is instantiated with observation staging
    if (lds > 48 * 1024) {                           // beyond the default dynamic-LDS limit: raise it (to what this launch needs)
        static size_t raised[2] = {0, 0};
        if (raised[wobs ? 1 : 0] < lds) {
            const void* fn = wobs ? reinterpret_cast<const void*>(&poker_step_kernel<PH, POLICY, LPT, SPL, W, MULTI>)
                                  : reinterpret_cast<const void*>(&poker_step_kernel<PH, POLICY, LPT, SPL, false, MULTI>);
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) raised[wobs ? 1 : 0] = lds;
            else (void)hipGetLastError();            // the launch below reports what is wrong, not this call's sticky error
        }
    }
    if (wobs) hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, LPT, SPL, W, MULTI>), grid, block, lds, st, v, actions, actor_idx, rewards, pa, ca);
    else hipLaunchKernelGGL((poker_step_kernel<PH, POLICY, LPT, SPL, false, MULTI>), grid, block, lds, st, v, actions, actor_idx, rewards, pa, ca);
}

template <uint32_t PH, bool POLICY, int MULTI>
void launch_any(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, const PolicyArgs& pa, const ChunkArgs& ca,
                hipStream_t st) {
    // seats per lane must cover max_players (the observation's padding slots too)
    if (v.max_players <= 12) launch_one<PH, POLICY, 4, 3, MULTI>(v, actions, actor_idx, rewards, pa, ca, st);
    else launch_one<PH, POLICY, 4, 4, MULTI>(v, actions, actor_idx, rewards, pa, ca, st);
}

template <uint32_t PH, bool POLICY>
void launch_step(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, const PolicyArgs& pa, hipStream_t st) {
    launch_any<PH, POLICY, 0>(v, actions, actor_idx, rewards, pa, ChunkArgs{nullptr, nullptr, 1}, st);
}

void launch_chunk(const PulsePokerView& v, int64_t* actions, float* rewards_even, const PolicyArgs& pa, const ChunkArgs& ca, hipStream_t st) {
    if (lanes_for(v, true) == 2) {
        // The full LDS image lets 10 two-lane wavefronts live on a CU (2.5 per SIMD = 81,920 tables on the 256 CUs of an
        // MI355X), the slim one 12.  Batches with more wavefronts than that take the slim image -- 67.0 vs 68.9 us per chunk
        // at 131,072 tables, 84.9 vs 91.8 at 196,608, 361 vs 403 at 1,048,576; where every wavefront is resident anyway it
        // only costs the river's divisions (37.7 vs 37.2 us at 65,536 tables).
        if (v.n_games > 81920) launch_one<PULSE_PH_STEP, true, 2, 5, 2>(v, actions, nullptr, rewards_even, pa, ca, st);
        else launch_one<PULSE_PH_STEP, true, 2, 5, 1>(v, actions, nullptr, rewards_even, pa, ca, st);
    } else launch_any<PULSE_PH_STEP, true, 1>(v, actions, nullptr, rewards_even, pa, ca, st);
}

template <uint32_t PH>
void launch_phase(const PulsePokerView& v, int64_t* actions, const int32_t* actor_idx, float* rewards, hipStream_t st) {
    launch_step<PH, false>(v, actions, actor_idx, rewards, PolicyArgs{0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 0}, st);
}

}  // namespace

// HIP-event timer owned by the caller: brackets whole roll-out calls on their launch stream
struct PulseTimer {
    static constexpr int kMax = 4096;
    hipEvent_t start[kMax], stop[kMax];
    int launches[kMax], steps[kMax];
    int created = 0, used = 0;
    long long calls = 0;                  // chunks seen by pulse_poker_rollout_until (a bracket opens every time_every-th)
    bool open = false;                    // a bracket is open: start recorded, stop not yet
    int open_launches = 0, open_steps = 0;
};

namespace {
constexpr int kTimedSpan = 8;             // consecutive chunks per event pair (an episode has at most eight): the pair's own queue time is shared
int timer_begin(PulseTimer* tm, hipStream_t st) {
    if (tm->used >= PulseTimer::kMax) return 0;
    if (tm->used >= tm->created) {
        if (hipEventCreate(&tm->start[tm->created]) != hipSuccess || hipEventCreate(&tm->stop[tm->created]) != hipSuccess)
            return pulse::fail(PULSE_ENODEVICE, "roll-out timer: hipEventCreate failed");
        ++tm->created;
    }
    const hipError_t e = hipEventRecord(tm->start[tm->used], st);
    if (e != hipSuccess) return pulse::fail_hip((int)e, "roll-out timer: hipEventRecord");
    tm->open = true; tm->open_launches = 0; tm->open_steps = 0;
    return 0;
}
void timer_end(PulseTimer* tm, hipStream_t st) {
    (void)hipEventRecord(tm->stop[tm->used], st);
    tm->launches[tm->used] = tm->open_launches; tm->steps[tm->used] = tm->open_steps; ++tm->used;
    tm->open = false;
}
}  // namespace

extern "C" {

int pulse_poker_step(const PulsePokerView* v, const int64_t* actions, float* rewards, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_step")) return rc;
    if (!actions || !rewards) return pulse::fail(PULSE_EINVAL, "pulse_poker_step: null actions/rewards");
    if (v->n_games == 0) return 0;
    launch_phase<PULSE_PH_STEP>(*v, const_cast<int64_t*>(actions), nullptr, rewards, (hipStream_t)stream);
    return pulse::finish_launch("pulse_poker_step");
}

int pulse_poker_policy_step(const PulsePokerView* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                            uint64_t table_id0, int64_t* actions, float* rewards, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_policy_step")) return rc;
    if (!actions || !rewards || !agent_types) return pulse::fail(PULSE_EINVAL, "pulse_poker_policy_step: null argument");
    if (v->n_games == 0) return 0;
    const PolicyArgs pa{pulse::pack_types(agent_types, v->n_players), seed, step_counter, table_id0, nullptr, nullptr, 0, nullptr, 0};
    launch_step<PULSE_PH_STEP, true>(*v, actions, nullptr, rewards, pa, (hipStream_t)stream);
    return pulse::finish_launch("pulse_poker_policy_step");
}

int pulse_poker_phases(const PulsePokerView* v, uint32_t phases, const int64_t* actions, const int32_t* actor_idx,
                       float* rewards, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_phases")) return rc;
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
    return pulse::finish_launch("pulse_poker_phases");
}

/* Diagnostic (tools/ablate_step.py): the fused policy+step with some phases compiled out.
 * Results are NOT a valid transition; used only to price phases. */
int pulse_poker_ablate(const PulsePokerView* v, uint32_t phases, int64_t* actions, float* rewards, uint64_t types_packed,
                       uint64_t step_counter, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_ablate")) return rc;
    if (v->max_players > 12) return pulse::fail(PULSE_EINVAL, "pulse_poker_ablate: max_players <= 12");
    const dim3 grid = step_grid(*v, 4), block(kStepBlock);
    const PolicyArgs pa{types_packed, 1, step_counter, 0, nullptr, nullptr, 0, nullptr, 0};
    const ChunkArgs ca{nullptr, nullptr, 1};
    hipStream_t st = (hipStream_t)stream;
#define PULSE_ABL(MASK) case (MASK): hipLaunchKernelGGL((poker_step_kernel<(MASK), true, 4, 3, false, false>), grid, block, 0, st, *v, actions, (const int32_t*)nullptr, rewards, pa, ca); break;
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
    return pulse::finish_launch("pulse_poker_ablate");
}

#if PULSE_STAMPS
int pulse_debug_set_stamp_buffer(unsigned long long* buf) {
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    return e == hipSuccess ? 0 : pulse::fail_hip((int)e, "pulse_debug_set_stamp_buffer");
}
#endif

/* ---- roll-out: n_steps fused policy+step transitions enqueued by one native call --------------------------- */
int pulse_timer_create(void** out) {
    if (!out) return pulse::fail(PULSE_EINVAL, "pulse_timer_create: null argument");
    *out = new PulseTimer();
    return 0;
}

int pulse_timer_destroy(void* timer) {
    PulseTimer* tm = static_cast<PulseTimer*>(timer);
    if (!tm) return 0;
    for (int i = 0; i < tm->created; ++i) { (void)hipEventDestroy(tm->start[i]); (void)hipEventDestroy(tm->stop[i]); }
    delete tm;
    return 0;
}

int pulse_timer_collect(void* timer, float* sum_ms, int32_t* n_launches, int64_t* n_steps) {
    PulseTimer* tm = static_cast<PulseTimer*>(timer);
    if (!tm || !sum_ms || !n_launches || !n_steps) return pulse::fail(PULSE_EINVAL, "pulse_timer_collect: null argument");
    float total = 0.0f; int launches = 0; long long steps = 0;
    for (int i = 0; i < tm->used; ++i) {
        float ms = 0.0f;
        const hipError_t e = hipEventElapsedTime(&ms, tm->start[i], tm->stop[i]);
        if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_timer_collect (call it after a stream sync)");
        total += ms; launches += tm->launches[i]; steps += tm->steps[i];
    }
    *sum_ms = total; *n_launches = launches; *n_steps = steps;
    tm->used = 0;
    return 0;
}

namespace {
// one chunk launch of a paired sequence (pulse_internal.h: StopRulePair)
void launch_pair(const PulsePokerView& v_even, const PulsePokerView& v_odd, uint64_t packed, uint64_t seed, uint64_t step_counter0,
                 uint64_t table_id0, int64_t* actions, float* rewards_even, float* rewards_odd, int n_steps, int chunk_steps,
                 const pulse::StopRulePair& plan, hipStream_t st) {
    PolicyArgs pa{packed, seed, step_counter0, table_id0, plan.wave_done_fin, plan.carry[0].partials, plan.carry[0].n, plan.carry[0].host, plan.carry[0].seq};
    pa.carry2_partials = plan.carry[1].partials; pa.carry2_n = plan.carry[1].n; pa.carry2_host = plan.carry[1].host; pa.carry2_seq = plan.carry[1].seq;
    if (pa.carry_n == 0 && pa.carry2_n > 0) {        // (cannot happen: carries are filled in order; kept for safety)
        pa.carry_partials = pa.carry2_partials; pa.carry_n = pa.carry2_n; pa.carry_host = pa.carry2_host; pa.carry_seq = pa.carry2_seq; pa.carry2_n = 0;
    }
    pa.wave_done_mid = plan.wave_done_mid; pa.mid_step = plan.n_chunks == 2 ? chunk_steps : 0;
    pa.verdict_host = plan.verdict_host; pa.verdict_dev = plan.verdict_dev; pa.verdict_id = plan.launch_id; pa.verdict_err = plan.verdict_err;
    pa.verdict_ticks = plan.wait_ticks;
    const ChunkArgs ca{v_odd.obs, rewards_odd, n_steps};
    launch_chunk(v_even, actions, rewards_even, pa, ca, st);
}
}  // namespace

int pulse_poker_rollout(const PulsePokerView* v_even, const PulsePokerView* v_odd, const uint8_t* agent_types,
                        uint64_t seed, uint64_t step_counter0, uint64_t table_id0, int64_t* actions, float* rewards_even,
                        float* rewards_odd, int32_t n_steps, void* timer, void* stoprule, void* stream) {
    if (int rc = pulse::check_view(v_even, "pulse_poker_rollout")) return rc;
    if (int rc = pulse::check_view(v_odd, "pulse_poker_rollout")) return rc;
    if (!actions || !rewards_even || !rewards_odd || !agent_types || n_steps < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_rollout: bad argument");
    if (v_even->is_done != v_odd->is_done_out || v_even->is_done_out != v_odd->is_done || v_even->n_games != v_odd->n_games)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_rollout: v_odd must be v_even with is_done / is_done_out swapped");
    if (v_even->n_games == 0 || n_steps == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t packed = pulse::pack_types(agent_types, v_even->n_players);
    // one step is what the single-step kernel is for (15.4 vs 18.1 us at 65,536 tables: no LDS staging to amortise)
    const bool chunk = !(v_even->flags & PULSE_VIEW_NO_CHUNK) && n_steps > 1;
    PulseTimer* tm = static_cast<PulseTimer*>(timer);
    const bool timed = tm && !tm->open && tm->used < PulseTimer::kMax;
    if (timed) if (int rc = timer_begin(tm, st)) return rc;
    PulseStopRule* rule = static_cast<PulseStopRule*>(stoprule);
    const int n_waves = (int)(((long long)v_even->n_games * lanes_for(*v_even, chunk) + 63) / 64);
    uint32_t* wave_done = nullptr;
    pulse::StopRuleCarry carry{nullptr, 0, nullptr, 0};
    if (rule) if (int rc = pulse::stoprule_claim(rule, n_waves, &wave_done, &carry)) return rc;
    if (chunk) {
        const PolicyArgs pa{packed, seed, step_counter0, table_id0, wave_done, carry.partials, carry.n, carry.host, carry.seq};
        ChunkArgs ca{v_odd->obs, rewards_odd, n_steps};
        // (streaming, non-temporal observation stores were measured: -1 % per chunk up to 262,144 tables, +4 % at 1 M --
        // and +50 % fabric write traffic, since ordinary stores to the two ping-pong blocks are largely absorbed by the
        // caches.  Not used.)
        launch_chunk(*v_even, actions, rewards_even, pa, ca, st);
    } else {
        for (int i = 0; i < n_steps; ++i) {
            const PulsePokerView& v = (i & 1) ? *v_odd : *v_even;
            float* rw = (i & 1) ? rewards_odd : rewards_even;
            PolicyArgs pa{packed, seed, step_counter0 + (uint64_t)i, table_id0, i == n_steps - 1 ? wave_done : nullptr, nullptr, 0, nullptr, 0};
            if (i == 0) { pa.carry_partials = carry.partials; pa.carry_n = carry.n; pa.carry_host = carry.host; pa.carry_seq = carry.seq; }
            launch_step<PULSE_PH_STEP, true>(v, actions, nullptr, rw, pa, st);
        }
    }
    if (timed) { tm->open_launches = chunk ? 1 : n_steps; tm->open_steps = n_steps; timer_end(tm, st); }
    if (int rc = pulse::finish_launch("pulse_poker_rollout")) return rc;
    if (rule) return pulse::stoprule_commit(rule, n_waves, st);
    return 0;
}

/* The trainer's step with the learner in it, first half, as ONE launch: pulse_qnet_act_select on the
 * observation `act->states` followed by pulse_poker_policy_step on `v` -- same results, word for word (the workgroup that
 * picks the actions of a window of 128 tables steps those tables itself). */
int pulse_poker_act_policy_step(const PulsePokerView* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter, uint64_t table_id0,
                                int64_t* actions, float* rewards, const PulseQNet* net, const PulseQNetAct* act, void* stoprule, void* stream) {
    if (int rc = pulse::check_view(v, "pulse_poker_act_policy_step")) return rc;
    if (!actions || !rewards || !agent_types || !net || !act) return pulse::fail(PULSE_EINVAL, "pulse_poker_act_policy_step: null argument");
    if (v->n_games == 0) return 0;
    const int n = v->n_games;
    if (n % 128 != 0 || v->max_players > 10 || (v->flags & PULSE_VIEW_FOUR_LANES))
        return pulse::fail(PULSE_EINVAL, "pulse_poker_act_policy_step: needs a multiple of 128 tables of at most 10 seats (use pulse_qnet_act_select + pulse_poker_policy_step)");
    if (net->state_dim < 13 || net->state_dim > 40 || net->state_dim % 8 != 0 || net->n_actions < 1 || net->n_actions > 32 || act->row_stride % 4 != 0 ||
        act->row_stride < net->state_dim || ((uintptr_t)act->states & 15u) || ((uintptr_t)net->w1 & 15u) || ((uintptr_t)net->w2 & 15u) || ((uintptr_t)net->w3 & 15u) ||
        ((uintptr_t)net->w4 & 15u) || ((uintptr_t)net->w5 & 15u))
        return pulse::fail(PULSE_EINVAL, "pulse_poker_act_policy_step: needs 16-byte aligned fp32 rows of 16..40 inputs (a multiple of 8) and aligned weights");
    if (!net->w1 || !net->b1 || !net->w2 || !net->b2 || !net->w3 || !net->b3 || !net->w4 || !net->b4 || !net->w5 || !net->b5 || !act->states || !act->seat_idx || !act->row_mask_out)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_act_policy_step: null weight / state / seat_idx / row_mask_out pointer");
    if (!act->select_scratch || act->select_words < (int64_t)((n + 255) / 256) * 259 + 512)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_act_policy_step: select_scratch must hold 259 words per 256 rows + 512");
    hipStream_t st = (hipStream_t)stream;
    pulse_qnet::QNetArgs qa{};
    qa.net = *net; qa.states = act->states; qa.row_stride = act->row_stride; qa.n_rows = n; qa.seat_idx = act->seat_idx; qa.q_seat = act->q_seat;
    qa.epsilon = act->epsilon; qa.seed = act->seed; qa.step = act->step; qa.table_id0 = act->table_id0; qa.actions = actions; qa.q_out = nullptr;
    qa.terminated = act->terminated; qa.row_mask_out = act->row_mask_out;
    qa.tsel_rows = act->select_scratch; qa.tsel_counts = act->select_scratch + (size_t)((n + 255) / 256) * 256;
    PulseStopRule* rule = static_cast<PulseStopRule*>(stoprule);
    const int n_waves = n * 2 / 64;
    uint32_t* wave_done = nullptr;
    pulse::StopRuleCarry carry{nullptr, 0, nullptr, 0};
    if (rule) if (int rc = pulse::stoprule_claim(rule, n_waves, &wave_done, &carry)) return rc;
    const PolicyArgs pa{pulse::pack_types(agent_types, v->n_players), seed, step_counter, table_id0, wave_done, carry.partials, carry.n, carry.host, carry.seq};
    const ChunkArgs ca{v->obs, rewards, 1};
    const bool wobs = obs_staging(*v, v->obs, 2);
    const size_t lds = std::max(pulse_qnet::kActLdsBytes, sizeof(int32_t) * 4 * (size_t)chunk_lds_dwords(v->obs_size, 10, 32, false));
    const void* fn = wobs ? reinterpret_cast<const void*>(&poker_act_step_kernel<true>) : reinterpret_cast<const void*>(&poker_act_step_kernel<false>);
    static size_t raised[2] = {0, 0};
    if (raised[wobs ? 1 : 0] < lds) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_poker_act_policy_step: LDS size attribute");
        raised[wobs ? 1 : 0] = lds;
    }
    const dim3 grid((unsigned)(n / 128)), block(256);
    if (wobs) hipLaunchKernelGGL(poker_act_step_kernel<true>, grid, block, lds, st, *v, actions, (const int32_t*)nullptr, rewards, pa, ca, qa);
    else hipLaunchKernelGGL(poker_act_step_kernel<false>, grid, block, lds, st, *v, actions, (const int32_t*)nullptr, rewards, pa, ca, qa);
    if (int rc = pulse::finish_launch("pulse_poker_act_policy_step")) return rc;
    if (rule) return pulse::stoprule_commit(rule, n_waves, st);
    return 0;
}

int pulse_poker_rollout_until(const PulsePokerView* v_even, const PulsePokerView* v_odd, const uint8_t* agent_types,
                              uint64_t seed, uint64_t step_counter0, uint64_t table_id0, int64_t* actions, float* rewards_even,
                              float* rewards_odd, int32_t chunk_steps, int32_t max_steps, void* timer, int32_t time_every,
                              void* stoprule, void* stream, int32_t* steps_done, int32_t* over) {
    if (!steps_done || !over || chunk_steps <= 0 || max_steps < 0 || !stoprule)
        return pulse::fail(PULSE_EINVAL, "pulse_poker_rollout_until: bad argument");
    if (int rc = pulse::check_view(v_even, "pulse_poker_rollout_until")) return rc;
    if (int rc = pulse::check_view(v_odd, "pulse_poker_rollout_until")) return rc;
    if (!actions || !rewards_even || !rewards_odd || !agent_types) return pulse::fail(PULSE_EINVAL, "pulse_poker_rollout_until: null argument");
    PulseTimer* tm = static_cast<PulseTimer*>(timer);
    hipStream_t st = (hipStream_t)stream;
    const bool per_step = (v_even->flags & PULSE_VIEW_NO_CHUNK) != 0;
    PulseStopRule* rule = static_cast<PulseStopRule*>(stoprule);
    int done = 0, parity = 0, verdict = 0;
    bool fell_back = false;
    // ---- paired launches: with the lag-1 rule ONE launch runs up to two check intervals and takes the rule's verdicts on
    // the two check points before them itself (pulse_internal.h: StopRulePair) -- half as many state load bursts and
    // store tails per episode, the same episodes step for step.  (PULSE_VIEW_NO_PAIRS / lag 0 / lag 2 / RCCL: one check
    // interval per launch, below.)
    const int n_waves_chunk = (int)(((long long)v_even->n_games * lanes_for(*v_even, true) + 63) / 64);
    const bool pairs = !per_step && !(v_even->flags & PULSE_VIEW_NO_PAIRS) && chunk_steps > 1 && v_even->n_games > 0 &&
                       pulse::stoprule_pairs_supported(rule, n_waves_chunk);
    if (pairs) {
        const uint64_t packed = pulse::pack_types(agent_types, v_even->n_players);
        while (done < max_steps && !verdict) {
            const int left = max_steps - done;
            const int k = left >= 2 * chunk_steps ? 2 : 1;
            const int n = k == 2 ? 2 * chunk_steps : (left < chunk_steps ? left : chunk_steps);
            pulse::StopRulePair plan;
            const int c = pulse::stoprule_pair_claim(rule, n_waves_chunk, k, &plan);
            if (c < 0) return c;
            if (c == 1) { verdict = 1; break; }
            if (tm && time_every > 0 && !tm->open && (tm->calls % time_every) == 0) if (int rc = timer_begin(tm, st)) return rc;
            if (tm) ++tm->calls;
            launch_pair(parity ? *v_odd : *v_even, parity ? *v_even : *v_odd, packed, seed, step_counter0 + (uint64_t)done, table_id0, actions,
                        parity ? rewards_odd : rewards_even, parity ? rewards_even : rewards_odd, n, chunk_steps, plan, st);
            if (int rc = pulse::finish_launch("pulse_poker_rollout_until")) return rc;
            if (int rc = pulse::stoprule_pair_commit(rule, &plan, n_waves_chunk, st)) return rc;
            int chunks_run = 0, gave_up = 0;
            if (int rc = pulse::stoprule_pair_verdict(rule, &plan, &chunks_run, &verdict, &gave_up)) return rc;
            if (gave_up) {                  // the host was too late for this launch: it ran nothing; the rule pairs no more
                fell_back = !verdict;       // (unless the episode had ended before it anyway) its steps run below, one check interval per launch
                break;
            }
            const int ran = chunks_run == k ? n : chunks_run * chunk_steps;
            done += ran; parity ^= ran & 1;
            if (tm && tm->open) {
                tm->open_launches += 1; tm->open_steps += ran;
                if (tm->open_launches >= kTimedSpan) timer_end(tm, st);
            }
        }
        if (!fell_back) {
            if (tm && tm->open) timer_end(tm, st);
            *steps_done = done; *over = verdict;
            return 0;
        }
    }
    while (done < max_steps && !verdict) {
        const int n = chunk_steps < max_steps - done ? chunk_steps : max_steps - done;
        // an event pair brackets kTimedSpan consecutive chunks, every time_every-th chunk opens one
        if (tm && time_every > 0 && !tm->open && (tm->calls % time_every) == 0) if (int rc = timer_begin(tm, st)) return rc;
        if (tm) ++tm->calls;
        // after an odd number of steps the roles of the two views (and reward buffers) are swapped
        if (int rc = pulse_poker_rollout(parity ? v_odd : v_even, parity ? v_even : v_odd, agent_types, seed, step_counter0 + (uint64_t)done,
                                         table_id0, actions, parity ? rewards_odd : rewards_even, parity ? rewards_even : rewards_odd, n,
                                         nullptr, stoprule, stream)) return rc;
        done += n; parity ^= n & 1;
        if (tm && tm->open) {
            tm->open_launches += per_step ? n : 1; tm->open_steps += n;
            if (tm->open_launches >= (per_step ? kTimedSpan * chunk_steps : kTimedSpan)) timer_end(tm, st);
        }
        if (int rc = pulse_stoprule_decide(stoprule, &verdict)) return rc;
    }
    if (tm && tm->open) timer_end(tm, st);            // the episode ended inside a bracket: it covers what ran
    *steps_done = done; *over = verdict;
    return 0;
}

}  // extern "C"
