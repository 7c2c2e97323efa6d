// qtable.hip -- tabular Q-learning for the batched 2048 roll-out (BASELINE.json config 3).
//
// The reference keeps `defaultdict(state tuple -> float64[n])` in Python and calls two numba scalars per
// step and per (single) board: utils/numba.py:5-21 (epsilon-greedy argmax) and :25-39 (Q update), driven
// by agents/TemperalDifference/QLearningNumba.py:10-37.  Here B boards act and learn per launch:
//   * the dict is an open-addressing hash table in HBM of 64-byte ENTRIES {uint64 key = the board packed as 4-bit log2
//     tiles, double q[4], 24 spare bytes}: a state's key and values share one memory line, so a lookup costs one random
//     line instead of two (round 2: keys and values in separate arrays); linear probing, lock-free insert by atomicCAS;
//   * pulse_qtable_select: look up / insert the state, epsilon-greedy over its four doubles
//     (first maximal index, as numba.py:13-19), Philox draws keyed by (seed, board id, step);
//   * pulse_qtable_update: q[s][a] += alpha * (target - q[s][a]), target = r or r + gamma * max q[s'];
//   * pulse_qtable_rollout_step: select + the 2048 move (tfe_device.h) + update in ONE launch -- the board stays in
//     registers, s' found by the update is carried to the next step's select: two random lines per board-step.
// `region_slots` > 0 gives every board a private region of the table (B independent learners = B copies of
// the reference, bit-exact and race-free: the parity mode); 0 shares one table between all boards.
//
// Shared table, several boards updating ONE entry in the same launch (all boards leave reset from a few hundred two-tile
// states: ~1,000 contenders per entry).  A CAS retry loop is quadratic there (every round one contender wins, the others
// start over: the update launch took 5.8 ms at the first step, 37 us once the boards had spread).  Now every update tries
// its CAS ONCE; those that lose are deferred and combined: a per-launch scratch hash maps the cell to an accumulator,
// every deferred transition adds its target with plain atomic adds (linear), and one thread per cell applies
//     q <- q + (1 - (1 - alpha)^k) * (mean of the k targets - q)
// -- the k transitions applied one after another with the mean of their targets (k = 1: the reference's formula itself).
#include <hip/hip_runtime.h>

#include "pulse_internal.h"
#include "tfe_device.h"

namespace {

constexpr int kBlock = 256;

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;   // (v_mad_u64_u32)
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// one table entry = one 64-byte line
struct alignas(64) Entry { unsigned long long key; double q[4]; unsigned long long spare[3]; };
static_assert(sizeof(Entry) == PULSE_QTABLE_ENTRY_BYTES, "entry layout is part of the ABI");

// board -> key: 4 bits of log2(tile) per cell (0 = empty), row-major, cell 0 in the low nibble.
__device__ __forceinline__ uint64_t pack_cells(const int* b, int cells) {
    uint64_t key = 0;
    for (int i = 0; i < cells; ++i) {
        const int v = b[i];
        const uint64_t e = v > 0 ? (uint64_t)min(31 - __clz(v), 15) : 0ull;
        key |= e << (4 * i);
    }
    return key;
}
__device__ __forceinline__ uint64_t pack_board(const int32_t* __restrict__ b, int cells) {
    uint64_t key = 0;
    for (int i = 0; i < cells; ++i) {
        const int v = b[i];
        const uint64_t e = v > 0 ? (uint64_t)min(31 - __clz(v), 15) : 0ull;
        key |= e << (4 * i);
    }
    return key;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// Slot of `key` inside [base, base + slots): inserted (value row already zero) if absent.  -1 = no room.
// At most kMaxProbe slots are examined: a table filled to the brim would otherwise turn every lookup of every
// thread into a walk over the whole table (262,144 threads x 2^24 slots: a launch that never ends); past the
// limit the state counts as "no room" like a full region.
constexpr uint64_t kMaxProbe = 4096;
// (Tried and dropped, DESIGN.md section 3.4: a "shared by several boards" hint kept in the entry's spare words by plain loads
// and stores at the lookup of the state a move led to, so that the next launch sends the updates of a hot entry straight to the
// combine path instead of racing for a compare-and-swap -- 233 -> 140 us at the second step after a reset, but +8 us at EVERY
// step for dirtying the looked-up line, and the racy visitor count rarely passed 8 across the eight L2s.)
__device__ __forceinline__ long long find_or_insert(Entry* table, uint64_t base, uint64_t slots, uint64_t key) {
    uint64_t h = mix64(key) & (slots - 1);
    const uint64_t limit = slots < kMaxProbe ? slots : kMaxProbe;
    for (uint64_t probe = 0; probe < limit; ++probe) {
        const uint64_t s = base + ((h + probe) & (slots - 1));
        unsigned long long cur = table[s].key;
        if (cur == key) return (long long)s;
        if (cur == 0ull) {
            cur = atomicCAS(&table[s].key, 0ull, (unsigned long long)key);
            if (cur == 0ull || cur == key) return (long long)s;
        }
    }
    return -1;
}

// epsilon-greedy (numba.py:5-21): p from words x, y of the board's Philox call, the random action from word z
__device__ __forceinline__ int choose_action(const U4& r, double epsilon, const double* q, bool have_row) {
    const double p = (double)(((uint64_t)r.x << 21) ^ (uint64_t)(r.y >> 11)) * (1.0 / 9007199254740992.0);   // 53-bit uniform
    if (p < epsilon || !have_row) return (int)__umulhi(r.z, 4u);        // numba.py:9-11 random.randint(0, n-1)
    int a = 0; double mx = q[0];
    for (int i = 1; i < 4; ++i) if (q[i] > mx) { mx = q[i]; a = i; }    // numba.py:13-19
    return a;
}

// ---- deferred updates of a shared table (see the header comment) -------------------------------------------------
// The list of a launch's deferred transitions is kept in SEGMENTS, one per workgroup of the launch (slots 256 w .. 256 w + 255
// for workgroup w, its length in count[kCountWg + parity * W + w]): positions come from a counter in LDS, the lengths leave as
// plain stores.  (One list with one global counter: a returning atomic per wavefront on ONE address, ~7 ns apiece one after
// the other -- 28 of the 72 us of a launch while most wavefronts still had a loser, i.e. for the first ~25 steps after a reset.)
constexpr int kCountMeet = 0;       // count[0..8) / [8..16): the follow-up launch's arrival counters, by parity
constexpr int kCountAny = 16;       // count[16 + parity]: != 0 iff some workgroup of the launch deferred something (the follow-up launch's early exit)
constexpr int kCountWg = 64;        // count[64 + parity * W + w]: deferred transitions of workgroup w (W = ceil(n / 256))
struct Deferred {                 // per-launch scratch, owned by the caller (PulseQTableScratch)
    unsigned int* count;          // [64 + 2 W], see above
    unsigned long long* cells;    // [n]: entry index * 4 + action
    double* targets;              // [n]
    int* owner;                   // [n]: accumulator index if this transition claimed its cell, else -1
    unsigned long long* acc_key;  // [acc_slots]: cell + 1 (0 = free)
    unsigned int* acc_cnt;        // [acc_slots]
    double* acc_sum;              // [acc_slots]
    unsigned int acc_slots;       // power of two >= 2 n
    int parity;
    int n_wg;                     // W: workgroups of the launch that filled the list (= segments)
    long long wait_ticks;         // how long a workgroup of the follow-up launch waits at its meeting (100 MHz ticks)
    unsigned int extra_arrivals;  // test hook: arrivals the meeting expects beyond the grid's (never met)
    long long* gave_up;           // pinned host word, += 1 by a follow-up launch whose meeting was called off
    int ablate;                   // timing-only diagnostics (PulseQTableScratch.reserved0; results are then NOT valid updates):
                                  // 1 = no compare-and-swap, every update is deferred; 2 = losers are dropped, not listed; 4 = new states are not inserted
};
constexpr unsigned int kDeferOff = 0x80000000u;   // a meeting counter with this bit: called off

// q[s][a] <- q + alpha (target - q): race-free regions write, shared tables try ONE compare-and-swap and defer on a loss.
// Called by EVERY thread of the workgroup (`live` = this thread has a transition): the losers' list positions come from the
// workgroup's LDS counter `wg_n` (zeroed by the caller before a barrier), the segment's length is stored by finish_deferred.
__device__ __forceinline__ void apply_update(Entry* table, bool live, long long s, int a, double target, double alpha, bool shared_table, const Deferred& d,
                                             unsigned int* wg_n) {
    if (!shared_table) {
        if (!live) return;
        double* cell = &table[s].q[a & 3];
        const double old = *cell;
        *cell = __dadd_rn(old, __dmul_rn(alpha, __dsub_rn(target, old)));                       // numba.py:38-39
        return;
    }
    bool lost = false;
    if (live) {
        unsigned long long* raw = reinterpret_cast<unsigned long long*>(&table[s].q[a & 3]);
        const unsigned long long seen = *raw;
        const double old = __longlong_as_double((long long)seen);
        const double upd = __dadd_rn(old, __dmul_rn(alpha, __dsub_rn(target, old)));
        lost = true;
        if (!(d.ablate & 1)) lost = atomicCAS(raw, seen, (unsigned long long)__double_as_longlong(upd)) != seen;
        if (d.ablate & 2) lost = false;
    }
    const unsigned long long losers = __ballot(lost);
    if (!lost) return;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)losers) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(wg_n, (unsigned int)__popcll(losers));              // LDS
    base = (unsigned int)__shfl((int)base, leader);
    const unsigned int at = blockIdx.x * kBlock + base + (unsigned int)__popcll(losers & ((1ull << lane) - 1ull));
    d.cells[at] = (unsigned long long)s * 4ull + (unsigned long long)(a & 3);
    d.targets[at] = target;
}
// after apply_update, by every thread of the workgroup: the segment's length leaves (also when it is zero)
__device__ __forceinline__ void finish_deferred(const Deferred& d, unsigned int* wg_n) {
    if (!d.count) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int n = *wg_n;
        d.count[kCountWg + d.parity * d.n_wg + blockIdx.x] = n;
        if (n) atomicOr(d.count + kCountAny + d.parity, 1u);          // (no return value: pipelined; in the steady state few workgroups get here)
    }
}

// The deferred transitions of a launch, in ONE follow-up launch of kDeferBlocks workgroups: (1) every transition finds (or
// claims) the accumulator of its cell and adds its target; (2) the transition that claimed a cell applies the combined
// update and frees the accumulator.  Between the phases the workgroups meet at eight counters in the scratch -- they are all
// resident (one per CU at most) -- but ONLY when the list is long: a short one (the steady state: a few popular states are
// still shared by some boards) is ONE workgroup's work with a workgroup barrier between the phases, and with an empty one
// every workgroup sums the segment lengths and leaves.
constexpr int kDeferBlocks = 64;      // (256 workgroups: the follow-up launch took 50 us instead of 18 on the lists of steps 10..60 -- the meeting grows with its members)
constexpr unsigned int kDeferSolo = 1024;
__global__ __launch_bounds__(kBlock) void qtable_defer_kernel(Entry* table, Deferred d, double alpha) {
    // the segments' first item numbers (exclusive prefix sum of their lengths; every workgroup computes it for itself): item i of
    // the launch's list = slot 256 w + (i - seg_start[w]) for the segment w that holds it -- the items are dealt to the threads
    // densely (a thread per segment slot would leave most lanes idle for sixteen dependent rounds: 17 -> 57 us when tried)
    if (blockIdx.x == 0 && threadIdx.x < 9) d.count[threadIdx.x < 8 ? kCountMeet + 8 * (d.parity ^ 1) + threadIdx.x : kCountAny + (d.parity ^ 1)] = 0u;   // the next launch's meeting point and flag
    if (d.count[kCountAny + d.parity] == 0u) return;                          // nothing was deferred (boards spread over distinct states: the usual case)
    __shared__ unsigned int seg_start[4097];
    __shared__ unsigned int wave_tot[kBlock / 64];
    const unsigned int* lens = d.count + kCountWg + d.parity * d.n_wg;
    const int chunk = (d.n_wg + kBlock - 1) / kBlock;                          // <= 16 segments per thread
    unsigned int len[16], mine = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int w = (int)threadIdx.x * chunk + k;
        len[k] = (k < chunk && w < d.n_wg) ? lens[w] : 0u;
        mine += len[k];
    }
    unsigned int incl = mine;                                                   // inclusive scan over the workgroup's threads
    for (int m = 1; m < 64; m <<= 1) { const unsigned int up = (unsigned int)__shfl_up((int)incl, m); if ((int)(threadIdx.x & 63) >= m) incl += up; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned int before = 0, n = 0;
    for (int k = 0; k < kBlock / 64; ++k) { if (k < (int)(threadIdx.x >> 6)) before += wave_tot[k]; n += wave_tot[k]; }
    if (n == 0u) return;
    unsigned int run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int w = (int)threadIdx.x * chunk + k;
        if (k < chunk && w < d.n_wg) seg_start[w] = run;
        run += len[k];
    }
    if (threadIdx.x == 0) seg_start[d.n_wg] = n;
    __syncthreads();
    const bool solo = n <= kDeferSolo;
    if (solo && blockIdx.x != 0) return;
    // solo: the one workgroup takes all items; spread: item i goes to thread i mod (grid x 256)
    auto for_my_items = [&](auto&& body) {
        const unsigned int first = solo ? threadIdx.x : blockIdx.x * kBlock + threadIdx.x, stride = solo ? kBlock : gridDim.x * kBlock;
        for (unsigned int i = first; i < n; i += stride) {
            int lo = 0, hi = d.n_wg;                                            // the last segment that starts at or before item i
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_start[mid] <= i) lo = mid; else hi = mid; }
            body((unsigned int)lo * kBlock + (i - seg_start[lo]));
        }
    };
    for_my_items([&](unsigned int i) {
        const unsigned long long cell = d.cells[i], tag = cell + 1ull;
        unsigned int h = (unsigned int)mix64(cell) & (d.acc_slots - 1u);
        int claimed = -1;
        for (;;) {                                        // acc_slots >= 2 n: a free slot always exists
            unsigned long long cur = d.acc_key[h];
            if (cur == 0ull) { cur = atomicCAS(d.acc_key + h, 0ull, tag); if (cur == 0ull) { claimed = (int)h; break; } }
            if (cur == tag) break;
            h = (h + 1u) & (d.acc_slots - 1u);
        }
        atomicAdd(d.acc_cnt + h, 1u);
        atomicAdd(d.acc_sum + h, d.targets[i]);
        d.owner[i] = claimed;
    });
    __threadfence();
    __syncthreads();
    if (!solo) {
        // All or nothing: phase 2 runs iff all eight counters read exactly full.  A workgroup whose wait runs out marks a counter
        // that is still short (compare-and-swap from the value it read: the mark lands before the missing arrival or not at
        // all), after which that counter never reads full: no owner applies `sum / k` from accumulators other workgroups are
        // still adding to.  The accumulators then stay claimed; the host finds the pinned count changed, clears them and
        // fails its next call.  (Eight counters: arrivals on ONE address cost the meeting ~25 us.)
        __shared__ int go_s;
        if (threadIdx.x < 64) {
            unsigned int* meet = d.count + kCountMeet + 8 * d.parity;
            if (threadIdx.x == 0) atomicAdd(meet + (blockIdx.x & 7), 1u);
            unsigned int want = (gridDim.x + 7 - (threadIdx.x & 7)) / 8;
            if (threadIdx.x == 0) want += d.extra_arrivals;
            int go = 0;
            for (const long long t0 = wall_clock64();;) {        // (100 MHz)
                const unsigned int got = threadIdx.x < 8 ? __hip_atomic_load(meet + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                if (__ballot((got & kDeferOff) != 0u) != 0ull) break;
                const unsigned long long short_of = __ballot(got != want);
                if (short_of == 0ull) { go = 1; break; }
                if (wall_clock64() - t0 > d.wait_ticks) {
                    const int first = __ffsll((long long)short_of) - 1;
                    unsigned int marked = 0u;
                    if ((int)threadIdx.x == first) marked = atomicCAS(meet + threadIdx.x, got, got | kDeferOff) == got ? 1u : 0u;
                    if (__ballot(marked != 0u) != 0ull) {
                        if ((int)threadIdx.x == first && d.gave_up) __hip_atomic_fetch_add(d.gave_up, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    continue;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (threadIdx.x == 0) go_s = go;
        }
        __syncthreads();
        if (!go_s) return;
    }
    __threadfence();
    for_my_items([&](unsigned int i) {
        const int h = d.owner[i];
        if (h < 0) return;
        const unsigned long long cell = d.cells[i];
        const unsigned int k = __hip_atomic_load(d.acc_cnt + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double sum = __hip_atomic_load(d.acc_sum + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double mean = sum / (double)k;
        double* q = &table[cell >> 2].q[cell & 3ull];
        const double old = *q;
        // k = 1: exactly numba.py:38-39; k > 1: the k transitions one after another, each with the mean of their targets
        const double w = k == 1u ? alpha : 1.0 - pow(1.0 - alpha, (double)k);
        *q = __dadd_rn(old, __dmul_rn(w, __dsub_rn(mean, old)));
        d.acc_key[h] = 0ull; d.acc_cnt[h] = 0u; d.acc_sum[h] = 0.0;
    });
}

__global__ __launch_bounds__(kBlock) void qtable_select_kernel(Entry* table, uint64_t capacity, uint64_t region_slots,
                                                              const int32_t* __restrict__ boards, int n_boards, int cells,
                                                              double epsilon, uint64_t seed, uint64_t board_id0, uint64_t step_counter,
                                                              int64_t* __restrict__ actions, int64_t* __restrict__ slots_out) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) return;
    const uint64_t key = pack_board(boards + (size_t)g * cells, cells);
    const uint64_t base = region_slots ? (uint64_t)g * region_slots : 0, slots = region_slots ? region_slots : capacity;
    const long long s = find_or_insert(table, base, slots, key);
    slots_out[g] = s;
    const U4 r = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    if (s >= 0) { q[0] = table[s].q[0]; q[1] = table[s].q[1]; q[2] = table[s].q[2]; q[3] = table[s].q[3]; }
    actions[g] = choose_action(r, epsilon, q, s >= 0);
}

__global__ __launch_bounds__(kBlock) void qtable_update_kernel(Entry* table, uint64_t capacity, uint64_t region_slots,
                                                              const int64_t* __restrict__ slots_s, const int64_t* __restrict__ actions,
                                                              const int32_t* __restrict__ rewards, const int32_t* __restrict__ next_boards,
                                                              const uint8_t* __restrict__ terminal, int n_boards, int cells, double alpha,
                                                              double gamma, Deferred d) {
    __shared__ unsigned int wg_n;
    if (threadIdx.x == 0) wg_n = 0u;
    __syncthreads();
    const int g = blockIdx.x * kBlock + threadIdx.x;
    const long long s = g < n_boards ? slots_s[g] : -1;
    const bool live = s >= 0;
    double target = 0.0; int a = 0;
    if (live) {
        const uint64_t base = region_slots ? (uint64_t)g * region_slots : 0, slots = region_slots ? region_slots : capacity;
        // QLearningNumba.py:28-37 touches q[next_state] even for terminal transitions (defaultdict insert)
        const long long sn = find_or_insert(table, base, slots, pack_board(next_boards + (size_t)g * cells, cells));
        double mx = 0.0;
        if (sn >= 0) {
            mx = table[sn].q[0];
            for (int i = 1; i < 4; ++i) if (table[sn].q[i] > mx) mx = table[sn].q[i];             // numba.py:28-31
        }
        const double reward = (double)rewards[g];
        target = terminal[g] ? reward : __dadd_rn(reward, __dmul_rn(gamma, mx));                 // numba.py:33-36
        a = (int)(actions[g] & 3);
    }
    apply_update(table, live, s, a, target, alpha, region_slots == 0, d, &wg_n);
    finish_deferred(d, &wg_n);
}

// One roll-out step of B learners in one launch: what select -> pulse_tfe_step -> update do in three, with the board in
// registers throughout.  slots_io: in = the entry of every board's current state if a previous step found it (-2 =
// unknown: look it up), out = the entry of the state the move led to.  Draws: the agent's Philox stream (agent_seed,
// board, agent_step) as pulse_qtable_select, the environment's (env_seed, board, env_step) as pulse_tfe_step.
template <int NB>
__global__ __launch_bounds__(kBlock) void qtable_rollout_step_kernel(Entry* table, uint64_t capacity, uint64_t region_slots,
                                                                    int32_t* __restrict__ boards, int64_t* __restrict__ total_score,
                                                                    int n_boards, double epsilon, double alpha, double gamma,
                                                                    uint64_t agent_seed, uint64_t agent_step, uint64_t env_seed, uint64_t env_step,
                                                                    uint64_t board_id0, int64_t* __restrict__ actions, int32_t* __restrict__ rewards,
                                                                    uint8_t* __restrict__ dones, int64_t* __restrict__ slots_io, Deferred d,
                                                                    const uint32_t* __restrict__ lut) {
    using namespace pulse_tfe;
    __shared__ unsigned int wg_n;
    if (threadIdx.x == 0) wg_n = 0u;
    __syncthreads();
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) { apply_update(table, false, -1, 0, 0.0, alpha, region_slots == 0, d, &wg_n); finish_deferred(d, &wg_n); return; }
    int b[NB * NB];
    int32_t* bp = boards + (size_t)g * NB * NB;
#pragma unroll
    for (int i = 0; i < NB * NB; ++i) b[i] = bp[i];
    const uint64_t base = region_slots ? (uint64_t)g * region_slots : 0, slots = region_slots ? region_slots : capacity;
    long long s = slots_io[g];
    // 4 x 4: the board packed to 64 bits (tfe_device.h) IS the state key, and the move / spawn / game-over test run on that word
    // (four row-table lookups instead of ~900 per-cell selects); a wavefront that holds a board the packed form cannot (a tile
    // that is no power of two in 2 .. 16,384) takes the cell form as a whole.
    PackedBoard pb{0u, 0u};
    bool fast = false;
    if (NB == 4 && lut) { const bool ok = tfe_pack4(reinterpret_cast<const int (&)[16]>(b), pb); fast = !__any(!ok); }
    if (s == -2) s = find_or_insert(table, base, slots, fast ? ((uint64_t)pb.hi << 32 | pb.lo) : pack_cells(b, NB * NB));
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    if (s >= 0) { q[0] = table[s].q[0]; q[1] = table[s].q[1]; q[2] = table[s].q[2]; q[3] = table[s].q[3]; }
    const int a = choose_action(philox4x32(agent_seed, board_id0 + (uint64_t)g, agent_step), epsilon, q, s >= 0);
    const U4 rnd = philox4x32(env_seed, board_id0 + (uint64_t)g, env_step);
    int score; bool over; uint64_t key_next;
    if (fast) {
        score = tfe_move_packed(pb, a, lut);
        const int empty_before = tfe_spawn_packed(pb, rnd.x, rnd.y);                    // TFE.py:182 (always)
        over = tfe_over_packed(pb, empty_before);
        key_next = (uint64_t)pb.hi << 32 | pb.lo;
        tfe_unpack4(pb, reinterpret_cast<int (&)[16]>(b));
    } else {
        score = tfe_move<NB>(b, a);
        tfe_spawn<NB>(b, rnd.x, rnd.y);
        over = tfe_over<NB>(b);
        key_next = pack_cells(b, NB * NB);
    }
    const int reward = score > 0 ? 31 - __clz(score) : 0;                                // TFE.py:185-187
#pragma unroll
    for (int i = 0; i < NB * NB; ++i) bp[i] = b[i];
    total_score[g] += score;
    actions[g] = a; rewards[g] = reward; dones[g] = over;
    const long long sn = (d.ablate & 4) ? (long long)(base + (mix64(key_next) & (slots - 1))) : find_or_insert(table, base, slots, key_next);
    slots_io[g] = sn;
    double mx = 0.0;                                                                    // (s < 0: no room for the state: acted at random, learns nothing)
    if (s >= 0 && sn >= 0) {
        mx = table[sn].q[0];
        for (int i = 1; i < 4; ++i) if (table[sn].q[i] > mx) mx = table[sn].q[i];
    }
    const double target = over ? (double)reward : __dadd_rn((double)reward, __dmul_rn(gamma, mx));
    apply_update(table, s >= 0, s, a, target, alpha, region_slots == 0, d, &wg_n);
    finish_deferred(d, &wg_n);
}

int finish_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pulse::fail_hip((int)e, what);
    return 0;
}

int check_table(const PulseQTable* q, int32_t n_boards, int32_t n) {
    if (!q || !q->entries) return pulse::fail(PULSE_EINVAL, "PulseQTable: null table");
    if (((uintptr_t)q->entries & 63u) != 0) return pulse::fail(PULSE_EINVAL, "PulseQTable: entries must be 64-byte aligned");
    if (n < 3 || n > 4) return pulse::fail(PULSE_EINVAL, "PulseQTable: board side must be 3 or 4 (64-bit state key)");
    const uint64_t slots = q->region_slots ? q->region_slots : q->capacity;
    if (slots == 0 || (slots & (slots - 1))) return pulse::fail(PULSE_EINVAL, "PulseQTable: capacity / region_slots must be a power of two");
    if (q->region_slots && (uint64_t)n_boards * q->region_slots > q->capacity)
        return pulse::fail(PULSE_EINVAL, "PulseQTable: capacity < n_boards * region_slots");
    return 0;
}

// A follow-up launch whose workgroups did not gather (a co-tenant on the GPU) drops that launch's combined updates and
// leaves their accumulators claimed: counted in this pinned word; the next call on a shared table clears the accumulators
// it is handed and fails once (PULSE_EINTERNAL), after which the table is usable again.
long long* g_defer_gave_up = nullptr;
long long g_defer_gave_up_seen = 0;

// the shared table's scratch, checked and turned into the kernels' view of it
int deferred_of(const PulseQTable* q, const PulseQTableScratch* sc, int32_t n_boards, uint64_t launch_index, Deferred* d, hipStream_t st) {
    *d = Deferred{};
    if (q->region_slots) return 0;                      // private regions: no race, no scratch
    if (!sc || !sc->count || !sc->cells || !sc->targets || !sc->owner || !sc->acc_key || !sc->acc_cnt || !sc->acc_sum)
        return pulse::fail(PULSE_EINVAL, "PulseQTable: a shared table needs its PulseQTableScratch");
    if (sc->n < (uint32_t)n_boards || sc->acc_slots < 2u * (uint32_t)n_boards || (sc->acc_slots & (sc->acc_slots - 1u)))
        return pulse::fail(PULSE_EINVAL, "PulseQTableScratch: need n >= n_boards and acc_slots a power of two >= 2 n_boards");
    const int n_wg = (n_boards + kBlock - 1) / kBlock;
    if (n_wg > 4096) return pulse::fail(PULSE_EINVAL, "PulseQTable: a shared table takes at most 1,048,576 boards per launch");
    if (sc->n % kBlock) return pulse::fail(PULSE_EINVAL, "PulseQTableScratch: n must be a multiple of 256 (the list is kept in segments of 256)");
    if (!g_defer_gave_up) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
            return pulse::fail_hip((int)hipGetLastError(), "PulseQTable: pinned word of the follow-up launch");
        g_defer_gave_up = static_cast<long long*>(p); *g_defer_gave_up = 0;
    }
    const long long gone = __atomic_load_n(g_defer_gave_up, __ATOMIC_ACQUIRE);
    if (gone != g_defer_gave_up_seen) {
        g_defer_gave_up_seen = gone;
        (void)hipMemsetAsync(sc->acc_key, 0, (size_t)sc->acc_slots * sizeof(uint64_t), st);
        (void)hipMemsetAsync(sc->acc_cnt, 0, (size_t)sc->acc_slots * sizeof(uint32_t), st);
        (void)hipMemsetAsync(sc->acc_sum, 0, (size_t)sc->acc_slots * sizeof(double), st);
        return pulse::fail(PULSE_EINTERNAL, "PulseQTable: the follow-up launch of an earlier update could not gather its workgroups within its wait "
                                            "(another process or stream on the GPU?); that launch's combined updates were dropped, the scratch's "
                                            "accumulators have been cleared");
    }
    *d = Deferred{sc->count, reinterpret_cast<unsigned long long*>(sc->cells), sc->targets, sc->owner,
                  reinterpret_cast<unsigned long long*>(sc->acc_key), sc->acc_cnt, sc->acc_sum, sc->acc_slots, (int)(launch_index & 1u), n_wg,
                  sc->wait_ticks > 0 ? (long long)sc->wait_ticks : 300000000ll, sc->debug_meet_extra > 0 ? (unsigned int)sc->debug_meet_extra : 0u,
                  g_defer_gave_up, sc->reserved0};
    return 0;
}
void launch_deferred(Entry* table, const Deferred& d, double alpha, hipStream_t st) {
    if (!d.count) return;
    // few, fat workgroups: with nothing deferred (boards spread over distinct states: the usual case) each reads one word and leaves
    hipLaunchKernelGGL(qtable_defer_kernel, dim3(kDeferBlocks), dim3(kBlock), 0, st, table, d, alpha);
}

}  // namespace

extern "C" {

int pulse_qtable_select(const PulseQTable* q, const int32_t* boards, int32_t n_boards, int32_t n, double epsilon, uint64_t seed,
                        uint64_t board_id0, uint64_t step_counter, int64_t* actions, int64_t* slots, void* stream) {
    if (int rc = check_table(q, n_boards, n)) return rc;
    if (!boards || !actions || !slots || n_boards < 0) return pulse::fail(PULSE_EINVAL, "pulse_qtable_select: null argument");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(qtable_select_kernel, dim3((n_boards + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       static_cast<Entry*>(q->entries), q->capacity, q->region_slots, boards, n_boards, n * n,
                       epsilon, seed, board_id0, step_counter, actions, slots);
    return finish_launch("pulse_qtable_select");
}

int pulse_qtable_update(const PulseQTable* q, const PulseQTableScratch* scratch, uint64_t launch_index, const int64_t* slots,
                        const int64_t* actions, const int32_t* rewards, const int32_t* next_boards, const uint8_t* terminal,
                        int32_t n_boards, int32_t n, double alpha, double gamma, void* stream) {
    if (int rc = check_table(q, n_boards, n)) return rc;
    if (!slots || !actions || !rewards || !next_boards || !terminal || n_boards < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_qtable_update: null argument");
    if (n_boards == 0) return 0;
    Deferred d;
    if (int rc = deferred_of(q, scratch, n_boards, launch_index, &d, (hipStream_t)stream)) return rc;
    Entry* table = static_cast<Entry*>(q->entries);
    hipLaunchKernelGGL(qtable_update_kernel, dim3((n_boards + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       table, q->capacity, q->region_slots, slots, actions, rewards, next_boards, terminal, n_boards, n * n, alpha, gamma, d);
    launch_deferred(table, d, alpha, (hipStream_t)stream);
    return finish_launch("pulse_qtable_update");
}

int pulse_qtable_rollout_step(const PulseQTable* q, const PulseQTableScratch* scratch, uint64_t launch_index, int32_t* boards,
                              int64_t* total_score, int32_t n_boards, int32_t n, double epsilon, double alpha, double gamma,
                              uint64_t agent_seed, uint64_t agent_step, uint64_t env_seed, uint64_t env_step, uint64_t board_id0,
                              int64_t* actions, int32_t* rewards, uint8_t* dones, int64_t* slots_io, void* stream) {
    if (int rc = check_table(q, n_boards, n)) return rc;
    if (!boards || !total_score || !actions || !rewards || !dones || !slots_io || n_boards < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_qtable_rollout_step: null argument");
    if (env_step == 0) return pulse::fail(PULSE_EINVAL, "pulse_qtable_rollout_step: env_step must be >= 1 (0 is the reset draw)");
    if (n_boards == 0) return 0;
    Deferred d;
    if (int rc = deferred_of(q, scratch, n_boards, launch_index, &d, (hipStream_t)stream)) return rc;
    Entry* table = static_cast<Entry*>(q->entries);
    const dim3 grid((n_boards + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t* lut = nullptr;
    if (n == 4) if (int rc = pulse::tfe_row_lut(&lut)) return rc;
#define PULSE_QT_ARGS table, q->capacity, q->region_slots, boards, total_score, n_boards, epsilon, alpha, gamma, agent_seed, agent_step, env_seed, \
                      env_step, board_id0, actions, rewards, dones, slots_io, d, lut
    if (n == 3) hipLaunchKernelGGL(qtable_rollout_step_kernel<3>, grid, block, 0, st, PULSE_QT_ARGS);
    else hipLaunchKernelGGL(qtable_rollout_step_kernel<4>, grid, block, 0, st, PULSE_QT_ARGS);
#undef PULSE_QT_ARGS
    launch_deferred(table, d, alpha, st);
    return finish_launch("pulse_qtable_rollout_step");
}

}  // extern "C"
