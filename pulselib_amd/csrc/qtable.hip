// qtable.hip -- tabular Q-learning for the batched 2048 roll-out (BASELINE.json config 3).
//
// The reference keeps `defaultdict(state tuple -> float64[n])` in Python and calls two numba scalars per
// step and per (single) board: utils/numba.py:5-21 (epsilon-greedy argmax) and :25-39 (Q update), driven
// by agents/TemperalDifference/QLearningNumba.py:10-37.  Here B boards act and learn per launch:
//   * the dict is an open-addressing hash table in HBM (uint64 key = the board packed as 4-bit log2
//     tiles, double[4] values, linear probing, lock-free insert by atomicCAS);
//   * pulse_qtable_select: look up / insert the state, epsilon-greedy over its four doubles
//     (first maximal index, as numba.py:13-19), Philox draws keyed by (seed, board id, step);
//   * pulse_qtable_update: q[s][a] += alpha * (target - q[s][a]), target = r or r + gamma * max q[s'].
// `region_slots` > 0 gives every board a private region of the table (B independent learners = B copies of
// the reference, bit-exact and race-free: the parity mode); 0 shares one table between all boards, updates
// then race benignly (Hogwild) and are applied with a 64-bit CAS loop so no update is torn.
#include <hip/hip_runtime.h>

#include "pulse_internal.h"

namespace {

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

// board -> key: 4 bits of log2(tile) per cell (0 = empty), row-major, cell 0 in the low nibble.
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
__device__ __forceinline__ long long find_or_insert(unsigned long long* keys, uint64_t base, uint64_t slots, uint64_t key) {
    uint64_t h = mix64(key) & (slots - 1);
    const uint64_t limit = slots < kMaxProbe ? slots : kMaxProbe;
    for (uint64_t probe = 0; probe < limit; ++probe) {
        const uint64_t s = base + ((h + probe) & (slots - 1));
        unsigned long long cur = keys[s];
        if (cur == key) return (long long)s;
        if (cur == 0ull) {
            cur = atomicCAS(&keys[s], 0ull, (unsigned long long)key);
            if (cur == 0ull || cur == key) return (long long)s;
        }
    }
    return -1;
}

__global__ __launch_bounds__(kBlock) void qtable_select_kernel(unsigned long long* keys, const double* __restrict__ values,
                                                              uint64_t capacity, uint64_t region_slots,
                                                              const int32_t* __restrict__ boards, int n_boards, int cells,
                                                              double epsilon, uint64_t seed, uint64_t board_id0, uint64_t step_counter,
                                                              int64_t* __restrict__ actions, int64_t* __restrict__ slots_out) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) return;
    const uint64_t key = pack_board(boards + (size_t)g * cells, cells);
    const uint64_t base = region_slots ? (uint64_t)g * region_slots : 0, slots = region_slots ? region_slots : capacity;
    const long long s = find_or_insert(keys, base, slots, key);
    slots_out[g] = s;
    const U4 r = philox4x32(seed, board_id0 + (uint64_t)g, step_counter);
    const double p = (double)(((uint64_t)r.x << 21) ^ (uint64_t)(r.y >> 11)) * (1.0 / 9007199254740992.0);   // 53-bit uniform
    int a;
    if (p < epsilon || s < 0) {
        a = (int)__umulhi(r.z, 4u);                                      // numba.py:9-11 random.randint(0, n-1)
    } else {
        const double* q = values + (size_t)s * 4;
        a = 0; double mx = q[0];
        for (int i = 1; i < 4; ++i) if (q[i] > mx) { mx = q[i]; a = i; }   // numba.py:13-19
    }
    actions[g] = a;
}

__global__ __launch_bounds__(kBlock) void qtable_update_kernel(unsigned long long* keys, double* values, uint64_t capacity,
                                                              uint64_t region_slots, const int64_t* __restrict__ slots_s,
                                                              const int64_t* __restrict__ actions, const int32_t* __restrict__ rewards,
                                                              const int32_t* __restrict__ next_boards, const uint8_t* __restrict__ terminal,
                                                              int n_boards, int cells, double alpha, double gamma) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_boards) return;
    const long long s = slots_s[g];
    if (s < 0) return;
    const uint64_t base = region_slots ? (uint64_t)g * region_slots : 0, slots = region_slots ? region_slots : capacity;
    // QLearningNumba.py:28-37 touches q[next_state] even for terminal transitions (defaultdict insert)
    const long long sn = find_or_insert(keys, base, slots, pack_board(next_boards + (size_t)g * cells, cells));
    double mx = 0.0;
    if (sn >= 0) {
        const double* qn = values + (size_t)sn * 4;
        mx = qn[0];
        for (int i = 1; i < 4; ++i) if (qn[i] > mx) mx = qn[i];           // numba.py:28-31
    }
    const double reward = (double)rewards[g];
    const double target = terminal[g] ? reward : __dadd_rn(reward, __dmul_rn(gamma, mx));       // numba.py:33-36
    double* cell = values + (size_t)s * 4 + (actions[g] & 3);
    if (region_slots) {
        const double old = *cell;
        *cell = __dadd_rn(old, __dmul_rn(alpha, __dsub_rn(target, old)));                       // numba.py:38-39
    } else {
        unsigned long long* raw = reinterpret_cast<unsigned long long*>(cell);
        unsigned long long seen = *raw, assumed;
        do {
            assumed = seen;
            const double old = __longlong_as_double((long long)assumed);
            const double upd = __dadd_rn(old, __dmul_rn(alpha, __dsub_rn(target, old)));
            seen = atomicCAS(raw, assumed, (unsigned long long)__double_as_longlong(upd));
        } while (seen != assumed);
    }
}

int finish_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pulse::fail_hip((int)e, what);
    return 0;
}

int check_table(const PulseQTable* q, int32_t n_boards, int32_t n) {
    if (!q || !q->keys || !q->values) return pulse::fail(PULSE_EINVAL, "PulseQTable: null table");
    if (n < 3 || n > 4) return pulse::fail(PULSE_EINVAL, "PulseQTable: board side must be 3 or 4 (64-bit state key)");
    const uint64_t slots = q->region_slots ? q->region_slots : q->capacity;
    if (slots == 0 || (slots & (slots - 1))) return pulse::fail(PULSE_EINVAL, "PulseQTable: capacity / region_slots must be a power of two");
    if (q->region_slots && (uint64_t)n_boards * q->region_slots > q->capacity)
        return pulse::fail(PULSE_EINVAL, "PulseQTable: capacity < n_boards * region_slots");
    return 0;
}

}  // namespace

extern "C" {

int pulse_qtable_select(const PulseQTable* q, const int32_t* boards, int32_t n_boards, int32_t n, double epsilon, uint64_t seed,
                        uint64_t board_id0, uint64_t step_counter, int64_t* actions, int64_t* slots, void* stream) {
    if (int rc = check_table(q, n_boards, n)) return rc;
    if (!boards || !actions || !slots || n_boards < 0) return pulse::fail(PULSE_EINVAL, "pulse_qtable_select: null argument");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(qtable_select_kernel, dim3((n_boards + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(q->keys), q->values, q->capacity, q->region_slots, boards, n_boards, n * n,
                       epsilon, seed, board_id0, step_counter, actions, slots);
    return finish_launch("pulse_qtable_select");
}

int pulse_qtable_update(const PulseQTable* q, const int64_t* slots, const int64_t* actions, const int32_t* rewards,
                        const int32_t* next_boards, const uint8_t* terminal, int32_t n_boards, int32_t n, double alpha, double gamma,
                        void* stream) {
    if (int rc = check_table(q, n_boards, n)) return rc;
    if (!slots || !actions || !rewards || !next_boards || !terminal || n_boards < 0)
        return pulse::fail(PULSE_EINVAL, "pulse_qtable_update: null argument");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(qtable_update_kernel, dim3((n_boards + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(q->keys), q->values, q->capacity, q->region_slots, slots, actions, rewards,
                       next_boards, terminal, n_boards, n * n, alpha, gamma);
    return finish_launch("pulse_qtable_update");
}

}  // extern "C"
