// hand_eval_device.h -- closed-form value of the best 5-card poker hand among 5..7 distinct cards, on the device.
//
// The same evaluator the table generator is built on (csrc/handranks_gen.cpp: hand_value, which tests/test_handranks.py
// holds bit for bit to the 2+2 table): value = category << 12 | index in category (1 = weakest), categories 1..9, i.e.
// exactly what the table walk p = HR[p + card] ... returns for a set of distinct valid cards (PokerGPU.py:437-444).
// The reset kernel uses it to fill the evaluation cache without walking the 130 MB table: nine dependent gathers per
// seat (five of them into cold lines) become ~400 vector instructions and one gather into the table's hot first 150 KB.
// Cards are 1..52 = 4 * rank + suit + 1 (rank 0 = deuce .. 12 = ace).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pulse_dev {

// index (1..1277, weakest first) of a 13-bit rank mask with five bits set that is not a straight; 0 otherwise.
// Masks in ascending numeric order are in ascending strength (top card first).
struct FiveIndexTable {
    uint16_t v[8192];
    constexpr FiveIndexTable() : v{} {
        int next = 1;
        for (int m = 0; m < 8192; ++m) {
            int bits = 0;
            for (int b = 0; b < 13; ++b) bits += (m >> b) & 1;
            bool straight = (m & 0x100F) == 0x100F;                       // A-2-3-4-5
            for (int top = 12; top >= 4; --top) straight = straight || ((m >> (top - 4)) & 0x1F) == 0x1F;
            v[m] = (bits == 5 && !straight) ? (uint16_t)next++ : (uint16_t)0;
        }
    }
};
__device__ const FiveIndexTable kFiveIndex{};

// Rank masks by multiplicity (c0: ranks held at least once .. c3: four times), suit counts (4 bits per suit) and the
// rank mask of every suit (16 bits per suit): adding a card is a handful of bit operations.
struct HandAcc {
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, suit_n = 0;
    uint64_t suit_ranks = 0;
};
__device__ __forceinline__ void hand_add(HandAcc& a, int card) {
    const uint32_t x = (uint32_t)(card - 1), s = x & 3u, b = 1u << (x >> 2);
    a.c3 |= a.c2 & b; a.c2 |= a.c1 & b; a.c1 |= a.c0 & b; a.c0 |= b;
    a.suit_n += 1u << (4u * s);
    a.suit_ranks |= (uint64_t)b << (16u * s);
}

__device__ __forceinline__ uint32_t keep_top(uint32_t m, int n) {        // the n highest set bits (at most two to drop: 7 cards)
    m = __popc(m) > n ? m & (m - 1u) : m;
    m = __popc(m) > n ? m & (m - 1u) : m;
    return m;
}
__device__ __forceinline__ int top_bit(uint32_t m) { return 31 - __clz((int)m); }                 // m != 0
__device__ __forceinline__ int low_bit(uint32_t m) { return __ffs((int)m) - 1; }                  // m != 0
__device__ __forceinline__ int straight_top(uint32_t m) {               // top rank (3 = wheel .. 12) of the best straight in m, -1
    const uint32_t x = m & (m >> 1) & (m >> 2) & (m >> 3) & (m >> 4);
    if (x) return top_bit(x) + 4;
    return (m & 0x100Fu) == 0x100Fu ? 3 : -1;
}
__device__ __forceinline__ int below(int r, int a) { return r - (a < r ? 1 : 0); }               // index of r among ranks != a
__device__ __forceinline__ int choose2(int n) { return n * (n - 1) / 2; }
__device__ __forceinline__ int choose3(int n) { return (int)(((uint32_t)(n * (n - 1) * (n - 2)) * 43691u) >> 18); }   // /6, exact below 2^17

// `five`: the index table to read -- kFiveIndex.v in global memory, or a copy a workgroup staged in LDS
__device__ __forceinline__ int hand_value(const HandAcc& a, const uint16_t* five_index = kFiveIndex.v) {
    const uint32_t fl = (a.suit_n + 0x3333u) & 0x8888u;                 // a suit held five times or more (at most one)
    const uint32_t all = a.c0;
    uint32_t suited = 0;
    if (fl) suited = (uint32_t)(a.suit_ranks >> (4u * (uint32_t)(low_bit(fl) & ~3))) & 0x1FFFu;
    const int five = five_index[keep_top(fl ? suited : all, 5)];        // one unconditional load serves flush and high card
    if (fl) {
        const int sh = straight_top(suited);
        return sh >= 0 ? (9 << 12) | (sh - 2) : (6 << 12) | five;
    }
    if (a.c3) {
        const int q = top_bit(a.c3), kick = top_bit(all & ~(1u << q));
        return (8 << 12) | (q * 12 + below(kick, q) + 1);
    }
    const uint32_t pairs = a.c1 & ~a.c2;
    int trip = -1;
    if (a.c2) {
        trip = top_bit(a.c2);
        const uint32_t second = (a.c2 & ~(1u << trip)) | pairs;         // a second set of three plays as the pair
        if (second) return (7 << 12) | (trip * 12 + below(top_bit(second), trip) + 1);
    }
    const int sh = straight_top(all);
    if (sh >= 0) return (5 << 12) | (sh - 2);
    if (trip >= 0) {
        const uint32_t ks = keep_top(all & ~(1u << trip), 2);
        return (4 << 12) | (trip * 66 + choose2(below(top_bit(ks), trip)) + below(low_bit(ks), trip) + 1);
    }
    const int n_pairs = __popc(pairs);
    if (n_pairs >= 2) {
        const uint32_t pm = keep_top(pairs, 2);
        const int hi = top_bit(pm), lo = low_bit(pm), kick = top_bit(all & ~pm);
        return (3 << 12) | ((choose2(hi) + lo) * 11 + (kick - (hi < kick ? 1 : 0) - (lo < kick ? 1 : 0)) + 1);
    }
    if (n_pairs == 1) {
        const int p = low_bit(pairs);
        uint32_t ks = keep_top(all & ~pairs, 3);
        const int k3 = top_bit(ks);
        ks &= ~(1u << k3);
        return (2 << 12) | (p * 220 + choose3(below(k3, p)) + choose2(below(top_bit(ks), p)) + below(low_bit(ks), p) + 1);
    }
    return (1 << 12) | five;
}

}  // namespace pulse_dev
