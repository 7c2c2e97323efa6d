// handranks_gen.cpp -- host-side builder of the "2+2" 7-card evaluator table.
//
// Replaces the HandRanks.dat download the reference requires
// (environments/Poker/PokerGPU.py:47-58: int32[32,487,834], walked as p = HR[p + card] from
// p = 53 at PokerGPU.py:437-444).  The table layout is fixed by that public format:
//   - a state is a canonical multiset of <= 6 cards: one byte per card (rank 1..13 << 4 | suit 1..4),
//     suit dropped to 0 as soon as it cannot reach a flush, bytes sorted descending and packed
//     little-endian (highest card in byte 0);
//   - states are numbered by ascending packed value, the empty hand is state 0;
//   - HR[53*(s+1) + c] = 53*(next+1) while the hand has < 7 cards, the hand value at 7 cards;
//     HR[53*(s+1)] = hand value for 5- and 6-card states; impossible hands lead to state 0 / value 0;
//   - hand value = category << 12 | index in category (1 = weakest), categories 1..9.
//
// This implementation is MI355X-host native code written for speed (a few hundred ms on 16 cores):
// breadth-first levels with sort+unique, a closed-form 5..7-card evaluator (no 21-subset loop, no
// hash tables) and a thread pool over states.  tests/test_handranks.py checks it bit for bit against
// the independent restatement in oracle/handranks_oracle.c and against PokerGPU.py:13-18's constants.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "pulse_internal.h"

namespace {

constexpr int kStates = 612977;  // including the empty hand

// ---- rank-set tables (13-bit masks) ---------------------------------------------------------
struct RankTables {
    int16_t straight_high[8192];  // highest straight contained in the mask, as top rank 3..12, -1 if none
    int16_t five_index[8192];     // for popcount-5 non-straight masks: 1..1277, weakest first
    RankTables() {
        for (int m = 0; m < 8192; ++m) {
            int hi = -1;
            for (int top = 12; top >= 4 && hi < 0; --top)
                if (((m >> (top - 4)) & 0x1F) == 0x1F) hi = top;
            if (hi < 0 && (m & 0x100F) == 0x100F) hi = 3;  // A-2-3-4-5
            straight_high[m] = (int16_t)hi;
            five_index[m] = 0;
        }
        // popcount-5 masks in ascending numeric order == ascending strength (compare top card first)
        int next = 1;
        for (int m = 0; m < 8192; ++m)
            if (__builtin_popcount(m) == 5 && straight_high[m] < 0) five_index[m] = (int16_t)next++;
    }
};
const RankTables kRT;

inline int top_bits(int mask, int n) {  // keep the n highest set bits
    while (__builtin_popcount(mask) > n) mask &= mask - 1;
    return mask;
}
inline int pos_without(int r, int a) { return r - (a < r); }                     // index of r among ranks != a
inline int pos_without2(int r, int a, int b) { return r - (a < r) - (b < r); }   // ... among ranks != a,b
inline int choose2(int n) { return n * (n - 1) / 2; }
inline int choose3(int n) { return n * (n - 1) * (n - 2) / 6; }

// Value of the best 5-card hand among n (5..7) cards given per-rank counts, the rank mask of
// the cards that kept their suit (all of one suit by construction) and how many of those there are.
int hand_value(const int cnt[13], int suited_mask, int suited_n) {
    if (suited_n >= 5) {
        int sh = kRT.straight_high[suited_mask];
        if (sh >= 0) return (9 << 12) | (sh - 2);
        return (6 << 12) | kRT.five_index[top_bits(suited_mask, 5)];
    }
    int all = 0, quad = -1, trip_hi = -1, trip_lo = -1, pair_mask = 0;
    for (int r = 0; r < 13; ++r) {
        if (!cnt[r]) continue;
        all |= 1 << r;
        if (cnt[r] == 4) quad = r;
        else if (cnt[r] == 3) { trip_lo = trip_hi; trip_hi = r; }
        else if (cnt[r] == 2) pair_mask |= 1 << r;
    }
    if (quad >= 0) {
        int kick = 31 - __builtin_clz((unsigned)(all & ~(1 << quad)));
        return (8 << 12) | (quad * 12 + pos_without(kick, quad) + 1);
    }
    if (trip_hi >= 0) {
        int p = -1;
        if (trip_lo >= 0) p = trip_lo;
        if (pair_mask) { int hp = 31 - __builtin_clz((unsigned)pair_mask); if (hp > p) p = hp; }
        if (p >= 0) return (7 << 12) | (trip_hi * 12 + pos_without(p, trip_hi) + 1);
    }
    int sh = kRT.straight_high[all];
    if (sh >= 0) return (5 << 12) | (sh - 2);
    if (trip_hi >= 0) {
        int ks = top_bits(all & ~(1 << trip_hi), 2);
        int k2 = 31 - __builtin_clz((unsigned)ks), k1 = __builtin_ctz((unsigned)ks);
        return (4 << 12) | (trip_hi * 66 + choose2(pos_without(k2, trip_hi)) + pos_without(k1, trip_hi) + 1);
    }
    int npairs = __builtin_popcount(pair_mask);
    if (npairs >= 2) {
        int pm = top_bits(pair_mask, 2);
        int hi = 31 - __builtin_clz((unsigned)pm), lo = __builtin_ctz((unsigned)pm);
        int kick = 31 - __builtin_clz((unsigned)(all & ~pm));
        return (3 << 12) | ((choose2(hi) + lo) * 11 + pos_without2(kick, hi, lo) + 1);
    }
    if (npairs == 1) {
        int p = __builtin_ctz((unsigned)pair_mask);
        int ks = top_bits(all & ~pair_mask, 3);
        int k3 = 31 - __builtin_clz((unsigned)ks); ks &= ~(1 << k3);
        int k2 = 31 - __builtin_clz((unsigned)ks), k1 = __builtin_ctz((unsigned)ks);
        return (2 << 12) | (p * 220 + choose3(pos_without(k3, p)) + choose2(pos_without(k2, p)) + pos_without(k1, p) + 1);
    }
    return (1 << 12) | kRT.five_index[top_bits(all, 5)];
}

// Value of a packed state with 5..7 cards (0 for the impossible-hand marker).
int state_value(uint64_t id) {
    if (!id) return 0;
    int cnt[13] = {0}, suited_mask = 0, suited_n = 0, n = 0;
    for (; n < 7; ++n) {
        int b = (int)((id >> (8 * n)) & 0xFF);
        if (!b) break;
        int r = (b >> 4) - 1;
        cnt[r]++;
        if (b & 0xF) { suited_mask |= 1 << r; suited_n++; }
    }
    return n >= 5 ? hand_value(cnt, suited_mask, suited_n) : 0;
}

// Add card (1..52, 4*rank+suit+1) to a packed state.  Returns 0 for an impossible hand.
// n_out = number of cards of the attempted hand.
uint64_t add_card(uint64_t id, int card, int& n_out) {
    uint8_t c[8];
    int n = 0;
    card -= 1;
    const uint8_t nc = (uint8_t)((((card >> 2) + 1) << 4) | ((card & 3) + 1));
    c[n++] = nc;
    bool dup = false;
    for (int i = 0; i < 6; ++i) {
        uint8_t b = (uint8_t)(id >> (8 * i));
        if (!b) break;
        dup |= (b == nc);
        c[n++] = b;
    }
    n_out = n;
    if (dup) return 0;
    int suit_n[5] = {0, 0, 0, 0, 0};
    uint8_t rank_n[14] = {0};
    for (int i = 0; i < n; ++i) { suit_n[c[i] & 0xF]++; rank_n[c[i] >> 4]++; }
    if (n > 4)
        for (int r = 1; r < 14; ++r) if (rank_n[r] > 4) return 0;
    const int need = n - 2;
    if (need > 1)
        for (int i = 0; i < n; ++i) if (suit_n[c[i] & 0xF] < need) c[i] &= 0xF0;
    std::sort(c, c + n, [](uint8_t a, uint8_t b) { return a > b; });
    uint64_t out = 0;
    for (int i = 0; i < n; ++i) out |= (uint64_t)c[i] << (8 * i);
    return out;
}

}  // namespace

extern "C" int pulse_handranks_generate(int32_t* out, int n_threads) {
    if (!out) return pulse::fail(PULSE_EINVAL, "pulse_handranks_generate: out is null");
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = std::min(n_threads, 64);

    std::vector<uint64_t> ids;
    ids.reserve(kStates);
    ids.push_back(0);
    size_t level_begin[8] = {0};  // level_begin[k] = first index of k-card states
    level_begin[0] = 0; level_begin[1] = 1;
    for (int k = 1; k <= 6; ++k) {
        const size_t lo = level_begin[k - 1], hi = ids.size();
        std::vector<std::vector<uint64_t>> parts((size_t)n_threads);
        std::vector<std::thread> pool;
        for (int w = 0; w < n_threads; ++w)
            pool.emplace_back([&, w] {
                auto& v = parts[(size_t)w];
                int n;
                for (size_t s = lo + (size_t)w; s < hi; s += (size_t)n_threads)
                    for (int card = 1; card <= 52; ++card) {
                        uint64_t id = add_card(ids[s], card, n);
                        if (id) v.push_back(id);
                    }
                std::sort(v.begin(), v.end());
                v.erase(std::unique(v.begin(), v.end()), v.end());
            });
        for (auto& t : pool) t.join();
        std::vector<uint64_t> next;
        for (auto& p : parts) next.insert(next.end(), p.begin(), p.end());
        std::sort(next.begin(), next.end());
        next.erase(std::unique(next.begin(), next.end()), next.end());
        level_begin[k] = ids.size();
        ids.insert(ids.end(), next.begin(), next.end());
    }
    level_begin[7] = ids.size();
    if ((int)ids.size() != kStates) return pulse::fail(PULSE_EINTERNAL, "pulse_handranks_generate: state count mismatch");

    std::memset(out, 0, sizeof(int32_t) * (size_t)PULSE_HANDRANKS_LEN);
    std::atomic<int> missing{0};
    std::atomic<size_t> cursor{0};
    const size_t total = ids.size();
    std::vector<std::thread> pool;
    for (int w = 0; w < n_threads; ++w)
        pool.emplace_back([&] {
            for (;;) {
                const size_t begin = cursor.fetch_add(2048);
                if (begin >= total) break;
                const size_t end = std::min(total, begin + 2048);
                for (size_t s = begin; s < end; ++s) {
                    int32_t* row = out + 53 * (s + 1);
                    int n = 0;
                    for (int card = 1; card <= 52; ++card) {
                        const uint64_t id = add_card(ids[s], card, n);
                        if (n == 7) { row[card] = state_value(id); continue; }
                        size_t slot = 0;
                        if (id) {
                            auto first = ids.begin() + (ptrdiff_t)level_begin[n], last = ids.begin() + (ptrdiff_t)level_begin[n + 1];
                            auto it = std::lower_bound(first, last, id);
                            if (it == last || *it != id) { missing++; } else slot = (size_t)(it - ids.begin());
                        }
                        row[card] = (int32_t)(53 * (slot + 1));
                    }
                    if (n == 6 || n == 7) row[0] = state_value(ids[s]);
                }
            }
        });
    for (auto& t : pool) t.join();
    if (missing.load()) return pulse::fail(PULSE_EINTERNAL, "pulse_handranks_generate: unresolved successor state");
    return 0;
}
