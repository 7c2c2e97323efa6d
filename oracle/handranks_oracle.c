/*
 * oracle/handranks_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the public "Two-Plus-Two" 7-card evaluator table generator, the
 * algorithm behind `HandRanks.dat` (int32[32,487,834], 129,951,336 bytes) that the
 * reference loads at environments/Poker/PokerGPU.py:47-58 and walks at
 * PokerGPU.py:437-444, :477-478, :497-500, :518-521.
 *
 * Third-party dependency: the file itself comes from chenosaurus/poker-evaluator
 * (data/HandRanks.dat, URL at PokerGPU.py:49; un-vendored, no pinned version), which is the
 * output of the public 2+2 forum generator ("generate_table", Ray Wotton et al., using
 * Cactus Kev's 5-card equivalence-class numbering 1..7462).  The file is absent from
 * /root/reference and there is no network, so the published algorithm is restated here:
 *
 *   state ID   = up to 7 cards, one byte each (rank 1..13 in the high nibble, suit 1..4 in the
 *                low nibble, suit forced to 0 once it can no longer make a flush), bytes sorted
 *                descending, highest card in byte 0            (the generator's MakeID)
 *   state list = all IDs reachable from the empty hand by adding cards 1..52, kept sorted
 *                ascending, entry 0 = the empty hand             (the generator's SaveID)
 *   HR[s*53+53+c] = (next state index)*53+53 for <7 cards, else the hand value  (main loop)
 *   HR[s*53+53]   = hand value of a 5- or 6-card state
 *   hand value = (category << 12) | rank-in-category, category 1 (high card)..9 (straight
 *                flush), rank-in-category 1 = worst                 (the generator's DoEval)
 *
 * Deliberate differences from the published program, none of which can change the output:
 *   - SaveID's insertion into a sorted array is replaced by collect / sort / unique per card
 *     count (an ID with n+1 cards is numerically above every ID with n cards, so the
 *     published insertion never lands at or before the cursor and the final array is the same
 *     sorted set);
 *   - Cactus Kev's perfect-hash tables are replaced by a class table built here by
 *     enumerating all 7462 five-card classes and sorting them by poker strength (the
 *     numbering 1 = royal flush .. 7462 = 7-5-4-3-2 is a pure function of that order);
 *   - the 6- and 7-card values are the best of the C(n,5) five-card sub-hands, as published.
 *
 * Pinned by (tests/test_handranks.py): state count 612,977 (+53 leading zero entries = 612,978 x 53 ints); PokerGPU.py:13-18 constants
 * (4145, 36874, 4109, 74359, 823779); the reference tests' showdown payouts; an independent
 * combinatorial evaluator in oracle/poker_eval_ref.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp).  Exports hr_oracle_generate().
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define HR_STATES 612977                /* state IDs incl. the empty hand */
#define HR_SIZE   ((HR_STATES + 1) * 53)  /* 53 leading zeros + 53 slots per state = 32,487,834 */

/* ---------- 5-card class table (Cactus-Kev numbering, 1 = best .. 7462 = worst) ---------- */

typedef struct { uint64_t strength; uint64_t key; } class_t;

/* key: 3 bits of count per rank (39 bits) | flush bit 40 */
static uint64_t make_key(const int cnt[13], int flush) {
    uint64_t k = 0;
    for (int r = 0; r < 13; r++) k |= (uint64_t)cnt[r] << (3 * r);
    return k | ((uint64_t)(flush != 0) << 40);
}

/* Standard poker ordering packed as category(4 bits) then five 4-bit tiebreak ranks. */
static uint64_t strength_of(const int cnt[13], int flush) {
    int by_count[5][5]; int n_by[5] = {0,0,0,0,0};
    for (int r = 12; r >= 0; r--) if (cnt[r]) by_count[cnt[r]][n_by[cnt[r]]++] = r;
    int distinct = n_by[1] + n_by[2] + n_by[3] + n_by[4];
    int straight_high = -1;
    if (distinct == 5) {
        int hi = by_count[1][0], lo = by_count[1][4];
        if (hi - lo == 4) straight_high = hi;
        else if (hi == 12 && by_count[1][1] == 3) straight_high = 3; /* wheel A-5 */
    }
    int cat; int tb[5] = {0,0,0,0,0}; int nt = 0;
    if (straight_high >= 0 && flush) { cat = 9; tb[nt++] = straight_high; }
    else if (n_by[4]) { cat = 8; tb[nt++] = by_count[4][0]; tb[nt++] = by_count[1][0]; }
    else if (n_by[3] && n_by[2]) { cat = 7; tb[nt++] = by_count[3][0]; tb[nt++] = by_count[2][0]; }
    else if (flush) { cat = 6; for (int i = 0; i < 5; i++) tb[nt++] = by_count[1][i]; }
    else if (straight_high >= 0) { cat = 5; tb[nt++] = straight_high; }
    else if (n_by[3]) { cat = 4; tb[nt++] = by_count[3][0]; tb[nt++] = by_count[1][0]; tb[nt++] = by_count[1][1]; }
    else if (n_by[2] == 2) { cat = 3; tb[nt++] = by_count[2][0]; tb[nt++] = by_count[2][1]; tb[nt++] = by_count[1][0]; }
    else if (n_by[2] == 1) { cat = 2; tb[nt++] = by_count[2][0]; for (int i = 0; i < 3; i++) tb[nt++] = by_count[1][i]; }
    else { cat = 1; for (int i = 0; i < 5; i++) tb[nt++] = by_count[1][i]; }
    uint64_t s = (uint64_t)cat;
    for (int i = 0; i < 5; i++) s = (s << 4) | (uint64_t)tb[i];
    return s;
}

static int cmp_class_desc(const void* a, const void* b) {
    const class_t* x = (const class_t*)a; const class_t* y = (const class_t*)b;
    if (x->strength > y->strength) return -1;
    if (x->strength < y->strength) return 1;
    return 0;
}

#define CLS_HASH 32768
static uint64_t cls_keys[CLS_HASH];
static int16_t  cls_vals[CLS_HASH];
static int      cls_built = 0;

static unsigned cls_slot(uint64_t k) { return (unsigned)((k * 0x9E3779B97F4A7C15ULL) >> 49) & (CLS_HASH - 1); }

static void cls_put(uint64_t k, int v) {
    unsigned s = cls_slot(k);
    while (cls_vals[s]) s = (s + 1) & (CLS_HASH - 1);
    cls_keys[s] = k; cls_vals[s] = (int16_t)v;
}
static int cls_get(uint64_t k) {
    unsigned s = cls_slot(k);
    while (cls_vals[s]) { if (cls_keys[s] == k) return cls_vals[s]; s = (s + 1) & (CLS_HASH - 1); }
    return 0;
}

static int build_classes(void) {
    class_t* all = (class_t*)malloc(sizeof(class_t) * 8000);
    int n = 0, cnt[13];
    /* every multiset of 5 ranks with multiplicity <= 4, plus the flush twin of each all-distinct set */
    for (int a = 0; a < 13; a++) for (int b = a; b < 13; b++) for (int c = b; c < 13; c++)
    for (int d = c; d < 13; d++) for (int e = d; e < 13; e++) {
        memset(cnt, 0, sizeof cnt);
        cnt[a]++; cnt[b]++; cnt[c]++; cnt[d]++; cnt[e]++;
        int ok = 1, distinct = 0;
        for (int r = 0; r < 13; r++) { if (cnt[r] > 4) ok = 0; if (cnt[r]) distinct++; }
        if (!ok) continue;
        all[n].strength = strength_of(cnt, 0); all[n].key = make_key(cnt, 0); n++;
        if (distinct == 5) { all[n].strength = strength_of(cnt, 1); all[n].key = make_key(cnt, 1); n++; }
    }
    if (n != 7462) { free(all); return -1; }
    qsort(all, (size_t)n, sizeof(class_t), cmp_class_desc);
    memset(cls_vals, 0, sizeof cls_vals);
    for (int i = 0; i < n; i++) {
        if (i && all[i].strength == all[i - 1].strength) { free(all); return -2; }
        cls_put(all[i].key, i + 1);
    }
    free(all);
    cls_built = 1;
    return 0;
}

/* cards: rank 0..12, suit 1..4.  Returns the Cactus-Kev class number of the 5-card hand. */
static int eval5(const int rank[5], const int suit[5]) {
    int cnt[13]; memset(cnt, 0, sizeof cnt);
    for (int i = 0; i < 5; i++) cnt[rank[i]]++;
    int flush = (suit[0] == suit[1] && suit[1] == suit[2] && suit[2] == suit[3] && suit[3] == suit[4]);
    return cls_get(make_key(cnt, flush));
}

/* ---------- the generator's MakeID / DoEval ---------- */

/* Returns the new ID (0 = impossible hand) and the card count of the attempted hand. */
static uint64_t make_id(uint64_t id_in, int newcard, int* numcards_out) {
    int suitcount[5] = {0,0,0,0,0};
    int rankcount[14]; memset(rankcount, 0, sizeof rankcount);
    int wk[8]; memset(wk, 0, sizeof wk);
    for (int c = 0; c < 6; c++) wk[c + 1] = (int)((id_in >> (8 * c)) & 0xff);
    newcard--;
    wk[0] = (((newcard >> 2) + 1) << 4) + (newcard & 3) + 1;
    int numcards = 0, dup = 0;
    for (numcards = 0; wk[numcards]; numcards++) {
        suitcount[wk[numcards] & 0xf]++;
        rankcount[(wk[numcards] >> 4) & 0xf]++;
        if (numcards && wk[0] == wk[numcards]) dup = 1;
    }
    *numcards_out = numcards;
    if (dup) return 0;
    int needsuited = numcards - 2;
    if (numcards > 4)
        for (int r = 1; r < 14; r++) if (rankcount[r] > 4) return 0;
    if (needsuited > 1)
        for (int c = 0; c < numcards; c++)
            if (suitcount[wk[c] & 0xf] < needsuited) wk[c] &= 0xf0;
    /* sort descending (7 slots; empty ones are 0 and sink to the end) */
    for (int i = 1; i < 7; i++) {
        int v = wk[i], j = i - 1;
        while (j >= 0 && wk[j] < v) { wk[j + 1] = wk[j]; j--; }
        wk[j + 1] = v;
    }
    uint64_t id = 0;
    for (int c = 0; c < 7; c++) id |= (uint64_t)wk[c] << (8 * c);
    return id;
}

static int do_eval(uint64_t id) {
    if (!id) return 0;
    int hold[8]; int n = 0; int mainsuit = 20;
    for (int c = 0; c < 7; c++) {
        hold[c] = (int)((id >> (8 * c)) & 0xff);
        if (!hold[c]) break;
        n++;
        if (hold[c] & 0xf) mainsuit = hold[c] & 0xf;
    }
    int rank[7], suit[7]; int suititer = 1;
    for (int c = 0; c < n; c++) {
        rank[c] = (hold[c] >> 4) - 1;
        int s = hold[c] & 0xf;
        if (s == 0) {
            s = suititer++; if (suititer == 5) suititer = 1;
            if (s == mainsuit) { s = suititer++; if (suititer == 5) suititer = 1; }
        }
        suit[c] = s;
    }
    if (n < 5) return 0;
    int best = 9999;
    int idx[5];
    for (idx[0] = 0; idx[0] < n; idx[0]++) for (idx[1] = idx[0] + 1; idx[1] < n; idx[1]++)
    for (idx[2] = idx[1] + 1; idx[2] < n; idx[2]++) for (idx[3] = idx[2] + 1; idx[3] < n; idx[3]++)
    for (idx[4] = idx[3] + 1; idx[4] < n; idx[4]++) {
        int r5[5], s5[5];
        for (int k = 0; k < 5; k++) { r5[k] = rank[idx[k]]; s5[k] = suit[idx[k]]; }
        int v = eval5(r5, s5);
        if (v < best) best = v;
    }
    int hr = 7463 - best;
    if (hr < 1278) hr = hr - 0 + 4096 * 1;
    else if (hr < 4138) hr = hr - 1277 + 4096 * 2;
    else if (hr < 4996) hr = hr - 4137 + 4096 * 3;
    else if (hr < 5854) hr = hr - 4995 + 4096 * 4;
    else if (hr < 5864) hr = hr - 5853 + 4096 * 5;
    else if (hr < 7141) hr = hr - 5863 + 4096 * 6;
    else if (hr < 7297) hr = hr - 7140 + 4096 * 7;
    else if (hr < 7453) hr = hr - 7296 + 4096 * 8;
    else hr = hr - 7452 + 4096 * 9;
    return hr;
}

/* ---------- state list ---------- */

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) - (x < y);
}

static int find_id(const uint64_t* ids, int n, uint64_t id) {
    if (!id) return 0;
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (ids[mid] < id) lo = mid + 1; else if (ids[mid] > id) hi = mid - 1; else return mid;
    }
    return -1;
}

/* Fills out[HR_SIZE].  Returns the number of states (612977) or a negative error. */
int hr_oracle_generate(int32_t* out) {
    if (!cls_built && build_classes() != 0) return -1;
    uint64_t* ids = (uint64_t*)malloc(sizeof(uint64_t) * (HR_STATES + 16));
    if (!ids) return -2;
    int n_ids = 1; ids[0] = 0;
    int level_lo = 0, level_hi = 1;
    for (int ncards = 1; ncards <= 6; ncards++) {
        size_t cap = (size_t)(level_hi - level_lo) * 52;
        uint64_t* cand = (uint64_t*)malloc(sizeof(uint64_t) * (cap ? cap : 1));
        if (!cand) { free(ids); return -2; }
        size_t nc = 0; int dummy;
        for (int s = level_lo; s < level_hi; s++)
            for (int card = 1; card <= 52; card++) {
                uint64_t id = make_id(ids[s], card, &dummy);
                if (id) cand[nc++] = id;
            }
        qsort(cand, nc, sizeof(uint64_t), cmp_u64);
        size_t uniq = 0;
        for (size_t i = 0; i < nc; i++) if (!i || cand[i] != cand[i - 1]) cand[uniq++] = cand[i];
        if ((size_t)n_ids + uniq > HR_STATES) { free(cand); free(ids); return -3; }
        memcpy(ids + n_ids, cand, uniq * sizeof(uint64_t));
        free(cand);
        level_lo = n_ids; n_ids += (int)uniq; level_hi = n_ids; if (getenv("HR_DEBUG")) fprintf(stderr, "level %d: %zu ids (total %d)\n", ncards, uniq, n_ids);
    }
    if (n_ids != HR_STATES) { free(ids); return -4; }
    memset(out, 0, sizeof(int32_t) * (size_t)HR_SIZE);
    int bad = 0;
    #pragma omp parallel for schedule(dynamic, 1024) reduction(+:bad)
    for (int s = 0; s < n_ids; s++) {
        int numcards = 0;
        for (int card = 1; card <= 52; card++) {
            uint64_t id = make_id(ids[s], card, &numcards);
            int32_t v;
            if (numcards < 7) {
                int slot = find_id(ids, n_ids, id);
                if (slot < 0) { bad++; slot = 0; }
                v = slot * 53 + 53;
            } else {
                v = do_eval(id);
            }
            out[(size_t)s * 53 + 53 + card] = v;
        }
        if (numcards == 6 || numcards == 7) out[(size_t)s * 53 + 53] = do_eval(ids[s]);
    }
    free(ids);
    return bad ? -5 : n_ids;
}

#ifdef HR_ORACLE_MAIN
int main(int argc, char** argv) {
    int32_t* hr = (int32_t*)malloc(sizeof(int32_t) * (size_t)HR_SIZE);
    int rc = hr_oracle_generate(hr);
    fprintf(stderr, "states=%d\n", rc);
    if (rc > 0 && argc > 1) { FILE* f = fopen(argv[1], "wb"); fwrite(hr, 4, HR_SIZE, f); fclose(f); }
    free(hr);
    return rc > 0 ? 0 : 1;
}
#endif
