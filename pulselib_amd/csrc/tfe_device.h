// tfe_device.h -- the 2048 board move as device functions (environments/2048/TFE.py:17-108,152-189), shared by the
// step kernel (envs.hip) and the fused Q-learning roll-out step (qtable.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pulse_tfe {

// Board cell (i,j) of the k-times-rotated board in terms of the original: one rotation is
// out[n-1-j][i] = in[i][j]  (TFE.py:38-44)  <=>  out[r][c] = in[c][n-1-r].
template <int NB>
__device__ __forceinline__ int rot_src(int k, int r, int c) {
    // index into the un-rotated board of element (r,c) of the board rotated k times
    for (int q = 0; q < k; ++q) { const int nr = c, nc = NB - 1 - r; r = nr; c = nc; }
    return r * NB + c;
}

// ... and the inverse: the cell (r*NB + c) of the k-times-rotated board whose source is cell i of the original
template <int NB>
__device__ __forceinline__ int rot_dst(int k, int i) {
    int found = 0;
    for (int r = 0; r < NB; ++r)
        for (int c = 0; c < NB; ++c)
            if (rot_src<NB>(k, r, c) == i) found = r * NB + c;
    return found;
}

template <int NB>
__device__ __forceinline__ void tfe_spawn(int (&b)[NB * NB], uint32_t r_cell, uint32_t r_val) {
    int ne = 0;
#pragma unroll
    for (int i = 0; i < NB * NB; ++i) ne += b[i] == 0;
    if (!ne) return;
    const int k = (int)__umulhi(r_cell, (uint32_t)ne);
    const int val = (float)(r_val >> 8) * (1.0f / 16777216.0f) > 0.9f ? 4 : 2;   // TFE.py:30-33
    int seen = 0;
#pragma unroll
    for (int i = 0; i < NB * NB; ++i) {
        const bool empty = b[i] == 0;
        if (empty && seen == k) b[i] = val;
        seen += empty;
    }
}


// One move: direction k (0..3), returns the merge score (TFE.py:154-178); the board is updated in place.
template <int NB>
__device__ __forceinline__ int tfe_move(int (&b)[NB * NB], int k) {
    int out[NB * NB];
    int score = 0;
        // The four moves are one squash-left on the board rotated k times (TFE.py:158-178).  The boards of a wavefront
        // move in different directions, so four direction-specific copies of the squash would all be executed by every
        // lane; instead the rotation itself is data: every cell of the rotated board is a two-level select among its
        // four possible sources (3 v_cndmask per cell), ONE squash runs, and the inverse rotation is the same selects.
        const bool k_odd = (k & 1) != 0, k_hi = (k & 2) != 0;
        int rb[NB * NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const int lo = k_odd ? b[rot_src<NB>(1, r, c)] : b[rot_src<NB>(0, r, c)];
                const int hi = k_odd ? b[rot_src<NB>(3, r, c)] : b[rot_src<NB>(2, r, c)];
                rb[r * NB + c] = k_hi ? hi : lo;
            }
        int sq[NB * NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            int res[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) res[c] = 0;
            int w = 0; bool last_merged = false;
#pragma unroll
            for (int c = 0; c < NB; ++c) {                                              // TFE.py:85-101
                const int val = rb[r * NB + c];
                if (val != 0) {
                    int cur = 0;
#pragma unroll
                    for (int x = 0; x < NB; ++x) cur = x == w ? res[x] : cur;
                    if (cur == 0) {
#pragma unroll
                        for (int x = 0; x < NB; ++x) if (x == w) res[x] = val;
                    } else if (cur == val && !last_merged) {
#pragma unroll
                        for (int x = 0; x < NB; ++x) if (x == w) res[x] = val * 2;
                        score += val * 2; last_merged = true;
                    } else {
                        w += 1;
#pragma unroll
                        for (int x = 0; x < NB; ++x) if (x == w) res[x] = val;
                        last_merged = false;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) sq[r * NB + c] = res[c];
        }
        // back: out[rot_src(k, r, c)] = sq[r][c], i.e. out[i] = sq[cell of the rotated board that came from i]
#pragma unroll
        for (int i = 0; i < NB * NB; ++i) {
            const int lo = k_odd ? sq[rot_dst<NB>(1, i)] : sq[rot_dst<NB>(0, i)];
            const int hi = k_odd ? sq[rot_dst<NB>(3, i)] : sq[rot_dst<NB>(2, i)];
            out[i] = k_hi ? hi : lo;
        }
#pragma unroll
        for (int i = 0; i < NB * NB; ++i) b[i] = out[i];
    return score;
}

// ------------------------------------------------------------------------------------------------------------------
// The 4 x 4 board as 64 bits: 4 bits of log2(tile) per cell (0 = empty), row-major, cell 0 in the low nibble of `lo`
// (rows 0, 1 in `lo`, rows 2, 3 in `hi`) -- the Q-table's state key (qtable.hip) is this very word.  A row is 16 bits, and
// "squash a row to the left" (TFE.py:85-101) is ONE lookup in a table of 65,536 entries {row after the move, merge score / 2}
// (256 KB, built once per device by tfe_row_lut_kernel from the scalar rule below, resident in L2): the four moves are that
// lookup on the rows of the board, of the transposed board (bit transpose), or of either with the rows' cells reversed.
// The int32 cell form above stays the definition (and the path of boards this form cannot hold: a tile that is not a power
// of two, 1, negative or above 16,384 -- a wavefront that meets one takes the cell form for that lane); the move on the packed
// board costs ~60 vector instructions against ~900 (per-cell selects), and a step of the environment ~350 against 1,430.
struct PackedBoard { uint32_t lo, hi; };

// one row by the reference's rule on tile values; returns the merge score (a device-side table generator and test aid)
__device__ __forceinline__ uint32_t tfe_row_lut_entry(uint32_t row) {
    int v[4], res[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int n = (int)((row >> (4 * c)) & 15u); v[c] = n ? 1 << n : 0; }
    int w = 0, score = 0; bool last_merged = false;
    for (int c = 0; c < 4; ++c) {                                                           // TFE.py:85-101
        const int val = v[c];
        if (val == 0) continue;
        if (res[w] == 0) res[w] = val;
        else if (res[w] == val && !last_merged) { res[w] = val * 2; score += val * 2; last_merged = true; }
        else { w += 1; res[w] = val; last_merged = false; }
    }
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = res[c] > 0 ? 31 - __clz(res[c]) : 0;
        out |= (uint32_t)(n > 15 ? 15 : n) << (4 * c);         // (a row holding 2^15 twice cannot be written back: never looked up, see tfe_pack4)
    }
    return out | ((uint32_t)(score >> 1) & 0xFFFFu) << 16;
}

// cells -> packed; false: the board holds a value the packed form (or a merge of it) cannot: anything but 0 and 2^1 .. 2^14
__device__ __forceinline__ bool tfe_pack4(const int (&b)[16], PackedBoard& p) {
    uint32_t bad = 0, w[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t x = (uint32_t)b[i];
        bad |= (x & (x - 1u)) | (x & 0xFFFF8001u);
        const uint32_t n = 31u - (uint32_t)__clz((int)(x | 1u));
        w[i >> 3] |= n << (4 * (i & 7));
    }
    p.lo = w[0]; p.hi = w[1];
    return bad == 0u;
}
__device__ __forceinline__ void tfe_unpack4(const PackedBoard p, int (&b)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t n = ((i < 8 ? p.lo : p.hi) >> (4 * (i & 7))) & 15u;
        b[i] = (int)((1u << n) & ~1u);
    }
}

// nibble flags: bit 3 of every nibble of the result = "that nibble of x is non-zero" (shifts of at most 3 stay inside a nibble)
__device__ __forceinline__ uint32_t tfe_nz_nibbles(uint32_t x) {
    uint32_t t = x | (x << 1);
    t |= t << 2;
    return t & 0x88888888u;
}

// 4 x 4 nibble transpose of (lo, hi): delta swap inside the 2 x 2 blocks of nibbles, then the off-diagonal byte blocks change halves
__device__ __forceinline__ PackedBoard tfe_transpose(PackedBoard p) {
    uint32_t t = (p.lo ^ (p.lo >> 12)) & 0x0000F0F0u; p.lo ^= t ^ (t << 12);
    t = (p.hi ^ (p.hi >> 12)) & 0x0000F0F0u; p.hi ^= t ^ (t << 12);
    PackedBoard o;
    o.lo = __builtin_amdgcn_perm(p.hi, p.lo, 0x06020400u);       // bytes lo0, hi0, lo2, hi2
    o.hi = __builtin_amdgcn_perm(p.hi, p.lo, 0x07030501u);       // bytes lo1, hi1, lo3, hi3
    return o;
}
// the four cells of every row in reverse order
__device__ __forceinline__ uint32_t tfe_reverse_rows(uint32_t x) {
    x = __builtin_amdgcn_perm(x, x, 0x02030001u);                // bytes swapped inside each 16-bit row
    return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);  // nibbles swapped inside each byte
}

// One move of direction k on the packed board (TFE.py:154-178: squash-left of the board rotated k times, rotated back):
// rotation 1 reads the columns top to bottom (= the rows of the transpose), 2 the rows right to left, 3 the columns bottom to
// top; `lut` = the device's row table.  Returns the merge score.
__device__ __forceinline__ int tfe_move_packed(PackedBoard& p, int k, const uint32_t* __restrict__ lut) {
    const bool odd = (k & 1) != 0, rev = (k & 2) != 0;
    const PackedBoard t = tfe_transpose(p);
    uint32_t lo = odd ? t.lo : p.lo, hi = odd ? t.hi : p.hi;
    const uint32_t rl = tfe_reverse_rows(lo), rh = tfe_reverse_rows(hi);
    lo = rev ? rl : lo; hi = rev ? rh : hi;
    const uint32_t e0 = lut[lo & 0xFFFFu], e1 = lut[lo >> 16], e2 = lut[hi & 0xFFFFu], e3 = lut[hi >> 16];
    lo = __builtin_amdgcn_perm(e1, e0, 0x05040100u);             // low halves of e0, e1
    hi = __builtin_amdgcn_perm(e3, e2, 0x05040100u);
    const uint32_t rl2 = tfe_reverse_rows(lo), rh2 = tfe_reverse_rows(hi);
    lo = rev ? rl2 : lo; hi = rev ? rh2 : hi;
    PackedBoard o{lo, hi};
    const PackedBoard t2 = tfe_transpose(o);
    p.lo = odd ? t2.lo : o.lo; p.hi = odd ? t2.hi : o.hi;
    return (int)(((e0 >> 16) + (e1 >> 16) + (e2 >> 16) + (e3 >> 16)) << 1);
}

// TFE.py:17-34 on the packed board: the (r_cell * #empty >> 32)-th empty cell in row-major order gets a 2 (a 4 with
// probability 0.1).  Returns the number of empty cells BEFORE the spawn (0: nothing was spawned).
__device__ __forceinline__ int tfe_spawn_packed(PackedBoard& p, uint32_t r_cell, uint32_t r_val) {
    const uint32_t el = tfe_nz_nibbles(p.lo) ^ 0x88888888u, eh = tfe_nz_nibbles(p.hi) ^ 0x88888888u;      // bit 3 of every EMPTY nibble
    const int nl = __popc(el), ne = nl + __popc(eh);
    const int kth = (int)__umulhi(r_cell, (uint32_t)ne);
    const bool in_hi = kth >= nl;
    const uint32_t w = in_hi ? eh : el;
    const uint32_t want = (uint32_t)(kth - (in_hi ? nl : 0) + 1);
    // nibble i of (flags * 0x11111111) = number of empty cells among cells 0..i of this half (at most 8: no carry between nibbles);
    // the first nibble where it equals `want` is the cell
    const uint32_t prefix = (w >> 3) * 0x11111111u;
    const uint32_t hit = tfe_nz_nibbles(prefix ^ (want * 0x11111111u)) ^ 0x88888888u;
    const int at = __ffs((int)hit) - 4;                                                 // bit offset of the cell's nibble (hit: bit 3 of it)
    const uint32_t nib = (r_val >> 8) > 15099494u ? 2u : 1u;                           // (float)(r >> 8) * 2^-24 > 0.9f  <=>  r >> 8 > 0.9f * 2^24 (exact)
    const uint32_t v = ne > 0 ? nib << (at & 31) : 0u;
    p.lo |= in_hi ? 0u : v; p.hi |= in_hi ? v : 0u;
    return ne;
}

// TFE.py:48-67 after a spawn: `empty_before` = what tfe_spawn_packed returned (a board with two or more empty cells before it
// still has one)
__device__ __forceinline__ bool tfe_over_packed(const PackedBoard p, int empty_before) {
    const uint32_t hl = tfe_nz_nibbles(p.lo ^ (p.lo >> 4)) ^ 0x88888888u, hh = tfe_nz_nibbles(p.hi ^ (p.hi >> 4)) ^ 0x88888888u;
    const uint32_t horiz = (hl | hh) & 0x08880888u;                                    // cells 0..2 of a row equal their right neighbour
    const uint32_t mid = __builtin_amdgcn_alignbit(p.hi, p.lo, 16);                    // rows 1, 2
    const uint32_t v01_12 = tfe_nz_nibbles(p.lo ^ mid) ^ 0x88888888u;                  // rows 0 = 1, rows 1 = 2, cell by cell
    const uint32_t v23 = (tfe_nz_nibbles(p.hi ^ (p.hi >> 16)) ^ 0x88888888u) & 0x00008888u;
    return empty_before <= 1 && (horiz | v01_12 | v23) == 0u;
}

template <int NB>
__device__ __forceinline__ bool tfe_over(const int (&b)[NB * NB]) {
    bool over = true;                                                                   // TFE.py:48-67
#pragma unroll
    for (int i = 0; i < NB * NB; ++i) over = over && b[i] != 0;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int c = 0; c < NB - 1; ++c) over = over && b[r * NB + c] != b[r * NB + c + 1];
#pragma unroll
    for (int r = 0; r < NB - 1; ++r)
#pragma unroll
        for (int c = 0; c < NB; ++c) over = over && b[r * NB + c] != b[(r + 1) * NB + c];
    return over;
}

}  // namespace pulse_tfe
