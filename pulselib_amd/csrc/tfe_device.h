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
