// qnet_train16.h -- the training launch on tiles of SIXTEEN rows, two workgroups per CU (included by qnet.hip after the
// 32-row kernels).  For batches with more tiles than CUs: the 32-row kernels keep a SIMD busy 46 % of a
// tile (fp32 MFMA cycles + vector instructions, which do not overlap on this chip), the rest are the barriers between a
// tile's phases -- and their 157 KB of LDS allow nobody else on the CU to fill them.  Here:
//
//   * v_mfma_f32_16x16x4_f32 (layouts: tools/probes/mfma16x16_probe.hip): a tile of 16 rows, activations and derivatives in
//     LDS as [unit][row] at a pitch of 17 -- 71 KB, so TWO workgroups share a CU and run in each other's barrier waits;
//   * four wavefronts per workgroup and every phase four wide: the output tiles of a layer (16 units x 16 rows) of BOTH
//     networks are dealt to the wavefronts (no k splits, no exchange space), the backward phases give every wavefront delta
//     tiles AND weight-gradient blocks of the same layer;
//   * the gradient slice is 138 blocks of 16x16 ([block][lane][4 registers]: one 16-byte store per lane and block), owned
//     by the same wavefront for the whole launch; slice16_param maps it for the reduce launch.
// Same rows, same dropout draws (per row and unit, Philox call = unit / 8), same sums up to their order as the 32-row kernels.
#pragma once

constexpr int kP16 = 17;
struct T16Lds {                          // offsets in floats
    static constexpr int Xs = 0, A1 = Xs + 64 * kP16, A2 = A1 + 128 * kP16, A3 = A2 + 128 * kP16, A4 = A3 + 64 * kP16,
                         G1 = A4 + 32 * kP16, G2 = G1 + 128 * kP16, G3 = G2 + 128 * kP16, G4 = G3 + 64 * kP16,
                         Da = G4 + 32 * kP16, Db = Da + 128 * kP16,
                         List = Db + 128 * kP16,            // 256 first positions + 8 wavefront totals
                         Tgt = List + 264,                  // 16 words: max_a' Q_target per column; 16 spare (reward sums at the end)
                         End = Tgt + 32;
};
constexpr size_t kTrain16LdsBytes = (size_t)T16Lds::End * sizeof(float);
// slice: layer 1 8 x 4 blocks (block = 4 mo + nt), layer 2 8 x 8, layer 3 4 x 8, layer 4 2 x 4, layer 5 1 x 2; then the five bias
// vectors and 4 statistics as in the 32-row layout.  It fits the same allocation (pulse_qnet_slice_floats()).
constexpr int kS16Blk1 = 0, kS16Blk2 = 32, kS16Blk3 = 96, kS16Blk4 = 128, kS16Blk5 = 136, kS16Blocks = 138;
constexpr int kS16Bias = kS16Blocks * 256, kS16Stats = kS16Bias + 384, kS16Pitch = kS16Stats + 4;
static_assert(kS16Pitch <= kSlicePitch, "the 16-row slice fits the 32-row slice's allocation");

// flat parameter index (order w1,b1,...,w5,b5) of element j of a 16-row slice, or -1 for a padding element
__device__ __forceinline__ int slice16_param(int j, int K1, int A) {
    const int n_out[5] = {128, 128, 64, 32, A}, n_in[5] = {K1, 128, 128, 64, 32};
    const int blk0[6] = {kS16Blk1, kS16Blk2, kS16Blk3, kS16Blk4, kS16Blk5, kS16Blocks}, nts[5] = {4, 8, 8, 4, 2};
    const int bias0[6] = {0, 128, 256, 320, 352, 384};
    int base[5], bbase[5], acc = 0;
#pragma unroll
    for (int l = 0; l < 5; ++l) { base[l] = acc; acc += n_out[l] * n_in[l]; bbase[l] = acc; acc += n_out[l]; }
    if (j >= kS16Bias) {
        const int u = j - kS16Bias;
#pragma unroll
        for (int l = 0; l < 5; ++l)
            if (u >= bias0[l] && u < bias0[l + 1]) return (u - bias0[l]) < n_out[l] ? bbase[l] + (u - bias0[l]) : -1;
        return -1;
    }
    const int blk = j >> 8, lane = (j >> 2) & 63, r = j & 3, n = lane & 15, g = lane >> 4;      // [block][lane][4]
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        if (blk >= blk0[l] && blk < blk0[l + 1]) {
            const int b = blk - blk0[l], mo = b / nts[l], nt = b - mo * nts[l];
            const int o = 16 * mo + 4 * g + r, in = 16 * nt + n;
            return (o < n_out[l] && in < n_in[l]) ? base[l] + o * n_in[l] + in : -1;
        }
    }
    return -1;
}

typedef float f32x4t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4t t16_zero() { f32x4t z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }
__device__ __forceinline__ f32x4t mfma16(float a, float b, f32x4t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// A operands of output tile `mo` of a forward layer: lane (m, kk) holds W[16 mo + m][16 mt + 4 kk .. + 3] (a float4 per input
// group; zero past the real units / inputs)
template <int MT>
__device__ __forceinline__ void t16_w_load(float (&wa)[MT][4], const float* __restrict__ w, int K, int n_units, int mo, int m, int kk) {
    const int u = 16 * mo + m;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int k = 16 * mt + 4 * kk;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (u < n_units && k < K) v = *reinterpret_cast<const float4*>(w + (size_t)u * K + k);
        wa[mt][0] = v.x; wa[mt][1] = v.y; wa[mt][2] = v.z; wa[mt][3] = v.w;
    }
}
// B operands: lane (n, kk) holds S[16 mt + 4 kk + r][n]
template <int MT>
__device__ __forceinline__ void t16_b_read(float (&bv)[MT][4], const float* __restrict__ S, int n, int kk) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[mt][r] = S[(16 * mt + 4 * kk + r) * kP16 + n];
    }
}
// two independent products side by side (a dependent 16x16x4 waits 52 cycles, two interleaved chains issue every 32)
template <int MT>
__device__ __forceinline__ void t16_mul2(const float (&wa0)[MT][4], const float (&bv0)[MT][4], const float (&wa1)[MT][4], const float (&bv1)[MT][4],
                                         f32x4t& acc0, f32x4t& acc1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0 = mfma16(wa0[mt][r], bv0[mt][r], acc0); acc1 = mfma16(wa1[mt][r], bv1[mt][r], acc1); }
    }
}
template <int MT>
__device__ __forceinline__ void t16_mul1(const float (&wa)[MT][4], const float (&bv)[MT][4], f32x4t& acc) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(wa[mt][r], bv[mt][r], acc);
    }
}

// dropout keep bits of lane (row, g)'s four units 16 mo + 4 g + r of a layer whose Philox calls start at `call0` (layer 2: 0,
// layer 3: 16): unit u drops when the 16-bit uniform (call u / 8, word (u % 8) / 2, half u % 2) is below thr -- the definition
// of dropout_keep_bits (qnet_device.h) and of the oracle
__device__ __forceinline__ uint32_t t16_keep_bits(uint64_t seed, uint64_t gid, uint64_t step, int call0, int mo, int g, uint32_t thr) {
    const U4 w = philox4x32(seed ^ 0xD50F0D50F0ull, gid, step * 32 + (uint64_t)(call0 + 2 * mo + (g >> 1)));
    const uint32_t lo = (g & 1) ? w.z : w.x, hi = (g & 1) ? w.w : w.y;
    return (uint32_t)((lo & 0xFFFFu) >= thr) | (uint32_t)((lo >> 16) >= thr) << 1 | (uint32_t)((hi & 0xFFFFu) >= thr) << 2 | (uint32_t)((hi >> 16) >= thr) << 3;
}

// epilogue of an output tile: z = acc + bias -> a = gelu(z) * m to As; TRAIN also g = gelu'(z) * m to Gs (m = dropout keep * scale)
template <bool TRAIN>
__device__ __forceinline__ void t16_epilogue(const f32x4t& acc, const float* __restrict__ bias, int mo, int n, int g, uint32_t keep, float scale,
                                             float* __restrict__ As, float* __restrict__ Gs) {
    const float4 b = *reinterpret_cast<const float4*>(bias + 16 * mo + 4 * g);
    const f32x2 z01 = {acc[0] + b.x, acc[1] + b.y}, z23 = {acc[2] + b.z, acc[3] + b.w};
    f32x2 y01, y23, d01, d23;
    gelu_pair2(z01, y01, d01); gelu_pair2(z23, y23, d23);
    const float y[4] = {y01.x, y01.y, y23.x, y23.y}, dy[4] = {d01.x, d01.y, d23.x, d23.y};
    const int u0 = 16 * mo + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mk = ((keep >> r) & 1u) ? scale : 0.0f;
        As[(u0 + r) * kP16 + n] = y[r] * mk;
        if (TRAIN) Gs[(u0 + r) * kP16 + n] = dy[r] * mk;
    }
}

// tile `mo` (16 units of layer l-1) of delta_{l-1} = (W_l^T . delta_l) * g_{l-1}; W_l is n_out x K, delta_l = rows [0, 16 MT) of D.
// Lane (m, kk) reads W_l[16 mt + 4 kk + r][16 mo + m] down the columns (16 consecutive floats per lane group).
template <int MT>
__device__ __forceinline__ void t16_back_load(float (&wa)[MT][4], const float* __restrict__ w, int n_out, int K, int mo, int m, int kk) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * mt + 4 * kk + r;
            wa[mt][r] = u < n_out ? w[(size_t)u * K + 16 * mo + m] : 0.0f;
        }
    }
}
__device__ __forceinline__ void t16_back_store(const f32x4t& acc, int mo, int n, int g, const float* __restrict__ G, float* __restrict__ Dn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int u = 16 * mo + 4 * g + r; Dn[u * kP16 + n] = acc[r] * G[u * kP16 + n]; }
}

// NB blocks (mo, nt0 .. nt0 + NB - 1) of dW_l += delta_l . a_{l-1}^T for this tile: A = delta (lane (m, kk): unit 16 mo + m, row
// 4 s + kk), B = a_{l-1} (lane (n, kk): unit 16 nt + n); four MFMAs per block, the blocks' chains interleaved.  Returns the
// lane's part of db (sum over its four rows of delta): the caller adds the other three lane groups'.
template <int NB>
__device__ __forceinline__ float t16_dw(const float* __restrict__ D, const float* __restrict__ Ap, float* __restrict__ slice, int blk0, int mo, int nt0,
                                        int m, int kk, int lane, bool first) {
    // (the slice's old values are asked for first: the round trip runs beside the LDS reads and the MFMAs)
    float4 old[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        old[b] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!first) old[b] = *(reinterpret_cast<const float4*>(slice + (size_t)(blk0 + b) * 256) + lane);
    }
    float ad[4], ap[NB][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ad[s] = D[(16 * mo + m) * kP16 + 4 * s + kk];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ap[b][s] = Ap[(16 * (nt0 + b) + m) * kP16 + 4 * s + kk];
    }
    __builtin_amdgcn_sched_barrier(0);                            // (the LDS reads ahead of the MFMAs)
    f32x4t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = t16_zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = mfma16(ad[s], ap[b][s], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float4* dst = reinterpret_cast<float4*>(slice + (size_t)(blk0 + b) * 256) + lane;
        *dst = make_float4(acc[b][0] + old[b].x, acc[b][1] + old[b].y, acc[b][2] + old[b].z, acc[b][3] + old[b].w);
    }
    return (ad[0] + ad[1]) + (ad[2] + ad[3]);
}
__device__ __forceinline__ float t16_rowsum(float bs) { bs += __shfl_xor(bs, 16); bs += __shfl_xor(bs, 32); return bs; }   // over the four lane groups

template <int MT1>
__global__ __launch_bounds__(256, 2) void qnet_train16_kernel(const TrainArgs a) {
    extern __shared__ float lds[];
    const FlatNet& net = a.net;
    const int wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int K1 = net.state_dim, A = net.n_actions;
    float* Xs = lds + T16Lds::Xs; float* A1 = lds + T16Lds::A1; float* A2 = lds + T16Lds::A2; float* A3 = lds + T16Lds::A3; float* A4 = lds + T16Lds::A4;
    float* G1 = lds + T16Lds::G1; float* G2 = lds + T16Lds::G2; float* G3 = lds + T16Lds::G3; float* G4 = lds + T16Lds::G4;
    float* Da = lds + T16Lds::Da; float* Db = lds + T16Lds::Db; float* Tg = lds + T16Lds::Tgt;
    float* Xn = Db; float* T1 = Da; float* T2 = Db; float* T3 = Da; float* T4 = Db;      // the target network's activations borrow the delta buffers
    const uint32_t thr = (uint32_t)(a.drop_p * 65536.0f);
    const float scale = 1.0f / (1.0f - a.drop_p);

    // this wavefront's rows of db, live across every tile of the launch (its blocks of dW accumulate in the slice): output tiles
    // 2 wq, 2 wq + 1 of layers 1 and 2, tile wq of layer 3; wavefront 0 both tiles of layer 4, wavefront 2 layer 5
    float b1a = 0.0f, b1b = 0.0f, b2a = 0.0f, b2b = 0.0f, b3 = 0.0f, b4a = 0.0f, b4b = 0.0f, b5 = 0.0f;
    float rows_sum = 0.0f, sq_sum = 0.0f;                        // wavefront 1
    bool used = false;
    float* part = a.partials + (size_t)blockIdx.x * kSlicePitch;

    if (blockIdx.x == 0 && threadIdx.x < 8) a.meet[threadIdx.x] = 0u;          // the fused reduce launch's arrival counters
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[0] = 0.0f;
    QSTAMP(0);
    float reward_sum = 0.0f;
    if (a.book) {                                                // what the select launch does per candidate row, when act made the lists
        for (int win = blockIdx.x; win * 256 < a.n_rows; win += gridDim.x) {
            const int row = win * 256 + threadIdx.x;
            const bool cand = row < a.n_rows && (a.row_mask == nullptr || a.row_mask[row] != 0);
            float rew = cand ? a.rewards[row] : 0.0f;            // episode reward: rows of row_mask, before the status filter (trainGPU.py:96)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) rew += __shfl_xor(rew, off);
            reward_sum += rew;
            if (a.terminated && row < a.n_rows && a.dones[row]) a.terminated[row] = 1;      // trainGPU.py:86
        }
    }
    const int W = 1 << a.win_shift;
    const int n_windows = (a.n_rows + W - 1) >> a.win_shift, per = (n_windows + 255) / 256;
    int* chunk = reinterpret_cast<int*>(lds + T16Lds::List);     // [256] first position of thread t's windows; [257..260] wavefront totals
    int T;
    {
        int mine = 0;
        for (int j = 0; j < per; ++j) { const int w = threadIdx.x * per + j; mine += w < n_windows ? a.sel_counts[w] : 0; }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); incl += lane0 >= off ? o : 0; }
        int* wtot = chunk + 257;
        if (lane0 == 63) wtot[wq] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) base += i < wq ? wtot[i] : 0;
        chunk[threadIdx.x] = base + incl - mine;
        T = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
        __syncthreads();
    }
    const int G = (int)gridDim.x;
    const int n_tiles = (T + 15) / 16;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.meet[kMeetUsed] = (unsigned)min(n_tiles, G);      // workgroups 0 .. n_tiles - 1 hold a gradient slice
    for (int ti = blockIdx.x; ti < n_tiles; ti += G) {
        const bool first = !used;
        used = true;
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        float* part_t = part + opaque;
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int n = lane & 15, g = lane >> 4;                  // (as A-operand coordinates: m = n, kk = g)
        // column n = position lo + n of the batch's selected rows (the list arithmetic of the 32-row kernels)
        const int lo = (int)((long long)ti * T / n_tiles), hi = (int)((long long)(ti + 1) * T / n_tiles);
        int rowc = -1;
        if (lo + n < hi) {
            const int p = lo + n;
            int t = 0;
#pragma unroll
            for (int s = 128; s >= 1; s >>= 1) t += (chunk[t + s] <= p) ? s : 0;
            int w = t * per, acc = chunk[t], cnt = a.sel_counts[w];
            while (p >= acc + cnt) { acc += cnt; ++w; cnt = a.sel_counts[w]; }
            rowc = a.sel_rows[((size_t)w << a.win_shift) + (p - acc)];
        }
        const bool live = rowc >= 0;
        const int rw = max(rowc, 0);
        const uint64_t gid = a.table_id0 + (uint64_t)rw;
        lds_barrier();                                            // the previous tile's readers are done
        QSTAMP(1);
        {   // the 16 rows of s and s' -> Xs / Xn as [input][row], zero above state_dim: thread (row tid & 15, inputs 4 (tid >> 4) ..)
            const int kq = (int)(threadIdx.x >> 4);              // (a thread's column is its lane's: tid & 15 == lane & 15)
            float4 xs = make_float4(0.0f, 0.0f, 0.0f, 0.0f), xn = xs;
            if (live && 4 * kq < K1) {
                xs = *reinterpret_cast<const float4*>(a.states + (size_t)rw * a.stride + 4 * kq);
                xn = *reinterpret_cast<const float4*>(a.next_states + (size_t)rw * a.next_stride + 4 * kq);
            }
            Xs[(4 * kq + 0) * kP16 + n] = xs.x; Xs[(4 * kq + 1) * kP16 + n] = xs.y; Xs[(4 * kq + 2) * kP16 + n] = xs.z; Xs[(4 * kq + 3) * kP16 + n] = xs.w;
            Xn[(4 * kq + 0) * kP16 + n] = xn.x; Xn[(4 * kq + 1) * kP16 + n] = xn.y; Xn[(4 * kq + 2) * kP16 + n] = xn.z; Xn[(4 * kq + 3) * kP16 + n] = xn.w;
        }
        const float row_done = (live && a.dones[rw]) ? 1.0f : 0.0f, row_reward = live ? a.rewards[rw] : 0.0f;
        const int act = live ? (int)a.actions[rw] : -1;
        lds_barrier();
        // ---- the two forwards, layer by layer: the output tiles of both networks dealt to the four wavefronts
        {   // layer 1: tiles 2 wq, 2 wq + 1 of both networks (all four tiles' weights are asked for before anything waits for them)
            float wt0[MT1][4], wt1[MT1][4], wo0[MT1][4], wo1[MT1][4], bv[MT1][4];
            t16_w_load<MT1>(wt0, net_w(a.tgt, 0), K1, 128, 2 * wq, n, g); t16_w_load<MT1>(wt1, net_w(a.tgt, 0), K1, 128, 2 * wq + 1, n, g);
            t16_w_load<MT1>(wo0, net_w(net, 0), K1, 128, 2 * wq, n, g); t16_w_load<MT1>(wo1, net_w(net, 0), K1, 128, 2 * wq + 1, n, g);
            t16_b_read<MT1>(bv, Xn, n, g);
            f32x4t c0 = t16_zero(), c1 = t16_zero(), c2 = t16_zero(), c3 = t16_zero();
            t16_mul2<MT1>(wt0, bv, wt1, bv, c0, c1);
            t16_b_read<MT1>(bv, Xs, n, g);
            t16_mul2<MT1>(wo0, bv, wo1, bv, c2, c3);
            t16_epilogue<false>(c0, net_b(a.tgt, 0), 2 * wq, n, g, 0xFu, 1.0f, T1, nullptr);
            t16_epilogue<false>(c1, net_b(a.tgt, 0), 2 * wq + 1, n, g, 0xFu, 1.0f, T1, nullptr);
            t16_epilogue<true>(c2, net_b(net, 0), 2 * wq, n, g, 0xFu, 1.0f, A1, G1);
            t16_epilogue<true>(c3, net_b(net, 0), 2 * wq + 1, n, g, 0xFu, 1.0f, A1, G1);
        }
        lds_barrier();
        QSTAMP(10);
        {   // layer 2 (+ Dropout on the training side, Player.py:194); T2 goes to Db (free: x' was layer 1's input)
            float wt0[8][4], wt1[8][4], wo0[8][4], wo1[8][4], bv[8][4];
            t16_w_load<8>(wt0, net_w(a.tgt, 1), 128, 128, 2 * wq, n, g); t16_w_load<8>(wt1, net_w(a.tgt, 1), 128, 128, 2 * wq + 1, n, g);
            t16_w_load<8>(wo0, net_w(net, 1), 128, 128, 2 * wq, n, g); t16_w_load<8>(wo1, net_w(net, 1), 128, 128, 2 * wq + 1, n, g);
            const uint32_t keep0 = t16_keep_bits(a.seed, gid, a.step, 0, 2 * wq, g, thr), keep1 = t16_keep_bits(a.seed, gid, a.step, 0, 2 * wq + 1, g, thr);
            t16_b_read<8>(bv, T1, n, g);
            f32x4t c0 = t16_zero(), c1 = t16_zero(), c2 = t16_zero(), c3 = t16_zero();
            t16_mul2<8>(wt0, bv, wt1, bv, c0, c1);
            t16_b_read<8>(bv, A1, n, g);
            t16_mul2<8>(wo0, bv, wo1, bv, c2, c3);
            t16_epilogue<false>(c0, net_b(a.tgt, 1), 2 * wq, n, g, 0xFu, 1.0f, T2, nullptr);
            t16_epilogue<false>(c1, net_b(a.tgt, 1), 2 * wq + 1, n, g, 0xFu, 1.0f, T2, nullptr);
            t16_epilogue<true>(c2, net_b(net, 1), 2 * wq, n, g, keep0, scale, A2, G2);
            t16_epilogue<true>(c3, net_b(net, 1), 2 * wq + 1, n, g, keep1, scale, A2, G2);
        }
        lds_barrier();
        QSTAMP(11);
        {   // layer 3 (+ Dropout, :197): tile wq of both networks, side by side
            float wa0[8][4], wa1[8][4], bv0[8][4], bv1[8][4];
            t16_w_load<8>(wa0, net_w(a.tgt, 2), 128, 64, wq, n, g); t16_w_load<8>(wa1, net_w(net, 2), 128, 64, wq, n, g);
            const uint32_t keep3 = t16_keep_bits(a.seed, gid, a.step, 16, wq, g, thr);      // (the Philox rounds run under the weights' round trip)
            t16_b_read<8>(bv0, T2, n, g); t16_b_read<8>(bv1, A2, n, g);
            f32x4t c0 = t16_zero(), c1 = t16_zero();
            t16_mul2<8>(wa0, bv0, wa1, bv1, c0, c1);
            t16_epilogue<false>(c0, net_b(a.tgt, 2), wq, n, g, 0xFu, 1.0f, T3, nullptr);
            t16_epilogue<true>(c1, net_b(net, 2), wq, n, g, keep3, scale, A3, G3);
        }
        lds_barrier();
        QSTAMP(12);
        {   // layer 4: wavefronts 0, 1 the target network's two tiles, 2, 3 the trained network's
            const int mo = wq & 1;
            float wa[4][4], bv[4][4];
            f32x4t c0 = t16_zero();
            if (wq < 2) { t16_w_load<4>(wa, net_w(a.tgt, 3), 64, 32, mo, n, g); t16_b_read<4>(bv, T3, n, g); }
            else { t16_w_load<4>(wa, net_w(net, 3), 64, 32, mo, n, g); t16_b_read<4>(bv, A3, n, g); }
            t16_mul1<4>(wa, bv, c0);
            // (T4 goes to Db, which layer 3 read as T2 -- a barrier ago; T3 = Da is only read in this phase)
            if (wq < 2) t16_epilogue<false>(c0, net_b(a.tgt, 3), mo, n, g, 0xFu, 1.0f, T4, nullptr);
            else t16_epilogue<true>(c0, net_b(net, 3), mo, n, g, 0xFu, 1.0f, A4, G4);
        }
        lds_barrier();
        QSTAMP(13);
        f32x4t qv = t16_zero();                                   // layer 5: wavefront 0 Q_target(s', .), wavefront 1 Q(s, .) (actions 4 g + r of row n)
        if (wq < 2) {
            float wa[2][4], bv[2][4];
            t16_w_load<2>(wa, wq == 0 ? net_w(a.tgt, 4) : net_w(net, 4), 32, A, 0, n, g);
            t16_b_read<2>(bv, wq == 0 ? T4 : A4, n, g);
            t16_mul1<2>(wa, bv, qv);
            const float* b5p = wq == 0 ? net_b(a.tgt, 4) : net_b(net, 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) qv[r] += (4 * g + r < A) ? b5p[4 * g + r] : 0.0f;
        }
        if (wq == 0) {                                            // max_a' Q_target(s', a') per row -> Tg
            float best = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < A) best = fmaxf(best, qv[r]);
            best = fmaxf(best, __shfl_xor(best, 16)); best = fmaxf(best, __shfl_xor(best, 32));
            if (g == 0) Tg[n] = best;
        }
        lds_barrier();                                            // (also: T4 = Db has been read, delta_4 may go there)
        QSTAMP(2);
        if (wq == 1) {                                            // delta_5 and the loss terms (:270-279)
            const float target = row_reward + a.gamma * Tg[n] * (1.0f - row_done);
            float qa = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) qa += (4 * g + r == act) ? qv[r] : 0.0f;
            qa += __shfl_xor(qa, 16); qa += __shfl_xor(qa, 32);
            const float td = live ? qa - target : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) Da[(4 * g + r) * kP16 + n] = (4 * g + r == act) ? 2.0f * td : 0.0f;
            float sq = (g == 0) ? td * td : 0.0f, cnt = (g == 0 && live) ? 1.0f : 0.0f;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off); cnt += __shfl_xor(cnt, off); }
            rows_sum += cnt; sq_sum += sq;                        // (lanes 0..15 hold the sums; lane 0 is read at the end)
        }
        lds_barrier();
        QSTAMP(3);
        // ---- backward: delta_5 in Da.  Every phase: delta tiles of the layer below AND weight-gradient blocks of this layer
        {   // layer 5: delta_4 tiles 0, 1 -> Db on wavefronts 0, 1 | dW5 blocks (., 0), (., 1) on wavefronts 2, 3
            if (wq < 2) {
                float wa[1][4], bv[1][4];
                t16_back_load<1>(wa, net_w(net, 4), A, 32, wq, n, g);
                t16_b_read<1>(bv, Da, n, g);
                f32x4t c0 = t16_zero();
                t16_mul1<1>(wa, bv, c0);
                t16_back_store(c0, wq, n, g, G4, Db);
            } else {
                const float bs = t16_dw<1>(Da, A4, part_t, kS16Blk5 + (wq - 2), 0, wq - 2, n, g, lane, first);
                if (wq == 2) b5 += t16_rowsum(bs);
            }
        }
        lds_barrier();
        QSTAMP(4);
        {   // layer 4 (delta_4 in Db): delta_3 tile wq -> Da | dW4 blocks (0, wq), (1, wq)
            float wa[2][4], bv[2][4];
            t16_back_load<2>(wa, net_w(net, 3), 32, 64, wq, n, g);
            t16_b_read<2>(bv, Db, n, g);
            const float bs0 = t16_dw<1>(Db, A3, part_t, kS16Blk4 + wq, 0, wq, n, g, lane, first);
            const float bs1 = t16_dw<1>(Db, A3, part_t, kS16Blk4 + 4 + wq, 1, wq, n, g, lane, first);
            if (wq == 0) { b4a += t16_rowsum(bs0); b4b += t16_rowsum(bs1); }
            f32x4t c0 = t16_zero();
            t16_mul1<2>(wa, bv, c0);
            t16_back_store(c0, wq, n, g, G3, Da);
        }
        lds_barrier();
        QSTAMP(5);
        {   // layer 3 (delta_3 in Da): delta_2 tiles 2 wq, 2 wq + 1 -> Db | dW3 blocks (wq, 0..7)
            float wa0[4][4], wa1[4][4], bv[4][4];
            t16_back_load<4>(wa0, net_w(net, 2), 64, 128, 2 * wq, n, g); t16_back_load<4>(wa1, net_w(net, 2), 64, 128, 2 * wq + 1, n, g);
            t16_b_read<4>(bv, Da, n, g);
            float bs = t16_dw<4>(Da, A2, part_t, kS16Blk3 + 8 * wq, wq, 0, n, g, lane, first);
            t16_dw<4>(Da, A2, part_t, kS16Blk3 + 8 * wq + 4, wq, 4, n, g, lane, first);
            b3 += t16_rowsum(bs);
            f32x4t c0 = t16_zero(), c1 = t16_zero();
            t16_mul2<4>(wa0, bv, wa1, bv, c0, c1);
            t16_back_store(c0, 2 * wq, n, g, G2, Db); t16_back_store(c1, 2 * wq + 1, n, g, G2, Db);
        }
        lds_barrier();
        QSTAMP(6);
        {   // layer 2 (delta_2 in Db): delta_1 tiles 2 wq, 2 wq + 1 -> Da | dW2 blocks (2 wq, 0..7), (2 wq + 1, 0..7)
            float wa0[8][4], wa1[8][4], bv[8][4];
            t16_back_load<8>(wa0, net_w(net, 1), 128, 128, 2 * wq, n, g); t16_back_load<8>(wa1, net_w(net, 1), 128, 128, 2 * wq + 1, n, g);
            t16_b_read<8>(bv, Db, n, g);
            float bsa = t16_dw<4>(Db, A1, part_t, kS16Blk2 + 8 * (2 * wq), 2 * wq, 0, n, g, lane, first);
            t16_dw<4>(Db, A1, part_t, kS16Blk2 + 8 * (2 * wq) + 4, 2 * wq, 4, n, g, lane, first);
            float bsb = t16_dw<4>(Db, A1, part_t, kS16Blk2 + 8 * (2 * wq + 1), 2 * wq + 1, 0, n, g, lane, first);
            t16_dw<4>(Db, A1, part_t, kS16Blk2 + 8 * (2 * wq + 1) + 4, 2 * wq + 1, 4, n, g, lane, first);
            b2a += t16_rowsum(bsa); b2b += t16_rowsum(bsb);
            f32x4t c0 = t16_zero(), c1 = t16_zero();
            t16_mul2<8>(wa0, bv, wa1, bv, c0, c1);
            t16_back_store(c0, 2 * wq, n, g, G1, Da); t16_back_store(c1, 2 * wq + 1, n, g, G1, Da);
        }
        lds_barrier();
        QSTAMP(7);
        {   // layer 1 (delta_1 in Da): dW1 blocks (2 wq, 0..MT1-1), (2 wq + 1, 0..MT1-1)
            const float bsa = t16_dw<MT1>(Da, Xs, part_t, kS16Blk1 + 4 * (2 * wq), 2 * wq, 0, n, g, lane, first);
            const float bsb = t16_dw<MT1>(Da, Xs, part_t, kS16Blk1 + 4 * (2 * wq + 1), 2 * wq + 1, 0, n, g, lane, first);
            b1a += t16_rowsum(bsa); b1b += t16_rowsum(bsb);
        }
    }

    QSTAMP(8);
    // db rows: every bias is written by exactly one wavefront (lanes 0..15 = its units)
    if (used && lane0 < 16) {
        float* pb = part + kS16Bias;
        pb[16 * (2 * wq) + lane0] = b1a; pb[16 * (2 * wq + 1) + lane0] = b1b;
        pb[128 + 16 * (2 * wq) + lane0] = b2a; pb[128 + 16 * (2 * wq + 1) + lane0] = b2b;
        pb[256 + 16 * wq + lane0] = b3;
        if (wq == 0) { pb[320 + lane0] = b4a; pb[336 + lane0] = b4b; }
        if (wq == 2) { pb[352 + lane0] = b5; pb[368 + lane0] = 0.0f; }
    }
    float* wave_reward = Tg + 16;                                 // (16 spare words)
    lds_barrier();
    if (lane0 == 0) wave_reward[wq] = reward_sum;
    if (wq == 1 && lane0 == 0) { wave_reward[8] = rows_sum; wave_reward[9] = sq_sum; }
    lds_barrier();
    if (wq == 0 && lane0 == 0) {
        float* ps = part + kS16Stats;
        ps[0] = wave_reward[8]; ps[1] = wave_reward[9]; ps[2] = (wave_reward[0] + wave_reward[1]) + (wave_reward[2] + wave_reward[3]);
        ps[3] = used ? 1.0f : 0.0f;
    }
    QSTAMP(9);
}
