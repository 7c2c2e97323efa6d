// v_mfma_f32_4x4x1_16B_f32 on gfx950: operand layout, the A-broadcast controls (CBSZ / ABID) and issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
// Part 1 prints, for every (accumulator register r, lane l), which lane's A value and which lane's B value the product
// came from -- the layout the 4-row forward of csrc/qnet_rows4.h is written against.  Part 2 times dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID, int BLGP>
__global__ void layout(float* out) {
    const int l = threadIdx.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, CBSZ, ABID, BLGP);    // A's source lane + 1
    const f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, CBSZ, ABID, BLGP);    // B's source lane + 1
    for (int r = 0; r < 4; ++r) { out[(r * 64 + l) * 2] = da[r]; out[(r * 64 + l) * 2 + 1] = db[r]; }
}

template <int CHAINS, bool LDSB>
__global__ __launch_bounds__(1024) void rate(long long* out, int iters, float seed) {
    __shared__ float4 w[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) w[i] = make_float4(seed, seed * 2, seed * 3, seed * 4);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed * lane;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        float4 b = make_float4(seed, seed, seed, seed);
        if (LDSB) b = w[(i & 15) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b.x, acc[c], 4, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b.y, acc[c], 4, 1, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b.z, acc[c], 4, 2, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b.w, acc[c], 4, 3, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (lane == 0) { out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = (long long)s; }
}

template <int CBSZ, int ABID, int BLGP> void show(const char* name, float* d) {
    hipLaunchKernelGGL((layout<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(4 * 64 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("== %s (cbsz %d abid %d blgp %d): (r, l) <- A lane, B lane\n", name, CBSZ, ABID, BLGP);
    int plain_ok = 1;
    for (int r = 0; r < 4; ++r) {
        printf(" r%d:", r);
        for (int l = 0; l < 64; ++l) {
            const int sa = (int)h[(r * 64 + l) * 2] - 1, sb = (int)h[(r * 64 + l) * 2 + 1] - 1;
            if (l < 12 || l >= 60) printf(" %d<-(%d,%d)", l, sa, sb);
            const int grp = 1 << CBSZ, blk = l / 4, src_blk = (blk / grp) * grp + (CBSZ ? ABID : blk % grp);
            if (sa != 4 * src_blk + r || (BLGP == 0 && sb != l)) plain_ok = 0;
        }
        printf("\n");
    }
    printf(" hypothesis D[r][l] = A[4 * source_block(l / 4) + r] * B[l]: %s\n", plain_ok ? "HOLDS" : "does NOT hold");
}

template <int CHAINS, bool LDSB> void time_it(const char* name, long long* d, int waves_per_wg, int wgs) {
    const int iters = 4096;
    std::vector<long long> h(2 * waves_per_wg * wgs);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate<CHAINS, LDSB>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, iters, 1e-9f);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    }
    double mx = 0; for (int i = 0; i < waves_per_wg * wgs; ++i) mx = h[2 * i] > mx ? h[2 * i] : mx;
    printf("%-34s %2d waves/WG x %4d WGs: %.2f ticks per MFMA per wave (slowest wave), %.2f per MFMA per SIMD\n", name, waves_per_wg, wgs,
           mx / (iters * 4.0 * CHAINS), mx / (iters * 4.0 * CHAINS) / ((waves_per_wg + 3) / 4));
}

int main() {
    float* d; hipMalloc(&d, 4 * 64 * 2 * 4);
    show<0, 0, 0>("plain", d);
    show<4, 3, 0>("A of block 3 to all 16 blocks", d);
    show<3, 2, 0>("A of block 2 of each 8", d);
    show<2, 1, 0>("A of block 1 of each 4", d);
    show<0, 0, 4>("B lane group 0 to all", d);
    long long* t; hipMalloc(&t, 2 * 16 * 1024 * 8);
    time_it<1, false>("1 chain, B in registers", t, 1, 1);
    time_it<2, false>("2 chains, B in registers", t, 1, 1);
    time_it<1, false>("1 chain, B in registers", t, 4, 1);
    time_it<1, false>("1 chain, B in registers", t, 8, 1);
    time_it<2, false>("2 chains, B in registers", t, 8, 1);
    time_it<1, true>("1 chain, B from LDS (b128 / 4)", t, 4, 1);
    time_it<1, true>("1 chain, B from LDS (b128 / 4)", t, 16, 1);
    time_it<2, true>("2 chains, one LDS read for both", t, 16, 1);
    time_it<1, true>("1 chain, B from LDS, whole chip", t, 16, 256);
    return 0;
}
