// v_mfma_f32_16x16x4_f32 on gfx950: operand layout (which lane holds which (row, k) of A and (k, column) of B, which
// (register, lane) holds which (row, column) of D) and issue rate of independent / dependent chains.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma16x16_probe.hip -o /tmp/mfma16x16_probe && /tmp/mfma16x16_probe
// The 16-row forward of csrc/qnet_rows16.h is written against: A lane l = (m = l % 16, k = l / 16), B lane l = (n = l % 16,
// k = l / 16), D register r of lane l = (m = 4 (l / 16) + r, n = l % 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(l + 1), 1.0f, z, 0, 0, 0);     // sum over k of (A lane + 1)
    const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[(r * 64 + l) * 2] = da[r]; out[(r * 64 + l) * 2 + 1] = db[r]; }
    for (int q = 0; q < 4; ++q)
        for (int p = 0; p < 4; ++p) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32((l >> 4) == q ? 1.0f : 0.0f, (l >> 4) == p ? 1.0f : 0.0f, z, 0, 0, 0);
            if (l == 0) out[512 + q * 4 + p] = d[0];
        }
}

template <int CHAINS>
__global__ __launch_bounds__(1024) void rate(long long* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed * lane, b = seed;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (lane == 0) { out[(threadIdx.x >> 6) * 2] = t1 - t0; out[(threadIdx.x >> 6) * 2 + 1] = (long long)s; }
}

// VALU work of a second wavefront on the same SIMD beside a stream of MFMAs: does it hide behind them?
template <int CHAINS>
__global__ __launch_bounds__(512) void mixed(long long* out, int iters, float seed) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long t0 = clock64();
    float s = 0.f;
    if (wv < 4) {
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = seed * lane, b = seed;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
    } else {
        float x = seed * lane;
        for (int i = 0; i < iters * CHAINS * 4; ++i) x = fmaf(x, 1.0000001f, seed);        // 4 dependent VALU per MFMA of the partner
        s = x;
    }
    const long long t1 = clock64();
    if (lane == 0) { out[wv * 2] = t1 - t0; out[wv * 2 + 1] = (long long)s; }
}

int main() {
    float* d; (void)hipMalloc(&d, (512 + 16) * 4);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(512 + 16);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int r = 0; r < 4; ++r) {
        printf(" r%d:", r);
        for (int l = 0; l < 64; ++l) {
            const int m = ((int)h[(r * 64 + l) * 2] - 100) / 4, n = ((int)h[(r * 64 + l) * 2 + 1] - 100) / 4;
            if (l < 6 || (l >= 16 && l < 19) || l >= 61) printf(" %d->(m%d,n%d)", l, m, n);
            if (m != 4 * (l / 16) + r || n != l % 16) ok = 0;
        }
        printf("\n");
    }
    printf(" k pairing (A lanes of group q x B lanes of group p -> D): ");
    for (int q = 0; q < 4; ++q) for (int p = 0; p < 4; ++p) { printf("%d", (int)h[512 + q * 4 + p]); if ((int)h[512 + q * 4 + p] != (q == p)) ok = 0; }
    printf("\n hypothesis A(m = l %% 16, k = l / 16), B(n = l %% 16, k = l / 16), D[r](m = 4 (l / 16) + r, n = l %% 16): %s\n", ok ? "HOLDS" : "does NOT hold");
    long long* t; (void)hipMalloc(&t, 2 * 16 * 8);
    std::vector<long long> th(32);
    const int iters = 4096;
#define TIME(K, CH, WAVES, NAME) do { for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((K<CH>), dim3(1), dim3(64 * WAVES), 0, 0, t, iters, 1e-9f); \
        (void)hipMemcpy(th.data(), t, 2 * WAVES * 8, hipMemcpyDeviceToHost); } \
    printf("%-44s %2d waves:", NAME, WAVES); for (int w = 0; w < WAVES; w += (WAVES > 4 ? 4 : 1)) printf(" wave %d %.2f ticks/MFMA", w, th[2 * w] / (double)(iters * CH)); printf("\n"); } while (0)
    TIME(rate, 1, 1, "1 dependent chain");
    TIME(rate, 2, 1, "2 chains");
    TIME(rate, 4, 1, "4 chains");
    TIME(rate, 4, 8, "4 chains, 2 waves per SIMD");
    TIME(mixed, 4, 8, "4 chains + a VALU wave per SIMD (4 VALU / MFMA)");
    return 0;
}
