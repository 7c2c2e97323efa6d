// How fast a CU gets a layer's weights out of L2, by access shape (the training launch waits for its weights).
// Every workgroup (8 wavefronts, one per CU-slot as in qnet_train8_kernel) reads the same 128 x 128 fp32 matrix (64 KB), each
// wavefront its 32 rows x 128 columns as 16 float4 per lane, in three shapes:
//   rows   lane (c, h) reads W[32 w + c][8 i + 4 h ..] -- the torch layout, 32 rows x 32 bytes per load instruction (load_layer)
//   rows16 lane (m, kk) reads W[16 t + m][16 mt + 4 kk ..] -- 16 rows x 64 bytes per instruction (the 16x16x4 operand)
//   packed lane l reads P[(w * 16 + i) * 64 + l] -- an operand-order image, 1 KB contiguous per instruction
// and reports the slowest wavefront's cycles for the 16 loads (all issued, then one wait), first touch and repeated.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/weight_fetch_probe.hip -o /tmp/wf && /tmp/wf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SHAPE>
__global__ __launch_bounds__(512) void fetch(const float* __restrict__ w, long long* out, int reps, float* sink) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.0f;
    long long worst = 0;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        const long long t0 = clock64();
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* p;
            const int t = wv & 3;                                  // (two groups of four wavefronts read the same matrix, as the two networks' groups do their own)
            if (SHAPE == 0) p = w + (size_t)(32 * t + (lane & 31)) * 128 + 8 * i + 4 * (lane >> 5);
            else if (SHAPE == 1) p = w + (size_t)(16 * (2 * t + (i >> 3)) + (lane & 15)) * 128 + 16 * (i & 7) + 4 * (lane >> 4);
            else p = w + ((size_t)(t * 16 + i) * 64 + lane) * 4;
            v[i] = *reinterpret_cast<const float4*>(p + (wv >> 2) * 16384);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (rep == reps - 1 || rep == 0) worst = t1 - t0 > worst || rep == 0 ? t1 - t0 : worst;
        if (rep == 0 && lane == 0) out[(blockIdx.x * 8 + wv) * 2] = t1 - t0;
        if (rep == reps - 1 && lane == 0) out[(blockIdx.x * 8 + wv) * 2 + 1] = t1 - t0;
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

int main() {
    float* w; (void)hipMalloc(&w, 2 * 16384 * 4 * 8); (void)hipMemset(w, 0, 2 * 16384 * 4 * 8);
    long long* out; (void)hipMalloc(&out, 256 * 8 * 2 * 8);
    float* sink; (void)hipMalloc(&sink, 4);
    std::vector<long long> h(256 * 8 * 2);
    const char* names[3] = {"rows   (32 rows x 32 B per instruction)", "rows16 (16 rows x 64 B per instruction)", "packed (1 KB contiguous per instruction)"};
    for (int pass = 0; pass < 2; ++pass)
        for (int s = 0; s < 3; ++s) {
            // a fresh region per launch so that the first repetition finds the L2 cold for it
            const float* base = w + (size_t)(pass * 3 + s) * 2 * 16384;
            if (s == 0) hipLaunchKernelGGL(fetch<0>, dim3(256), dim3(512), 0, 0, base, out, 8, sink);
            if (s == 1) hipLaunchKernelGGL(fetch<1>, dim3(256), dim3(512), 0, 0, base, out, 8, sink);
            if (s == 2) hipLaunchKernelGGL(fetch<2>, dim3(256), dim3(512), 0, 0, base, out, 8, sink);
            (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double f = 0, l = 0, fm = 0, lm = 0;
            for (int i = 0; i < 256 * 8; ++i) { f += h[2 * i]; l += h[2 * i + 1]; fm = h[2 * i] > fm ? h[2 * i] : fm; lm = h[2 * i + 1] > lm ? h[2 * i + 1] : lm; }
            printf("pass %d  %-44s first touch: mean %6.0f max %6.0f cycles   repeated: mean %6.0f max %6.0f  (64 KB per 4 wavefronts)\n", pass, names[s], f / 2048, fm,
                   l / 2048, lm);
        }
    return 0;
}
