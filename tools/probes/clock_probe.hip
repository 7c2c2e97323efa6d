// Shader clock during short bursts: ratio of s_memtime (clock64) to the 100 MHz s_memrealtime (wall_clock64) over a
// dependent-FMA loop of known length, for a few kernel lengths and launch spacings.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void probe(long long* out, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x * 1e-9f;
    for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0000001f, 1e-9f);      // one dependent v_fma per iteration
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)(x * 0.0f); }
}
int main() {
    long long* d; hipMalloc(&d, 64); long long h[3];
    for (int iters : {2000, 20000, 200000, 2000000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, iters);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("iters %8d: memtime %10lld ticks, realtime %8lld ticks (100 MHz) -> %.1f us, memtime rate %.3f GHz, %.2f ticks/iter\n", iters, h[0], h[1],
                   h[1] / 100.0, h[0] / (h[1] * 10.0), (double)h[0] / iters);
            usleep(200);
        }
    }
    return 0;
}
