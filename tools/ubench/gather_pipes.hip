// Micro-benchmark (development aid, not part of the product): what does a random 4-byte gather from a small read-only table cost per wave-instruction when it goes
// (a) through the LDS (ds_read_b32), (b) through the vector memory path (global_load_dword, table resident in the CU's L1), (c) both interleaved -- are the two pipes
// independent?  Geometry of the decoder: 512-thread workgroups, four per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_pipes gather_pipes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 642
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const int *tab_g, int *out, int iters, int spread) {
    __shared__ int tab_l[N];
    for (int i = threadIdx.x; i < N; i += 512) tab_l[i] = tab_g[i];
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    int acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = s * 1664525u + 1013904223u;
            const int key = (int)((s >> 20) & (unsigned)(spread - 1)) + 100;          // `spread` distinct keys (a power of two <= 512)
            if (MODE == 0) acc += tab_l[key];
            if (MODE == 1) acc += tab_g[key];
            if (MODE == 2) { if (u & 1) acc += tab_g[key]; else acc += tab_l[key]; }
            if (MODE == 3) { acc += tab_l[key]; acc += tab_g[key + 7]; }       // one of each per step
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE> float run(const int *tab, int *out, int iters, int spread) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, tab, out, 10, spread);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, tab, out, iters, spread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    int *tab, *out; hipMalloc(&tab, N * 4); hipMalloc(&out, 1024 * 512 * 4);
    std::vector<int> h(N); for (int i = 0; i < N; i++) h[i] = i * 3; hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int spread : {1, 32, 128, 512}) {
        const float t0 = run<0>(tab, out, iters, spread), t1 = run<1>(tab, out, iters, spread), t2 = run<2>(tab, out, iters, spread), t3 = run<3>(tab, out, iters, spread);
        // wave-instructions per CU: 1024 WGs x 8 waves x iters x 8 (x2 for mode 1: two loads; mode 3: one of each) / 256 CUs
        const double wi = 1024.0 * 8 * iters * 8 / 256;
        printf("spread %3d: LDS %.3f ms (%.1f ns per wave-gather per CU)  global %.3f ms (%.1f)  alternating %.3f ms (%.1f)  one of each %.3f ms (%.1f per pair)\n", spread, t0, t0 * 1e6 / wi,
               t1, t1 * 1e6 / wi, t2, t2 * 1e6 / wi, t3, t3 * 1e6 / wi);
    }
    return 0;
}
