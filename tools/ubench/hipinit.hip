// development micro-benchmark: cost of bringing up the HIP runtime in a fresh process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k(int *p) { *p = 1; }
int main() {
    auto t0 = std::chrono::steady_clock::now();
    int n = 0; hipGetDeviceCount(&n);
    auto t1 = std::chrono::steady_clock::now();
    int *d; hipMalloc(&d, 4);
    auto t2 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d); hipDeviceSynchronize();
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    printf("hipGetDeviceCount %.1f ms, first hipMalloc %.1f ms, first launch %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
    return 0;
}
