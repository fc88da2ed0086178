// development micro-benchmark (round 3): calibration of the VALU utilisation figure.
// Three kernels with a KNOWN instruction stream saturate every SIMD of the chip (8 wavefronts per SIMD, 64 dependent-free operations per loop
// trip): plain v_add_f32, packed v_pk_add_f32, and an even mix.  Run once bare (prints the wall time per wave-instruction and SIMD = the cycles a
// saturated SIMD needs per instruction) and under rocprofv3 --pmc (tools/gpu_pmc_cal.sh) to see what SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU,
// SQ_INST_CYCLES_VALU, SQ_BUSY_CYCLES ... count for them.  One wave alone (grid 1 x 64) is measured too: the issue interval of a single wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))

__global__ void cal_plain(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = 1.f, a2 = 2.f, a3 = 3.f, b = 1.0000001f;
    for (int i = 0; i < iters; i++)
        asm volatile(R16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void cal_packed(float *out, int iters) {
    v2f a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = {2.f, 3.f}, a2 = {4.f, 5.f}, a3 = {6.f, 7.f}, b = {1.0000001f, 0.9999999f};
    for (int i = 0; i < iters; i++)
        asm volatile(R16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}
__global__ void cal_mixed(float *out, int iters) {          // 32 plain + 32 packed per trip
    v2f a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = {2.f, 3.f}, b = {1.0000001f, 0.9999999f};
    float c0 = 1.f, c1 = 2.f;
    for (int i = 0; i < iters; i++)
        asm volatile(R16("v_pk_add_f32 %0, %0, %4\n v_add_f32 %2, %2, %5\n v_pk_add_f32 %1, %1, %4\n v_add_f32 %3, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1) : "v"(b), "v"(c1));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + c0 + c1;
}
__global__ void cal_dpp(float *out, int iters) {            // v_add_f32_dpp, independent accumulators (no hazard between them)
    float a0 = threadIdx.x * 1e-3f, a1 = 1.f, a2 = 2.f, a3 = 3.f, b = 1.0000001f;
    for (int i = 0; i < iters; i++)
        asm volatile(R16("v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %4, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <class K> void run(const char *name, K k, float *d, int blocks, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 16);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64, insts = waves * iters * 64;
    printf("%-10s grid %5d x %4d: %8.3f ms, %.4g wave-instructions, %.3f ns per wave-instruction per SIMD (1024 SIMDs), %.3f ns per instruction of one wave\n", name, blocks, threads, ms,
           insts, ms * 1e6 / (insts / 1024.0), ms * 1e6 / ((double)iters * 64));
}
int main() {
    float *d; hipMalloc(&d, 2048 * 512 * 4);
    const int it = 20000;
    for (int rep = 0; rep < 2; rep++) {
        run("plain", cal_plain, d, 2048, 512, it);      // 8 waves per SIMD
        run("packed", cal_packed, d, 2048, 512, it);
        run("mixed", cal_mixed, d, 2048, 512, it);
        run("dpp", cal_dpp, d, 2048, 512, it);
    }
    run("plain/1w", cal_plain, d, 1, 64, it * 8);
    run("packed/1w", cal_packed, d, 1, 64, it * 8);
    return 0;
}
