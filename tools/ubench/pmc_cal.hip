// development micro-benchmark: what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU count for plain, packed-f32 and DPP operations
// (one wave, 2^20 iterations of 8 dependent operations each; run under rocprofv3 --pmc ...)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void cal_plain(float *out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f;
    for (int i = 0; i < iters; i++)
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    out[threadIdx.x] = a;
}
__global__ void cal_packed(float *out, int iters) {
    v2f p = {threadIdx.x * 1e-3f + 1.0f, 2.f}, q = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; i++)
        asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                     "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    out[threadIdx.x] = p.x + p.y;
}
__global__ void cal_dpp(float *out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f;
    for (int i = 0; i < iters; i++)
        asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a) : "v"(b));
    out[threadIdx.x] = a;
}
int main() {
    float *d; hipMalloc(&d, 4096);
    const int it = 1 << 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(cal_plain, dim3(1), dim3(64), 0, 0, d, it); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("plain  %.3f ms (8 x 2^20 = 8388608 operations)\n", ms);
    hipEventRecord(e0); hipLaunchKernelGGL(cal_packed, dim3(1), dim3(64), 0, 0, d, it); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("packed %.3f ms\n", ms);
    hipEventRecord(e0); hipLaunchKernelGGL(cal_dpp, dim3(1), dim3(64), 0, 0, d, it); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("dpp    %.3f ms\n", ms);
    return 0;
}
