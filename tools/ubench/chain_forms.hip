// development micro-benchmark (round 3): what one wavefront's serial float recurrences cost on gfx950, and whether two independent
// recurrences in one instruction stream overlap.  Forms of the NCO step phi *= d (6 flops, each rounded once):
//   PK     packed, one lane per phasor: v_pk_mul x2, v_pk_add                             (3 VALU / step)
//   PAIR   re / im in neighbouring lanes: v_mul, s_nop, v_mul_dpp, v_add                  (3 VALU / step, what the duty wave runs today)
//   QUAD   the four products in four lanes: v_mul_dpp, s_nop 1, v_add_dpp, s_nop 1        (2 VALU / step)
// each alone and with one dependent v_add per step beside it (the ordered timing sum).  64 steps per loop trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
typedef float v2f __attribute__((ext_vector_type(2)));

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters, int prio) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0000001f, c = 0.5f, acc = 0.25f, t1, t2;
    float e0 = 1.f, e1 = 2.f, e2 = 3.f, e3 = 4.f, e4 = 5.f, e5 = 6.f, e6 = 7.f, e7 = 8.f;
    v2f p = {a, b}, q = {0.99999f, 0.001f}, u1, u2;
    if (prio) __builtin_amdgcn_s_setprio(2);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(R64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        else if (MODE == 1) asm volatile(R4(R4("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"))
                                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(b));
        else if (MODE == 2) asm volatile(R64("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n"
                                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]\n") : "+v"(p), "=&v"(u1), "=&v"(u2) : "v"(q));
        else if (MODE == 3) asm volatile(R64("v_mul_f32 %1, %0, %3\n s_nop 0\n v_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                             "v_add_f32 %0, %1, %2\n") : "+v"(a), "=&v"(t1), "=&v"(t2) : "v"(b), "v"(c));
        else if (MODE == 4) asm volatile(R64("v_mul_f32_dpp %1, %0, %2 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                                             "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a), "=&v"(t1) : "v"(b));
        // the same three with a dependent add per step in the hazard / latency slots
        else if (MODE == 5) asm volatile(R64("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n v_add_f32 %4, %4, %5\n"
                                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]\n") : "+v"(p), "=&v"(u1), "=&v"(u2) : "v"(q), "v"(acc), "v"(b));
        else if (MODE == 6) asm volatile(R64("v_mul_f32 %1, %0, %3\n v_add_f32 %5, %5, %3\n v_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                             "v_add_f32 %0, %1, %2\n") : "+v"(a), "=&v"(t1), "=&v"(t2), "+v"(acc) : "v"(b), "v"(c));
        else if (MODE == 7) asm volatile(R64("v_mul_f32_dpp %1, %0, %3 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %2, %2, %3\n s_nop 0\n"
                                             "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a), "=&v"(t1), "+v"(acc) : "v"(b));
        // QUAD with two independent chains interleaved (each fills the other's hazard slots): steps of chain A and chain B alternate
        else if (MODE == 8) asm volatile(R64("v_mul_f32_dpp %1, %0, %4 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n"
                                             "v_mul_f32_dpp %3, %2, %4 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                                             "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                             "v_add_f32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 0\n")
                                         : "+v"(a), "=&v"(t1), "+v"(c), "=&v"(t2) : "v"(b));
        // dependent adds alone, two independent accumulators interleaved
        else if (MODE == 9) asm volatile(R64("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") : "+v"(a), "+v"(acc) : "v"(b));
        // QUAD without any nop (WRONG results; the price of the wait states)
        else if (MODE == 10) asm volatile(R64("v_mul_f32_dpp %1, %0, %2 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n"
                                              "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a), "=&v"(t1) : "v"(b));
        // plain dependent mul, add pairs (no dpp): the latency floor of a 2-deep step
        else if (MODE == 11) asm volatile(R64("v_mul_f32 %1, %0, %2\n v_add_f32 %0, %1, %1\n") : "+v"(a), "=&v"(t1) : "v"(b));
    }
    long long t9 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + acc + p.x + p.y + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t9 - t0;
}

template <int MODE> void run(const char *name, int steps_per_iter) {
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8 * 4096);
    const int iters = 2000;
    for (int prio = 0; prio < 2; prio++)
        for (int waves : {1, 4, 8, 16}) {       // waves per block, one block: 1 = alone on a SIMD, 16 = four per SIMD
            if (prio && waves != 16) continue;
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 50, prio);
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters, prio);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-44s waves/CU %2d prio %d: %7.2f cycles/step (wave 0)\n", name, waves, prio, (double)c / iters / steps_per_iter);
        }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("dep v_add_f32", 64);
    run<1>("indep v_add_f32 (8 accumulators)", 64);
    run<9>("two dep v_add chains interleaved (per pair)", 64);
    run<11>("dep v_mul, v_add pair", 64);
    run<2>("NCO step PK", 64);
    run<3>("NCO step PAIR (today's duty wave)", 64);
    run<4>("NCO step QUAD", 64);
    run<10>("NCO step QUAD, no wait states (invalid)", 64);
    run<5>("NCO step PK + dep add", 64);
    run<6>("NCO step PAIR + dep add", 64);
    run<7>("NCO step QUAD + dep add", 64);
    run<8>("two QUAD chains interleaved (per step pair)", 64);
    return 0;
}
