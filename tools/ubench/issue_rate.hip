// development micro-benchmark: VALU issue interval for one wave and for several waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 2.f, f = 3.f, g = 4.f, h = 5.f;
    v2f p = {a, b}, q = {1.0001f, 0.9999f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // 8 dependent v_add_f32
            asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                         "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {   // 8 independent v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(p.x) : "v"(b));
        } else if (MODE == 2) {   // 8 dependent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                         "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
        } else if (MODE == 3) {   // complex multiply chain, packed (2 mul + nop + add), x4
            v2f t1, t2;
            for (int u = 0; u < 4; u++)
                asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n s_nop 0\n"
                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q));
        } else if (MODE == 4) {   // complex multiply chain, scalar f32 (4 mul + sub + add), x4
            float t1, t2, t3, t4;
            for (int u = 0; u < 4; u++)
                asm volatile("v_mul_f32 %2, %0, %6\n v_mul_f32 %3, %1, %7\n v_mul_f32 %4, %0, %7\n v_mul_f32 %5, %1, %6\n"
                             "v_sub_f32 %0, %2, %3\n v_add_f32 %1, %4, %5" : "+v"(a), "+v"(c), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4) : "v"(q.x), "v"(q.y));
        } else if (MODE == 5) {   // 8 dependent ds_read (LDS latency)
            __shared__ int lds[256];
            if (i == 0) { lds[threadIdx.x & 255] = (threadIdx.x * 7 + 1) & 255; __syncthreads(); }
            int idx = threadIdx.x & 255;
            for (int u = 0; u < 8; u++) idx = lds[idx];
            a += idx;
        } else if (MODE == 7) {   // packed cmul + nop, ds_write_b64 of every phasor (compiler-scheduled store)
            extern __shared__ v2f ldsb[];
            v2f t1, t2;
            v2f *row = ldsb + (threadIdx.x & 63) * 4;     // (only 2 lanes are active in the real kernel)
            if ((threadIdx.x & 63) < 2) {
            for (int u = 0; u < 4; u++) {
                row[u + 8 * (i & 31)] = p;
                asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n s_nop 0\n"
                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q));
            } }
        } else if (MODE == 8) {   // the store sits in the hazard slot instead of the nop (asm ds_write)
            extern __shared__ v2f ldsb[];
            v2f t1, t2;
            unsigned addr = (unsigned)(((threadIdx.x & 63) * 4 + 8 * (i & 31)) * 8);
            if ((threadIdx.x & 63) < 2) {
            for (int u = 0; u < 4; u++) {
                asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n ds_write_b64 %4, %0\n"
                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q), "v"(addr) : "memory");
                addr += 8;
            } }
        } else if (MODE == 9) {   // lanes<2 predicate only, no store (cost of the exec mask itself)
            v2f t1, t2;
            if ((threadIdx.x & 63) < 2) {
            for (int u = 0; u < 4; u++)
                asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n s_nop 0\n"
                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q));
            }
        } else if (MODE == 10) {  // complex multiply chain on a lane PAIR (re in even lane, im in odd lane): mul, nop, mul_dpp, add
            float t1, t2;
            for (int u = 0; u < 4; u++)
                asm volatile("v_mul_f32 %1, %0, %3\n s_nop 0\n v_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32 %0, %1, %2" : "+v"(a), "=&v"(t1), "=&v"(t2) : "v"(b), "v"(c));
        } else if (MODE == 11) {  // lane pair, both products local, swap inside the add: mul, mul, nop 1, add_dpp
            float t1, t2;
            for (int u = 0; u < 4; u++)
                asm volatile("v_mul_f32 %2, %0, %4\n v_mul_f32 %1, %0, %3\n s_nop 0\n v_add_f32_dpp %0, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(a), "=&v"(t1), "=&v"(t2) : "v"(b), "v"(c));
        } else if (MODE == 12) {  // dependent v_mul_f32 x8
            asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                         "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 13 || MODE == 14 || MODE == 15) {   // packed cmul / dependent plain adds with EXEC narrowed ONCE (no branches in the loop)
            if (i == 0) {
                if (MODE == 13 || MODE == 15) asm volatile("s_mov_b64 exec, 3" ::: "memory");
                else asm volatile("s_mov_b64 exec, 0xffffffff" ::: "memory");          // lower 32 lanes only
            }
            v2f t1, t2;
            if (MODE == 15) {
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
            } else {
                for (int u = 0; u < 4; u++)
                    asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n"
                                 "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q));
            }
            if (i == iters - 1) asm volatile("s_mov_b64 exec, -1" ::: "memory");
        } else if (MODE == 6) {   // packed cmul without the nop
            v2f t1, t2;
            for (int u = 0; u < 4; u++)
                asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[0,1]\n"
                             "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p), "=&v"(t1), "=&v"(t2) : "v"(q));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e + f + g + h + p.x + p.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int ops_per_iter) {
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8 * 4096);
    const int iters = 20000;
    for (int waves : {1, 4, 8, 16}) {       // waves per block (block = one CU here: 1 block)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 32768, 0, out, cyc, 100);
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 32768, 0, out, cyc, iters); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/CU %2d: %6.2f ticks/op (wave0)  %7.2f ns/op/wave  aggregate %7.1f Mops/s\n", name, waves,
               (double)c / iters / ops_per_iter, ms * 1e6 / iters / ops_per_iter, (double)waves * iters * ops_per_iter / (ms * 1e3));
    }
}
int main() {
    run<0>("dep v_add_f32", 8);
    run<1>("indep v_add_f32", 8);
    run<2>("dep v_pk_add_f32", 8);
    run<3>("cmul packed+nop (per step)", 4);
    run<6>("cmul packed no nop", 4);
    run<4>("cmul scalar (per step)", 4);
    run<12>("dep v_mul_f32", 8);
    run<13>("cmul packed, exec=2 lanes", 4);
    run<14>("cmul packed, exec=32 lanes", 4);
    run<15>("dep v_add_f32, exec=2 lanes", 8);
    run<10>("cmul lane-pair mul_dpp", 4);
    run<11>("cmul lane-pair add_dpp", 4);
    run<9>("cmul packed+nop, 2 lanes", 4);
    run<7>("cmul + ds_write (compiler)", 4);
    run<8>("cmul + ds_write in nop slot", 4);
    return 0;
}
