// Development micro-benchmark (round 5): the decoder's cross-wavefront protocol in isolation -- does a wavefront ever read another value from an LDS cell than the other
// wavefronts of its workgroup, behind a workgroup barrier?
//
// Geometry of wenet_decode_kernel: 512-thread workgroups (eight wavefronts), four per CU, ~39 KB of LDS each.  Per iteration every wavefront does "work" (random 4-byte
// table reads; optionally a 12- or 16-byte read of a second table on the lanes that met a marked cell, i.e. under a partial EXEC mask, or on all lanes), lane 0 of every
// wavefront adds its share to the cell red[parity] (ds_add_u32), workgroup barrier, every lane reads the cell (one address for the whole wavefront), thread 0 clears the
// cell of the other parity, more work, and the value read is compared on every lane -- in the decoder a wavefront that sees another count than 516 stays in the iteration
// loop while the other seven leave it (tools/experiments/README.md, round 5).  The expected value is known here, so a wrong lane is caught at once.
//   MODE 0: no wide reads   1: ds_read_b96 on marked lanes (partial EXEC)   2: ds_read_b128 on marked lanes   3: ds_read_b96 on all lanes   4: ds_read_b128 on all lanes
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_cell_protocol lds_cell_protocol.hip ; run: ./lds_cell_protocol [iterations per launch] [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define T1_OFF 28896                 // float table[2562] (the decoder's first phi0 table)
#define T1_N   2562
#define T2_OFF 39152                 // uint4 second[16]
#define RED_OFF 39408                // int red[4]
#define LDS_BYTES 39488
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Err { unsigned it, wave, mask_lo, mask_hi, seen, expect, hwid, xcc; };

template <int MODE>
__global__ __launch_bounds__(512, 8) void k(Err *err, unsigned *nerr, unsigned long long *checks, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *t1 = (unsigned *)(smem + T1_OFF);
    v4u *t2 = (v4u *)(smem + T2_OFF);
    int *red = (int *)(smem + RED_OFF);
    unsigned *msg = (unsigned *)smem;                                    // 14 x 516 words of "messages": written lane-linearly, read at random
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T1_N; i += 512) t1[i] = (i % 183 == 7) ? (0x7fc00000u | (unsigned)(i & 15)) : (unsigned)i * 2654435761u;      // a few "marked" (NaN) cells
    if (tid < 16) t2[tid] = v4u{(unsigned)tid * 4 + 1, (unsigned)tid * 4 + 2, (unsigned)tid * 4 + 3, 0u};
    if (tid < 4) red[tid] = 0;
    for (int i = tid; i < 14 * 516; i += 512) msg[i] = (unsigned)i;
    __syncthreads();
    unsigned s = (unsigned)tid * 2654435761u + blockIdx.x * 40503u + seed;
    unsigned acc = 0, bad_wide = 0;
    unsigned long long nchk = 0;
    const int expect = 8 * 64 + 28;                                      // shares 64 + wave
    auto work = [&](int reads) __attribute__((always_inline)) {
        for (int u = 0; u < reads; u++) {
            s = s * 1664525u + 1013904223u;
            const unsigned key = (s >> 9) % T1_N;
            unsigned v = t1[key];
            const bool mk = (v & 0x7fc00000u) == 0x7fc00000u && MODE != 0;
            if (MODE == 1 || MODE == 2) {
                if (__builtin_amdgcn_ballot_w64(mk) != 0ull) {
                    if (mk) {
                        const unsigned a = T2_OFF + (v & 15u) * 16u;
                        if (MODE == 1) { v3u e; asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a) : "memory"); if (e.x != (v & 15u) * 4 + 1 || e.y != e.x + 1 || e.z != e.x + 2) bad_wide++; v = e.y; }
                        else { v4u e; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a) : "memory"); if (e.x != (v & 15u) * 4 + 1 || e.y != e.x + 1 || e.z != e.x + 2) bad_wide++; v = e.y; }
                    }
                }
            } else if (MODE == 3 || MODE == 4) {
                const unsigned a = T2_OFF + (mk ? (v & 15u) : (s >> 28)) * 16u;
                if (MODE == 3) { v3u e; asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a) : "memory"); if (e.y != e.x + 1 || e.z != e.x + 2) bad_wide++; v = mk ? e.y : v; }
                else { v4u e; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a) : "memory"); if (e.y != e.x + 1 || e.z != e.x + 2) bad_wide++; v = mk ? e.y : v; }
            }
            acc += v;
            msg[(u % 14) * 516 + tid] = acc;                             // lane-linear store (the check pass's)
            acc ^= msg[(s >> 12) % (14 * 516)];                          // random read (the variable pass's)
        }
    };
    for (int it = 0; it < iters; it++) {
        const int par = it & 1;
        work(6);
        if (lane == 0) atomicAdd(&red[par * 2], 64 + wave);
        __syncthreads();
        const int seen = red[par * 2];
        if (tid == 0) { red[(par ^ 1) * 2] = 0; red[(par ^ 1) * 2 + 1] = 0; }
        work(6);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(seen != expect);
        nchk++;
        if (m != 0ull && lane == (int)__builtin_ctzll(m)) {
            const unsigned i = atomicAdd(nerr, 1u);
            if (i < 256) err[i] = Err{(unsigned)it, (unsigned)wave, (unsigned)m, (unsigned)(m >> 32), (unsigned)seen, (unsigned)expect,
                                      (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11))};
        }
        __syncthreads();
    }
    if (bad_wide) atomicAdd(nerr + 1, bad_wide);
    if (lane == 0) atomicAdd(checks, nchk);
    if (acc == 0x12345678u) err[255].it = acc;                           // keep the work alive
}

template <int MODE> void run(int iters, int launches, Err *d_err, unsigned *d_nerr, unsigned long long *d_chk) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipMemset(d_nerr, 0, 8); hipMemset(d_chk, 0, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), LDS_BYTES, 0, d_err, d_nerr, d_chk, iters, 977u * (unsigned)l + 13u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned nerr[2]; unsigned long long chk; hipMemcpy(nerr, d_nerr, 8, hipMemcpyDeviceToHost); hipMemcpy(&chk, d_chk, 8, hipMemcpyDeviceToHost);
    static const char *names[5] = {"no wide reads", "ds_read_b96 on marked lanes", "ds_read_b128 on marked lanes", "ds_read_b96 on all lanes", "ds_read_b128 on all lanes"};
    printf("mode %d (%s): %llu wavefront checks of the cell, %u with a lane that saw another value, %u wrong wide reads, %.1f ms\n", MODE, names[MODE], chk, nerr[0], nerr[1], ms);
    if (nerr[0]) {
        std::vector<Err> h(256); hipMemcpy(h.data(), d_err, sizeof(Err) * 256, hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < nerr[0] && i < 12; i++)
            printf("    iteration %u wave %u (simd %u): lanes %08x%08x saw %u, expected %u; xcc %u se %u cu %u\n", h[i].it, h[i].wave, (h[i].hwid >> 4) & 3, h[i].mask_hi, h[i].mask_lo, h[i].seen,
                   h[i].expect, h[i].xcc & 15, (h[i].hwid >> 13) & 7, (h[i].hwid >> 8) & 15);
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, launches = argc > 2 ? atoi(argv[2]) : 4;
    Err *d_err; unsigned *d_nerr; unsigned long long *d_chk;
    hipMalloc(&d_err, sizeof(Err) * 256); hipMalloc(&d_nerr, 8); hipMalloc(&d_chk, 8);
    run<0>(iters, launches, d_err, d_nerr, d_chk);
    run<1>(iters, launches, d_err, d_nerr, d_chk);
    run<2>(iters, launches, d_err, d_nerr, d_chk);
    run<3>(iters, launches, d_err, d_nerr, d_chk);
    run<4>(iters, launches, d_err, d_nerr, d_chk);
    return 0;
}
