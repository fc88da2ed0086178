// The BARE sequence (round 5): an LDS store issued right before s_barrier without a completion wait, read by the other wavefronts right behind the barrier.  This is what
// wenet_decode_kernel's empty-slot path came down to (thread 0's claim store: ds_write_b32, s_branch, s_barrier -- hipcc 7.2 emitted no s_waitcnt lgkmcnt(0) at that loop header,
// tools/ubench/syncthreads_loop_header.hip) and what made one wavefront in ~10^6 packets read the cell stale.  In THIS form -- steps of equal shape, light or heavy LDS load
// beside them -- 3*10^9 reads returned no stale value: the window needs more of the decoder's shape.  tools/ubench/claim_cell_race.hip (the packet loop's skeleton: a global
// atomic in front of the store, a tenth of the steps empty, barrier-separated passes over a message array otherwise) DOES reproduce it: profiles/r05_claim_cell_race.txt.
//
// Here: workgroups of eight wavefronts, four per CU.  Per step thread 0 stores the step number into cell[step & 1]; then s_barrier (mode 0: no wait, exactly the sequence
// above; mode 1: s_waitcnt lgkmcnt(0) first); every wavefront reads the cell and compares; a little LDS traffic of varying length between the steps so that the wavefronts
// arrive in changing order.  Counts wavefront-reads that returned something else than the step's value, by wavefront index.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_store_barrier_order lds_store_barrier_order.hip ; run: ./lds_store_barrier_order [steps per launch] [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LDS_BYTES 39488
#define CELL_OFF 39424

template <int MODE, bool HAMMER>
__global__ __launch_bounds__(512, 8) void k(unsigned long long *stale_by_wave, unsigned long long *reads, unsigned *example, int steps, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *tab = (unsigned *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 9000; i += 512) tab[i] = (unsigned)i * 2654435761u;
    if (tid < 2) ((unsigned *)(smem + CELL_OFF))[tid] = 0xffffffffu;
    __syncthreads();
    unsigned s = (unsigned)tid * 2654435761u + blockIdx.x * 40503u + seed, acc = 0;
    if (HAMMER && ((blockIdx.x * 2654435761u) >> 30) != 0u) {               // three workgroups in four only load the CU's LDS (random 16-byte reads, many in flight), as the decoder's table reads do
        const uint4 *t4 = (const uint4 *)smem;
        for (int it = 0; it < steps * 12; it++) {
            s = s * 1664525u + 1013904223u;
            const uint4 a4 = t4[(s >> 8) % 2200u], b4 = t4[(s >> 15) % 2200u];
            acc += a4.x + b4.y;
        }
        if (acc == 0x12345u) example[7] = acc;
        return;
    }
    unsigned long long stale = 0, n = 0;
    unsigned ex_step = 0, ex_seen = 0;
    const unsigned cell_addr = CELL_OFF;
    for (int step = 1; step <= steps; step++) {
        // a little work of varying length (per wavefront: the arrival order at the barrier changes from step to step)
        const int reps = HAMMER ? (wave == 0 ? 24 : (int)(s >> 29)) : (int)((s >> 27) + (unsigned)((wave * 7 + step) & 7));     // (HAMMER: the writer's wavefront arrives last, as in the decoder)
        for (int u = 0; u < reps; u++) { s = s * 1664525u + 1013904223u; acc += tab[(s >> 10) % 9000u]; }
        s = __builtin_amdgcn_readfirstlane((int)(s * 1664525u + 1013904223u)) + (unsigned)lane * 0x9E3779B1u;
        const unsigned a = cell_addr + (unsigned)(step & 1) * 4u;
        unsigned seen;
        if (MODE == 0) {
            if (tid == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"((unsigned)step) : "memory");
            asm volatile("s_barrier\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(a) : "memory");
        } else {
            if (tid == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"((unsigned)step) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(a) : "memory");
        }
        n++;
        if (__builtin_amdgcn_ballot_w64(seen != (unsigned)step) != 0ull) { stale++; ex_step = (unsigned)step; ex_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (everyone has read before thread 0 writes the other cell's successor two steps on)
    }
    if (lane == 0) {
        atomicAdd(&stale_by_wave[wave], stale); atomicAdd(reads, n);
        if (stale) { example[0] = ex_step; example[1] = ex_seen; example[2] = (unsigned)wave; example[3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
    }
    if (acc == 0x12345u) example[7] = acc;
}

template <int MODE, bool HAMMER> void run(int steps, int launches, unsigned long long *d_st, unsigned long long *d_rd, unsigned *d_ex) {
    (void)hipFuncSetAttribute((const void *)k<MODE, HAMMER>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipMemset(d_st, 0, 64); (void)hipMemset(d_rd, 0, 8); (void)hipMemset(d_ex, 0, 32);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL((k<MODE, HAMMER>), dim3(1024), dim3(512), LDS_BYTES, 0, d_st, d_rd, d_ex, steps, 977u * (unsigned)l + 13u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long st[8], rd; unsigned ex[8];
    (void)hipMemcpy(st, d_st, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(&rd, d_rd, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ex, d_ex, 32, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (int w = 0; w < 8; w++) tot += st[w];
    printf("%s%s: %llu wavefront reads behind the barrier, %llu returned another value than the one stored in front of it; by wavefront 0..7: %llu %llu %llu %llu %llu %llu %llu %llu; %.1f ms\n",
           HAMMER ? "[three workgroups in four loading the LDS, the writer last at the barrier] " : "", MODE == 0 ? "ds_write_b32, s_barrier, ds_read_b32 (no wait: what hipcc 7.2 emits for an LDS store in front of __syncthreads())" : "ds_write_b32, s_waitcnt lgkmcnt(0), s_barrier, ds_read_b32", rd, tot,
           st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], ms);
    if (tot) printf("    e.g. step %u: wavefront %u (simd %u) read %u\n", ex[0], ex[2], (ex[3] >> 4) & 3, ex[1]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 100000, launches = argc > 2 ? atoi(argv[2]) : 4;
    unsigned long long *d_st, *d_rd; unsigned *d_ex;
    (void)hipMalloc(&d_st, 64); (void)hipMalloc(&d_rd, 8); (void)hipMalloc(&d_ex, 32);
    run<0, false>(steps, launches, d_st, d_rd, d_ex);
    run<1, false>(steps, launches, d_st, d_rd, d_ex);
    run<0, true>(steps, launches, d_st, d_rd, d_ex);
    run<1, true>(steps, launches, d_st, d_rd, d_ex);
    run<0, true>(steps, launches, d_st, d_rd, d_ex);
    return 0;
}
