// Hardware-side reproducer attempt (round 5): wenet_decode_kernel's packet loop with its LDS traffic, nothing else.  Persistent workgroups of eight wavefronts take slots from a
// global counter through the two-cell claim in LDS; a tenth of the slots is empty (`continue` right behind thread 0's claim store: ds_write_b32, s_branch, s_barrier -- no wait,
// as hipcc 7.2 compiled it); the others cost a few barrier-separated passes over the message array with random table reads beside them.  Every wavefront folds the slots it READ
// into a scalar hash; at the end the eight hashes of a workgroup must agree.  A wavefront that once reads the claim cell stale takes another slot sequence: hashes differ.
//   MODE 0: as compiled by hipcc 7.2 (inline asm keeps the store / barrier sequence);  MODE 1: s_waitcnt lgkmcnt(0) between the store and the barrier
// Build: hipcc --offload-arch=gfx950 -O3 -o claim_cell_race claim_cell_race.hip ; run: ./claim_cell_race [slots in thousands = 20000] [launches = 3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_BYTES 39488
#define CLAIM_OFF 39424
#define NMSG (14 * 516)

template <int MODE>
__global__ __launch_bounds__(512, 8) void k(const unsigned char *full, unsigned *counter, long long nslots, unsigned *hashes, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *msg = (float *)smem;
    unsigned *tab = (unsigned *)(smem + NMSG * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned claim_addr = CLAIM_OFF;
    for (int i = tid; i < 2562; i += 512) tab[i] = (unsigned)i * 2654435761u;
    for (int i = tid; i < NMSG; i += 512) msg[i] = (float)i;
    if (tid == 0) { const unsigned s0 = atomicAdd(counter, 1u); ((int *)(smem + CLAIM_OFF))[0] = (long long)s0 < nslots ? (int)s0 : -1; }
    unsigned h = 0, s = (unsigned)tid * 2654435761u + blockIdx.x * 40503u;
    float acc = 0.f;
    int cur = 0;
    for (;; cur ^= 1) {
        // the loop header: barrier, then every wavefront reads the claim
        unsigned seen;
        asm volatile("s_barrier\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(claim_addr + (unsigned)cur * 4u) : "memory");
        const int slot_i = __builtin_amdgcn_readfirstlane((int)seen);
        if (slot_i < 0 || slot_i >= nslots) break;
        h = h * 31u + (unsigned)slot_i;
        unsigned nxt = 0;
        if (tid == 0) nxt = atomicAdd(counter, 1u);                      // (returns long after: thread 0's wavefront arrives last at the next barrier of the empty path)
        const int f = __builtin_amdgcn_readfirstlane((int)full[slot_i]);
        const unsigned other = claim_addr + (unsigned)(cur ^ 1) * 4u;
        if (f == 0) {                                                    // empty slot: the claim store, then straight round to the header's barrier
            if (tid == 0) {
                const int v = (long long)nxt < nslots ? (int)nxt : -1;
                if (MODE == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(other), "v"(v) : "memory");
                else asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(other), "v"(v) : "memory");
            }
            continue;
        }
        // a packet: `iters` iterations of two passes over the message array, barrier-separated, random table reads beside them (the decoder's LDS load, none of its arithmetic)
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kk = 0; kk < 14; kk++) {
                s = s * 1664525u + 1013904223u;
                acc += msg[kk * 516 + tid] + __uint_as_float(tab[(s >> 12) % 2562u] & 0x3fffffffu);
            }
#pragma unroll
            for (int kk = 0; kk < 14; kk++) msg[kk * 516 + tid] = acc * 0.5f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 15; kk++) {
                s = s * 1664525u + 1013904223u;
                const unsigned a = (s >> 10) % (unsigned)NMSG;
                acc += msg[a] + __uint_as_float(tab[(s >> 14) % 2562u] & 0x3fffffffu);
                msg[a] = acc * 0.25f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (tid == 0) { const int v = (long long)nxt < nslots ? (int)nxt : -1; asm volatile("ds_write_b32 %0, %1" :: "v"(other), "v"(v) : "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the normal path: two more barriers with waits lie between this store and its readers)
        smem[2592 + (tid & 255)] = (unsigned char)(int)acc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (lane == 0) hashes[blockIdx.x * 8 + (tid >> 6)] = h ^ (acc == 12345.f ? 1u : 0u);
}

template <int MODE> void run(const unsigned char *d_full, unsigned *d_counter, long long nslots, unsigned *d_hash, int launches, int iters) {
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    long long bad_wg = 0, wgs = 0;
    float ms_total = 0;
    for (int l = 0; l < launches; l++) {
        (void)hipMemset(d_counter, 0, 4); (void)hipMemset(d_hash, 0, 1024 * 8 * 4);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), LDS_BYTES, 0, d_full, d_counter, nslots, d_hash, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms_total += ms;
        std::vector<unsigned> h(1024 * 8);
        (void)hipMemcpy(h.data(), d_hash, h.size() * 4, hipMemcpyDeviceToHost);
        for (int g = 0; g < 1024; g++) {
            bool same = true;
            for (int w = 1; w < 8; w++) same = same && h[g * 8 + w] == h[g * 8];
            bad_wg += same ? 0 : 1; wgs++;
        }
    }
    printf("%s: %d launches x %lld slots (a tenth empty), %d iterations per packet: %lld of %lld workgroups ended with wavefronts that had read DIFFERENT slot sequences; %.0f ms\n",
           MODE == 0 ? "claim store, s_barrier (no wait)" : "claim store, s_waitcnt lgkmcnt(0), s_barrier", launches, nslots, iters, bad_wg, wgs, ms_total);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const long long nslots = 1000ll * (argc > 1 ? atoll(argv[1]) : 20000);
    const int launches = argc > 2 ? atoi(argv[2]) : 3;
    std::vector<unsigned char> full(nslots);
    unsigned s = 12345u;
    for (long long i = 0; i < nslots; i++) { s = s * 1664525u + 1013904223u; full[i] = (s >> 24) % 10u != 0u; }
    unsigned char *d_full; unsigned *d_counter, *d_hash;
    (void)hipMalloc(&d_full, nslots); (void)hipMalloc(&d_counter, 4); (void)hipMalloc(&d_hash, 1024 * 8 * 4);
    (void)hipMemcpy(d_full, full.data(), nslots, hipMemcpyHostToDevice);
    for (int iters : {1, 6}) {
        run<0>(d_full, d_counter, nslots / (iters == 6 ? 4 : 1), d_hash, launches, iters);
        run<1>(d_full, d_counter, nslots / (iters == 6 ? 4 : 1), d_hash, launches, iters);
    }
    return 0;
}
