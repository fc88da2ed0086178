// Compiler-side minimal reproducer (round 5): hipcc 7.2 (--offload-arch=gfx950 -O3) emits the barrier that HEADS this packet loop without the s_waitcnt lgkmcnt(0) that
// __syncthreads()'s release fence asks for, although two paths reach it straight from an LDS store (the `continue` paths: ds_write_b32 claim ..., s_branch, s_barrier) and a third
// from the epilogue's byte store.  The loop is wenet_decode_kernel's skeleton (rounds 2-4): a claim cell written by thread 0, read by every wavefront behind the barrier.  A smaller
// loop of the same shape (one continue path, static LDS) keeps its wait, so the condition inside the wait-count pass is narrower than "loop header + back-edge store"; we did not
// establish it.  On gfx950 the missing wait is not harmless: another wavefront's read behind the barrier overtook the store about once in 10^6 packets (tools/experiments/README.md).
//   hipcc --offload-arch=gfx950 -O3 -c syncthreads_loop_header.hip -o /tmp/slh.o && python tools/isa_barrier_audit.py /tmp/slh.o
//   -> k   7 barriers, 1 reached by an LDS store without a completed wait:  s_barrier <- ds_write_b32 v4, v2 offset:39184
// (tests/test_isa_audit.py compiles this file to keep the audit tool honest; the product's kernels write the wait out: WR_LDS_BARRIER, lds_barrier.)
#include <hip/hip_runtime.h>
struct Args { const int *work; int *out; const unsigned long long *pbase; unsigned *counter; long long nslots; int stop; };
__global__ __launch_bounds__(512, 8) void k(Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *msg = (float *)smem;
    int *claim = (int *)(smem + 39184);
    const int tid = threadIdx.x;
    if (tid == 0) { const unsigned s0 = atomicAdd(A.counter, 1u); claim[0] = (long long)s0 < A.nslots ? (int)s0 : -1; }
    int cur = 0, acc = 0;
    for (;; cur ^= 1) {
        __syncthreads();                                  // <- loop header
        const int slot_i = __builtin_amdgcn_readfirstlane(claim[cur]);
        if (slot_i < 0) break;
        unsigned nxt = 0;
        if (tid == 0) nxt = atomicAdd(A.counter, 1u);
        auto put_claim = [&]() __attribute__((always_inline)) { if (tid == 0) claim[cur ^ 1] = (long long)nxt < A.nslots ? (int)nxt : -1; };
        const unsigned long long base = A.pbase[slot_i];
        if (base == 0ull) { put_claim(); continue; }
        if (A.stop) { put_claim(); continue; }
        if (tid == 0) msg[13 * 516] = 0.f;
        __syncthreads();
        msg[tid] = (float)slot_i;
        __syncthreads();
        for (int it = 0; it < 10; it++) {
            acc += (int)msg[(tid * 7 + it) & 511];
            __syncthreads();
            msg[(tid + it) & 511] = (float)acc;
            __syncthreads();
            if (acc & 1024) break;
        }
        put_claim();
        __syncthreads();
        msg[tid] = (float)acc;
        __syncthreads();
        ((unsigned char *)smem)[2592 + (tid & 255)] = (unsigned char)acc;
        A.out[slot_i * 512 + tid] = acc;
    }
}
