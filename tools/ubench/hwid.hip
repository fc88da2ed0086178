// development micro-benchmark: where do the waves of co-resident 512-thread workgroups land? (HW_REG_HW_ID / XCC_ID)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <set>
__global__ __launch_bounds__(512) void k(unsigned *out, int spin) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) { lds[threadIdx.x] = 1; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + wave) * 2] = hw; out[(blockIdx.x * 8 + wave) * 2 + 1] = xcc; }
}
int main() {
    const int nwg = 768;
    unsigned *d; hipMalloc(&d, nwg * 8 * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 52000);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 52000, 0, d, 20000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu2wg;
    for (int g = 0; g < nwg; g++) {
        unsigned hw = h[g * 16], xcc = h[g * 16 + 1] & 0xf;
        unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8);
        cu2wg[cu].push_back(g);
    }
    printf("distinct CUs %zu\n", cu2wg.size());
    int shown = 0;
    for (auto &kv : cu2wg) {
        if (shown++ >= 4) break;
        printf("CU key %#x:", kv.first);
        for (int g : kv.second) {
            printf("  [wg %d tg %u:", g, (h[g * 16] >> 16) & 0xf);
            for (int w = 0; w < 8; w++) printf(" s%u.w%u", (h[(g * 8 + w) * 2] >> 4) & 3, h[(g * 8 + w) * 2] & 0xf);
            printf("]");
        }
        printf("\n");
    }
    std::map<int, int> hist; for (auto &kv : cu2wg) hist[(int)kv.second.size()]++;
    for (auto &kv : hist) printf("CUs with %d WGs: %d\n", kv.first, kv.second);
    return 0;
}
