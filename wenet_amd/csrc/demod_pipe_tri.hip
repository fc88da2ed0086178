// demod_pipe_tri.hip -- three-captures-per-workgroup instantiations of the pipelined demod kernel (demod_tri_impl.h).
#include "demod_tri_impl.h"

extern "C" hipError_t wr_launch_demod_tri(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream) {
    if (nchan <= 0) return hipSuccess;
    const int groups = (nchan + WT_CAPS - 1) / WT_CAPS;
#define WT_LAUNCH(MM, LL)                                                                                                      \
    do {                                                                                                                       \
        wr_attr_ok(hipFuncSetAttribute((const void *)wenet_demod_tri_kernel<MM, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, cfg->p_lds_bytes)); \
        hipLaunchKernelGGL((wenet_demod_tri_kernel<MM, LL>), dim3(groups), dim3(WP_THREADS), cfg->p_lds_bytes, stream, *cfg, d_chans, nchan);    \
    } while (0)
    if (cfg->M == 2) { if (cfg->p_live) WT_LAUNCH(2, true); else WT_LAUNCH(2, false); }
    else { if (cfg->p_live) WT_LAUNCH(4, true); else WT_LAUNCH(4, false); }
#undef WT_LAUNCH
    return hipGetLastError();
}
