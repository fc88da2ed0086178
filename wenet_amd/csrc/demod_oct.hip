// demod_oct.hip -- instantiations and launcher of the batch demodulator with one wavefront per capture (demod_oct_impl.h).
#include "demod_oct_impl.h"

extern "C" hipError_t wr_launch_demod_oct(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream, int fast) {
    if (nchan <= 0) return hipSuccess;
    if (!cfg->o_ok) return hipErrorInvalidValue;
    const int groups = (nchan + cfg->o_caps - 1) / cfg->o_caps;
    const int threads = (cfg->o_caps + (fast ? 0 : 1)) * 64;
#define WO_LAUNCH(TT, FF)                                                                                                          \
    do {                                                                                                                           \
        hipError_t e = hipFuncSetAttribute((const void *)wenet_demod_oct_kernel<2, TT, FF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           cfg->o_lds_bytes);                                                                      \
        if (e != hipSuccess) return e;                                                                                             \
        hipLaunchKernelGGL((wenet_demod_oct_kernel<2, TT, FF>), dim3(groups), dim3(threads), cfg->o_lds_bytes, stream, *cfg, d_chans, nchan); \
    } while (0)
    if (cfg->Ts == 10) { if (fast) WO_LAUNCH(10, true); else WO_LAUNCH(10, false); }
    else               { if (fast) WO_LAUNCH(8, true);  else WO_LAUNCH(8, false); }
#undef WO_LAUNCH
    return hipGetLastError();
}
