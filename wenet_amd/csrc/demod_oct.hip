// demod_oct.hip -- instantiations and launcher of the batch demodulator with one wavefront per capture (demod_oct_impl.h).
#include "demod_oct_impl.h"

extern "C" hipError_t wr_launch_demod_oct(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream) {
    if (nchan <= 0) return hipSuccess;
    if (!cfg->o_ok) return hipErrorInvalidValue;
    const int groups = (nchan + cfg->o_caps - 1) / cfg->o_caps;
    const int threads = (cfg->o_caps + 1) * 64;                        // the capture waves + the duty wave
#define WO_LAUNCH(MM, TT, NN)                                                                                                             \
    do {                                                                                                                           \
        hipError_t e = hipFuncSetAttribute((const void *)wenet_demod_oct_kernel<MM, TT, NN>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                           cfg->o_lds_bytes);                                                                      \
        if (e != hipSuccess) return e;                                                                                             \
        hipLaunchKernelGGL((wenet_demod_oct_kernel<MM, TT, NN>), dim3(groups), dim3(threads), cfg->o_lds_bytes, stream, *cfg, d_chans, nchan);     \
    } while (0)
    if (cfg->M == 2 && cfg->Ts == 10 && cfg->Ndft == 256)       WO_LAUNCH(2, 10, 256);
    else if (cfg->M == 2 && cfg->Ts == 8 && cfg->Ndft == 256)   WO_LAUNCH(2, 8, 256);
    else if (cfg->M == 4 && cfg->Ts == 32 && cfg->Ndft == 1024) WO_LAUNCH(4, 32, 1024);
    else return hipErrorInvalidValue;
#undef WO_LAUNCH
    return hipGetLastError();
}
