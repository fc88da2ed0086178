// demod_oct_sliced.hip -- the batch demodulator (demod_oct_impl.h) over TIME SLICES of every capture inside one launch (WrSliceCtl,
// wenet_internal.h): the SL instantiations and their launcher.  A translation unit of its own: the plain instantiations (demod_oct.hip) keep their
// register allocation, and the two compile side by side.
#include "demod_oct_impl.h"

extern "C" hipError_t wr_launch_demod_oct_sliced(const WrDemodCfg *cfg, WrChan *d_chans, int nchan, WrSliceCtl *d_ctl, int nslices, hipStream_t stream) {
    if (nchan <= 0) return hipSuccess;
    if (!cfg->o_ok || !d_ctl || nslices < 1 || cfg->o_hlp) return hipErrorInvalidValue;
    const int groups = (nchan + cfg->o_caps - 1) / cfg->o_caps;
    const int threads = (cfg->o_caps + cfg->o_nd) * 64;
    if (cfg->o_nd < 1 || cfg->o_nd > 2 || threads > 1024) return hipErrorInvalidValue;
    const int lds = cfg->o_lds_bytes + 16;                                // + the two words behind the tables (capture group, slice)
#define WO_LAUNCH(MM, TT, NN, DD)                                                                                                              \
    do {                                                                                                                           \
        hipError_t e = hipFuncSetAttribute((const void *)wenet_demod_oct_kernel<MM, TT, NN, DD, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return e;                                                                                             \
        hipLaunchKernelGGL((wenet_demod_oct_kernel<MM, TT, NN, DD, false, true>), dim3(groups * nslices), dim3(threads), lds, stream, *cfg, d_chans, nchan, d_ctl); \
    } while (0)
    if (cfg->M == 2 && cfg->Ts == 10 && cfg->Ndft == 256 && cfg->o_nd == 1)       WO_LAUNCH(2, 10, 256, 1);
    else if (cfg->M == 2 && cfg->Ts == 8 && cfg->Ndft == 256 && cfg->o_nd == 1)   WO_LAUNCH(2, 8, 256, 1);
    else if (cfg->M == 4 && cfg->Ts == 32 && cfg->Ndft == 1024 && cfg->o_nd == 2) WO_LAUNCH(4, 32, 1024, 2);
    else return hipErrorInvalidValue;
#undef WO_LAUNCH
    return hipGetLastError();
}
