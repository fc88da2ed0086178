// demod_pipe_impl.h -- the pipelined demod kernel template (included by demod_pipe_kernel.hip, demod_pipe_raw.hip and
// demod_pipe_prof.hip: one code object per group of instantiations, so that a process loads only what it launches --
// the drop-in fsk_demod executable never touches the batch or the instrumented variants).
#pragma once
// Pipelined M-FSK demodulator: eight wavefronts (512 threads) per capture.
//
// Why: on gfx950 a lone wavefront issues one VALU instruction every ~7.5-9 cycles no matter how much ILP
// it has (tools/ubench/issue_rate.hip), and the reference's frame loop is one long dependency chain
//     estimate tones -> NCO phasor chain -> mix/integrate -> timing sum -> nin -> next frame.
// The chain and the timing sum are float recurrences that must run in reference order (bit-exactness),
// so the only way to go faster on ONE capture is to overlap the stages of neighbouring frames:
//
//     wave 1      E(k+3)   tone estimator three frames ahead             (fsk.c:540-677)
//     wave 0      C(k+2)   NCO phasor chain two frames ahead (checkpoints) (fsk.c:756-764,781-824)
//     waves 3-7   D(k+1)   sample staging, chain replay + down-conversion, integrate-and-dump, timing products
//     wave 2      T(k)     ordered timing sum, atan2f, nin, resample/decide, soft decisions out (fsk.c:858-993)
//
// E, C and D of later frames need nin(k+1), which only T(k) produces; they run SPECULATIVELY with nin = N
// (true for >99 % of frames on a locked signal).  Every stage keeps its carried state in small rings
// (spectrum x4, NCO phase x3, tone bins x4, checkpoints x2, integrator outputs / timing products x2, samples in a
// 5-frame ring), so when T(k) reports nin(k+1) != N the speculative stages are simply re-run from the
// untouched state of frame k.  Results are bit-identical to the sequential kernel (demod_kernel.hip) and
// hence to the reference.
//
// Synchronisation: one workgroup barrier per frame; the five D waves meet at LDS-counter barriers so that
// the other waves are never stalled inside their long serial loops.
#include <type_traits>

#include "demod_common.h"

#pragma clang fp contract(off)

#define WP_THREADS 512
#define WP_DSP_THREADS 320          // waves 3..7
#define WP_KP 2                     // raw samples prefetched per D thread (2*320 >= N+Ts/2 is required)
#define WP_DSP_WAVES 5
#define WP_CK 8                     // the chain wave stores every WP_CK-th phasor; D threads replay the steps in between
#define WP_SPIN_SLEEP 3                // s_sleep units (64 clk) between polls of a D-wave barrier: spinning waves steal issue slots
#define WP_CKROW 80                 // checkpoints per (segment, tone) row; needs >= (Nmem-Ts/P)/WP_CK + 2

namespace {

// barrier among the D waves only: monotone LDS counter, one arrival per wave per phase
__device__ __forceinline__ void dsp_barrier(int *cnt, int target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(WP_SPIN_SLEEP);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

enum { CT_NIN_NEXT = 0, CT_CNT = 1, CT_FBIN = 4 /* [4 frames][4 tones] */, CT_INTS = 24 };


// NCO chain of one frame with the real / imaginary part of tone m in lanes 2m / 2m+1 (nco_step_split): the batch form of
// C(j) below -- same statements, half the SIMD time per step, a longer dependent path.  Out of line so that the kernel's
// register allocation (80 VGPRs in the three-captures-per-CU variant) is not disturbed by it.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
__device__ __forceinline__ void nco_chain_split(int j, int nin_j, int lane, int M, int N, int NH, int Nmem, int L, lds_i32 *CT, lds_f32 *PHE,
                                             lds_f32 *CKb, lds_f32 *CKD, const lds_f32 *dphi_t, const float *bin_freq, const float2 *backoff_tab) {
    if (lane >= 2 * M) return;
    const int m = lane >> 1, part = lane & 1;
    const int nold = Nmem - nin_j;
    int bc = CT[CT_FBIN + (j & 3) * 4 + m];
    int bp = CT[CT_FBIN + ((j + 3) & 3) * 4 + m];
    const int bp0 = CT[CT_FBIN + ((j + 3) & 3) * 4 + 0];
    if (bin_freq[bp0] < 1.0f) bp = bc;                                   // first run (fsk.c:750-753)
    const int ncase = (nin_j < N) ? 0 : ((nin_j > N) ? 2 : 1);
    const float2 bo = backoff_tab[ncase * NH + bp];
    const lds_f32 *pc = PHE + (((j + 2) % 3) * 4 + m) * 2;
    const v2f phi0 = cmul_pk((v2f){bo.x, bo.y}, (v2f){pc[0], pc[1]});    // fsk.c:758-759 (both lanes of the pair)
    float own = part ? phi0.y : phi0.x;
    float dx = dphi_t[2 * bp], dy = dphi_t[2 * bp + 1];
    float k1 = dx, k2 = part ? dy : -dy;
    lds_f32 *ckA = CKb + ((((j & 1) * 2 + 0) * M + m) * WP_CKROW) * 2 + part;
    lds_f32 *ckB = CKb + ((((j & 1) * 2 + 1) * M + m) * WP_CKROW) * 2 + part;
    CKD[(((j & 1) * 2 + 0) * M + m) * 2 + part] = part ? dy : dx;
    int s = 0, c = 0;
    for (; s + WP_CK <= nold; s += WP_CK, c++) {
        ckA[2 * c] = own;
        static_assert(WP_CK == 8, "nco_step_split8"); own = nco_step_split8(own, k1, k2);
    }
    if (s < nold) { ckA[2 * c] = own; for (; s < nold; s++) own = nco_step_split(own, k1, k2); }
    {
        const float oth = __shfl_xor(own, 1, 64);
        const float re = part ? oth : own, im = part ? own : oth;
        const float av = sqrtf(re * re + im * im);                       // comp_normalize (fsk.c:787)
        own = own / av;
        dx = dphi_t[2 * bc]; dy = dphi_t[2 * bc + 1];
        k1 = dx; k2 = part ? dy : -dy;
    }
    CKD[(((j & 1) * 2 + 1) * M + m) * 2 + part] = part ? dy : dx;
    c = 0;
    for (; s + 4 * WP_CK <= L; s += 4 * WP_CK, c += 4) {                     // four checkpoints per trip: a taken branch costs ~16 cycles
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ckB[2 * (c + k)] = own;
            own = nco_step_split8(own, k1, k2);
        }
    }
    for (; s + WP_CK <= L; s += WP_CK, c++) {
        ckB[2 * c] = own;
        static_assert(WP_CK == 8, "nco_step_split8"); own = nco_step_split8(own, k1, k2);
    }
    if (s < L) { ckB[2 * c] = own; for (; s < L; s++) own = nco_step_split(own, k1, k2); }
    PHE[((j % 3) * 4 + m) * 2 + part] = own;                             // un-normalised (fsk.c:846)
}

}  // namespace

// RAW: every capture of the launch is cu8 -> the sample ring keeps the raw byte pairs (2 B instead of 8 B per
// sample; (u8-127)/128 is exact, so converting at each read gives the same floats) and the timing-product
// phasors stay in global memory.  That brings the workgroup under a third of a CU's LDS and, with the register
// bound below, lets THREE captures share a CU instead of two.
template <int M, bool PROF, bool RAW>
__global__ __launch_bounds__(WP_THREADS, RAW ? 6 : 5) void wenet_demod_pipe_kernel(WrDemodCfg cfg, const WrChan *chans, int nchan) {
    const int ch = blockIdx.x;
    if (ch >= nchan) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WrChan C = chans[ch];
    const int fmt_k = RAW ? (int)WR_FMT_CU8 : C.fmt;                  // the raw-ring variant only ever sees cu8: the format switches fold away

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *XR = (float2 *)(smem + cfg.p_off_XR);     // [ring]        sample ring, index (abs + nstash) & mask
    unsigned short *XRr = (unsigned short *)(smem + cfg.p_off_XR);      // the same ring as raw cu8 pairs (RAW)
    float2 *DCb = (float2 *)(smem + cfg.p_off_PH);    // [M][Lpad]     mixed samples -> timing products
    float2 *CKb = (float2 *)(smem + cfg.p_off_CK);    // [2 frames][2 segments][M][WP_CKROW] phasor checkpoints
    float2 *CKD = (float2 *)(smem + cfg.p_off_CKD);   // [2 frames][2 segments][M] NCO step of each segment
    float2 *FIb = (float2 *)(smem + cfg.p_off_FI);    // [2][M][NI]    integrator outputs of frame j in slot j&1
    float2 *TPb = (float2 *)(smem + cfg.p_off_TP);    // [2][NIq]      timing products of frame j in slot j&1: (re, im) pairs, or -- for the
                                                      //               lane-split timing sum of batch launches -- a row of re and a row of im
    const int NIq = (cfg.NI + 3) & ~3;                // products per row (16-byte rows), pairs per slot
    float2 *FB = (float2 *)(smem + cfg.p_off_FB);     // [Ndft]
    float  *FEr = (float *)(smem + cfg.p_off_FE);     // [4][Ndft/2]   smoothed spectrum after frame j in slot j&3
    float  *FW = (float *)(smem + cfg.p_off_FW);      // [Ndft/2]
    float  *SDL = (float *)(smem + cfg.p_off_SD);     // [Nbits]
    float  *SC = (float *)(smem + cfg.p_off_SC);      // scratch (Eb/N0)
    float2 *PHE = (float2 *)(smem + cfg.p_off_PHE);   // [3][4]        NCO phase at the end of frame j in slot j%3
    int    *CT = (int *)(smem + cfg.p_off_CT);        // control words
    const float2 *tw_t = (const float2 *)(smem + cfg.p_off_TW);
    const float  *hann_t = (const float *)(smem + cfg.p_off_HANN);
    const int    *src_t = (const int *)(smem + cfg.p_off_SRC);
    const float2 *pft_t = (const float2 *)(smem + cfg.p_off_PFT);
    const float2 *dphi_t = (const float2 *)(smem + cfg.p_off_DPHI);

    const int Ts = cfg.Ts, N = cfg.N, P = cfg.P, Nmem = cfg.Nmem, nstash = cfg.nstash;
    const int Ndft = cfg.Ndft, NH = cfg.Ndft / 2, L = cfg.L, NI = cfg.NI, q = cfg.q, Lpad = cfg.Lpad;
    const int Nbits = cfg.Nbits, Nmax = N + Ts / 2;
    const int rmask = cfg.p_ring - 1;
#define RIDX(a) ((int)(((a) + nstash) & rmask))
    auto ring_get = [&](int ri) -> float2 {
        if (RAW) { const unsigned w = XRr[ri]; return make_float2(((float)(w & 0xffu) - 127.0f) / 128.0f, ((float)(w >> 8) - 127.0f) / 128.0f); }
        return XR[ri];
    };
    auto ring_put_raw = [&](int ri, uint2 r, int fmt) {               // r as returned by load_raw
        if (RAW) XRr[ri] = (unsigned short)r.x; else XR[ri] = convert_raw(r, fmt);
    };
    auto ring_put_f = [&](int ri, float2 v) {                          // carried samples: exact inverse of the cu8 conversion
        if (RAW) XRr[ri] = (unsigned short)((unsigned)(int)(v.x * 128.0f + 127.0f) | ((unsigned)(int)(v.y * 128.0f + 127.0f) << 8));
        else XR[ri] = v;
    };

    // ---- carried state -> LDS ------------------------------------------------------------------
    WrChanHdr *hdr = (WrChanHdr *)C.state;
    float *st_fft = C.state + cfg.st_fft_est;
    float2 *st_old = (float2 *)(C.state + cfg.st_samp_old);
    float *st_sd = C.state + cfg.st_sd_last;
    {
        float2 *tw_w = (float2 *)(smem + cfg.p_off_TW); float *hann_w = (float *)(smem + cfg.p_off_HANN);
        int *src_w = (int *)(smem + cfg.p_off_SRC); float2 *pft_w = (float2 *)(smem + cfg.p_off_PFT);
        float2 *dphi_w = (float2 *)(smem + cfg.p_off_DPHI);
        for (int i = tid; i < Ndft; i += WP_THREADS) { tw_w[i] = cfg.tw[i]; hann_w[i] = cfg.hann[i]; src_w[i] = cfg.fft_src[i]; }
        if (!RAW) for (int i = tid; i < NI; i += WP_THREADS) pft_w[i] = cfg.phi_ft[i];
        for (int i = tid; i < NH; i += WP_THREADS) dphi_w[i] = cfg.dphi_tab[i];
    }
    for (int i = tid; i < NH; i += WP_THREADS) FEr[3 * NH + i] = st_fft[i];           // "after frame -1" lives in slot 3
    for (int i = tid; i < Nbits; i += WP_THREADS) SDL[i] = st_sd[i];
    for (int i = tid; i < nstash; i += WP_THREADS) ring_put_f(RIDX((long long)(i - nstash)), st_old[i]);
    if (tid < M) { PHE[2 * 4 + tid] = hdr->phi_c[tid]; CT[CT_FBIN + 3 * 4 + tid] = hdr->f_bin[tid]; }   // frame -1 -> slots 2 / 3
    if (tid == 0) { CT[CT_CNT] = 0; CT[CT_NIN_NEXT] = hdr->nin; }
    int nin = __builtin_amdgcn_readfirstlane(hdr->nin);
    // first 4*Nmax samples into the ring
    {
        const long long last = C.nsamples - 1;
        for (long long i = tid; i < 4LL * Nmax; i += WP_THREADS)
            if (C.nsamples > 0) ring_put_raw(RIDX(i), load_raw(C.raw, fmt_k, i < last ? i : last), fmt_k);
            else ring_put_f(RIDX(i), make_float2(0.f, 0.f));
    }
    long long filled = 4LL * Nmax;                    // ring holds absolute samples [off - nstash, filled)
    lds_barrier();

    // ================================ stage bodies ============================================
    // E(j): tone estimator of frame j.  One wavefront.  slot_in/out index the spectrum ring.
    auto estimate = [&](int j, long long off_j, int nin_j) {
        const float *FEin = FEr + ((j + 3) & 3) * NH;          // after frame j-1
        float *FEout = FEr + (j & 3) * NH;
        const int fft_loops = nin_j / Ndft;
        for (int jl = 0; jl < fft_loops; jl++) {
            const int samps = nin_j - (jl + 1) * Ndft;                  // fsk.c:583
            const int fft_samps = samps >= Ndft ? Ndft : samps;         // fsk.c:584
            for (int n = lane; n < Ndft; n += 64) {
                const int idx = src_t[n];
                float2 v = make_float2(0.f, 0.f);
                if (idx < fft_samps) {
                    const float h = hann_t[idx];
                    const float2 x = ring_get(RIDX(off_j + idx + Ndft * jl));
                    v = make_float2(h * x.x, h * x.y);
                }
                FB[n] = v;
            }
            wave_sync();
            for (int s = cfg.nstages - 1; s >= 0; s--) {
                const int m = cfg.mstage[s], p = cfg.radix[s], fs = cfg.fstride[s];
                const int lgm = 31 - __clz(m);
                const int nb = Ndft / p;
                for (int b = lane; b < nb; b += 64) {
                    const int blk = b >> lgm, k = b & (m - 1);           // m is a power of two (Ndft is)
                    float2 *F = FB + blk * m * p + k;
                    if (p == 4) {                                       // kf_bfly4 (kiss_fft.c:44-90)
                        const float2 s0 = cmul(F[m], tw_t[k * fs]);
                        const float2 s1 = cmul(F[2 * m], tw_t[k * fs * 2]);
                        const float2 s2 = cmul(F[3 * m], tw_t[k * fs * 3]);
                        float2 f0 = F[0];
                        const float2 s5 = make_float2(f0.x - s1.x, f0.y - s1.y);
                        f0 = make_float2(f0.x + s1.x, f0.y + s1.y);
                        const float2 s3 = make_float2(s0.x + s2.x, s0.y + s2.y);
                        const float2 s4 = make_float2(s0.x - s2.x, s0.y - s2.y);
                        F[2 * m] = make_float2(f0.x - s3.x, f0.y - s3.y);
                        F[0] = make_float2(f0.x + s3.x, f0.y + s3.y);
                        F[m] = make_float2(s5.x + s4.y, s5.y - s4.x);
                        F[3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
                    } else {                                            // kf_bfly2 (kiss_fft.c:21-42)
                        const float2 t = cmul(F[m], tw_t[k * fs]);
                        const float2 f0 = F[0];
                        F[m] = make_float2(f0.x - t.x, f0.y - t.y);
                        F[0] = make_float2(f0.x + t.x, f0.y + t.y);
                    }
                }
                wave_sync();
            }
            const float *FEcur = (jl == 0) ? FEin : FEout;
            for (int i = lane; i < NH; i += 64) {                       // fsk.c:612-628
                const float2 v = FB[i];
                float mag = (v.x * v.x) + (v.y * v.y);
                if (i < cfg.f_min) mag = 0.f;
                if (cfg.f_max - 1 >= 0 && i >= cfg.f_max - 1) mag = 0.f;
                const float e = (FEcur[i] * cfg.one_minus_tc) + (sqrtf(mag) * cfg.tc);
                FEout[i] = e;
                FW[i] = e;
            }
            wave_sync();
        }
        if (fft_loops == 0) {
            for (int i = lane; i < NH; i += 64) { FEout[i] = FEin[i]; FW[i] = 0.f; }
            wave_sync();
        }
        int fbin[M];
#pragma unroll
        for (int k = 0; k < M; k++) {                                   // fsk.c:633-654
            BestBin best; best.v = 0.f; best.i = 0;
            for (int jj = lane; jj < NH; jj += 64) {
                const float v = FW[jj];
                if (v > best.v) { best.v = v; best.i = jj; }
            }
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) {
                BestBin o;
                o.v = __shfl_xor(best.v, sh, 64);
                o.i = __shfl_xor(best.i, sh, 64);
                best = better(best, o);
            }
            const int imax = __builtin_amdgcn_readfirstlane((best.v > 0.f) ? best.i : 0);
            int lo = imax - cfg.f_zero; lo = lo < 0 ? 0 : lo;
            int hi = imax + cfg.f_zero; hi = hi > NH ? NH : hi;
            wave_sync();
            for (int jj = lo + lane; jj < hi; jj += 64) FW[jj] = 0.f;
            wave_sync();
            fbin[k] = imax;
        }
#pragma unroll
        for (int a = 1; a < M; a++) {
#pragma unroll
            for (int b = a; b > 0; b--)
                if (fbin[b - 1] > fbin[b]) { const int t = fbin[b]; fbin[b] = fbin[b - 1]; fbin[b - 1] = t; }
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) CT[CT_FBIN + (j & 3) * 4 + m] = fbin[m];
        }
        wave_sync();
    };

    // C(j): NCO phasor chain of frame j (one wavefront, lanes 0..M-1 carry one tone each)
    auto chain = [&](int j, int nin_j) {
        if constexpr (RAW) {                                             // the raw-ring variant only runs batches: lane-split form (less SIMD time)
            nco_chain_split(j, nin_j, lane, M, N, NH, Nmem, L, (lds_i32 *)CT, (lds_f32 *)PHE, (lds_f32 *)CKb, (lds_f32 *)CKD,
                            (const lds_f32 *)dphi_t, cfg.bin_freq, cfg.backoff_tab);
            wave_sync();
            return;
        } else {
        if (lane < M) {
            const int nold = Nmem - nin_j;
            int bc = CT[CT_FBIN + (j & 3) * 4 + lane];
            int bp = CT[CT_FBIN + ((j + 3) & 3) * 4 + lane];
            const int bp0 = CT[CT_FBIN + ((j + 3) & 3) * 4 + 0];
            if (cfg.bin_freq[bp0] < 1.0f) bp = bc;                       // first run (fsk.c:750-753)
            const int ncase = (nin_j < N) ? 0 : ((nin_j > N) ? 2 : 1);
            const float2 bo = cfg.backoff_tab[ncase * NH + bp];
            const float2 pc = PHE[((j + 2) % 3) * 4 + lane];
            v2f phi = cmul_pk((v2f){bo.x, bo.y}, (v2f){pc.x, pc.y});     // fsk.c:758-759
            float2 dd = dphi_t[bp];
            v2f d = {dd.x, dd.y};
            // segment A: nold steps on the old samples with the previous estimate; segment B: L-nold steps on
            // the new block after comp_normalize with the new estimate.  Only every WP_CK-th phasor is stored.
            v2f *ckA = (v2f *)(CKb + (((j & 1) * 2 + 0) * M + lane) * WP_CKROW);
            v2f *ckB = (v2f *)(CKb + (((j & 1) * 2 + 1) * M + lane) * WP_CKROW);
            CKD[((j & 1) * 2 + 0) * M + lane] = dd;
            int s = 0, c = 0;
            for (; s + WP_CK <= nold; s += WP_CK, c++) {
                ckA[c] = phi;
#pragma unroll
                for (int u = 0; u < WP_CK; u++) phi = cmul_pk(phi, d);
            }
            if (s < nold) { ckA[c] = phi; for (; s < nold; s++) phi = cmul_pk(phi, d); }
            {
                const float av = sqrtf(phi.x * phi.x + phi.y * phi.y);   // comp_normalize (fsk.c:787)
                phi = (v2f){phi.x / av, phi.y / av};
                dd = dphi_t[bc];
                d = (v2f){dd.x, dd.y};
            }
            CKD[((j & 1) * 2 + 1) * M + lane] = dd;
            c = 0;
            for (; s + 4 * WP_CK <= L; s += 4 * WP_CK, c += 4) {            // four checkpoints per trip: a taken branch costs ~16 cycles
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ckB[c + k] = phi;
#pragma unroll
                    for (int u = 0; u < WP_CK; u++) phi = cmul_pk(phi, d);
                }
            }
            for (; s + WP_CK <= L; s += WP_CK, c++) {
                ckB[c] = phi;
#pragma unroll
                for (int u = 0; u < WP_CK; u++) phi = cmul_pk(phi, d);
            }
            if (s < L) { ckB[c] = phi; for (; s < L; s++) phi = cmul_pk(phi, d); }
            PHE[(j % 3) * 4 + lane] = make_float2(phi.x, phi.y);        // un-normalised (fsk.c:846)
        }
        wave_sync();
        }
    };

    // D(j): mix + integrate + timing products of frame j.  Waves 3..7 (t = D thread index).
    const int t = tid - 192;
    int dsp_phase = 0;
    auto dstage = [&](int j, long long off_j, int nin_j) {
        const int nold = Nmem - nin_j;
        float2 *PH = DCb;
        float2 *FI = FIb + (j & 1) * M * NI;
        float2 *TP = TPb + (j & 1) * NIq;
        float *TPs = (float *)TP;                                        // split layout: TPs[i] = re, TPs[NIq + i] = im
        const long long src0 = off_j - nold;                             // chain step s <-> absolute sample src0 + s
        const bool fastI = (q == 1 && (Ts == 10 || Ts == 8) && NI % Ts == 0);                // fast integrator path
        const int padTs = fastI ? Ts : 0;                                          // ... with padded rows: one element after every Ts samples, so that the
                                                                                   // integrator's lane stride is Ts+1 elements (odd: all LDS banks) instead of Ts
        auto mix = [&](auto PADC) {
            constexpr int PADTS = decltype(PADC)::value;             // 0: rows unpadded; 8: one pad element after every 8 samples
            // one D thread per (tone, checkpoint): replay the <= WP_CK chain steps that follow the checkpoint
            // (the same cmul_pk sequence the chain wave ran) and mix each sample with its conjugate phasor
            const int nA = (nold + WP_CK - 1) / WP_CK, nB = (L - nold + WP_CK - 1) / WP_CK;
            const int per_tone = nA + nB;
            if (RAW && M == 2) {
                // batch variant: one D thread per checkpoint does ALL tones of its samples -- each raw sample is read from the ring
                // and converted once instead of once per tone (fewer instructions; one stream prefers the finer split below)
                for (int c = t; c < per_tone; c += WP_DSP_THREADS) {
                    const bool segB = c >= nA;
                    const int cc = segB ? c - nA : c;
                    const int s0 = segB ? nold + cc * WP_CK : cc * WP_CK;
                    const int send = segB ? L : nold;
                    const int cnt = (send - s0) < WP_CK ? (send - s0) : WP_CK;
                    v2f d[M], phi[M];
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        const float2 dd = CKD[((j & 1) * 2 + (segB ? 1 : 0)) * M + m];
                        d[m] = (v2f){dd.x, dd.y};
                        phi[m] = ((const v2f *)(CKb + (((j & 1) * 2 + (segB ? 1 : 0)) * M + m) * WP_CKROW))[cc];
                    }
                    const int pq0 = PADTS ? s0 / (PADTS ? PADTS : 1) : 0, pr0 = s0 - pq0 * PADTS;
                    float2 *row = PH + s0 + pq0;
                    float2 *dump = PH + M * Lpad;                        // steps past the end of a segment land here (no branches)
                    const int rbase = RIDX(src0 + s0);
                    float2 x[WP_CK];
#pragma unroll
                    for (int u = 0; u < WP_CK; u++) x[u] = ring_get((rbase + (u < cnt ? u : 0)) & rmask);
#pragma unroll
                    for (int u = 0; u < WP_CK; u++) {
                        float2 *dst = (u < cnt) ? row + u + ((PADTS && pr0 + u >= PADTS) ? 1 : 0) : dump;  // (WP_CK <= Ts: at most one pad crossed)
#pragma unroll
                        for (int m = 0; m < M; m++) {
                            dst[(u < cnt) ? m * Lpad : 0] = cmul(x[u], make_float2(phi[m].x, -phi[m].y));
                            phi[m] = cmul_pk(phi[m], d[m]);
                        }
                    }
                }
                return;
            }
            for (int w = t; w < M * per_tone; w += WP_DSP_THREADS) {
                const int m = w / per_tone, c = w - m * per_tone;
                const bool segB = c >= nA;
                const int cc = segB ? c - nA : c;
                const int s0 = segB ? nold + cc * WP_CK : cc * WP_CK;
                const int send = segB ? L : nold;
                const int cnt = (send - s0) < WP_CK ? (send - s0) : WP_CK;
                const float2 dd = CKD[((j & 1) * 2 + (segB ? 1 : 0)) * M + m];
                const v2f d = {dd.x, dd.y};
                v2f phi = ((const v2f *)(CKb + (((j & 1) * 2 + (segB ? 1 : 0)) * M + m) * WP_CKROW))[cc];
                // In the fast integrator layout one element of padding follows every Ts samples (sample s sits at s + s/Ts):
                // lane strides of 8 and Ts elements would hit only 2..16 of the 32 LDS banks, 9 and Ts+1 hit all of them.
                const int pq0 = PADTS ? s0 / (PADTS ? PADTS : 1) : 0, pr0 = s0 - pq0 * PADTS;
                float2 *row = PH + m * Lpad + s0 + pq0;
                float2 *dump = PH + M * Lpad;                            // steps past the end of a segment land here (no branches)
                const int rbase = RIDX(src0 + s0);
                float2 x[WP_CK];
#pragma unroll
                for (int u = 0; u < WP_CK; u++) x[u] = ring_get((rbase + (u < cnt ? u : 0)) & rmask);
#pragma unroll
                for (int u = 0; u < WP_CK; u++) {
                    float2 *dst = (u < cnt) ? row + u + ((PADTS && pr0 + u >= PADTS) ? 1 : 0) : dump;      // (WP_CK <= Ts: at most one pad crossed)
                    *dst = cmul(x[u], make_float2(phi.x, -phi.y));
                    phi = cmul_pk(phi, d);
                }
            }
                };
#ifdef WR_DBG_SKIP
        if (!(cfg.dbg_skip & 16)) {
#endif
        if (padTs == 8) mix(std::integral_constant<int, 8>()); else if (padTs == 10) mix(std::integral_constant<int, 10>()); else mix(std::integral_constant<int, 0>());
#ifdef WR_DBG_SKIP
        }
#endif
        dsp_barrier(&CT[CT_CNT], WP_DSP_WAVES * (++dsp_phase), lane);
        if (fastI) {
            // Fast path (one sample per integrator step).  The Ts circular-buffer slots are summed in SLOT order
            // (fsk.c:829-840), i.e. the window row[i .. i+Ts) rotated by o = (-i) mod Ts.  Each D wave takes whole
            // residue classes i mod Ts, so o is wave-uniform and the rotation is resolved at compile time (no index
            // arithmetic per element); one lane does both tones of its output and the timing product right away,
            // which also saves the barrier between the two steps.
            auto residue = [&](int r, auto TSC, auto OC) {
                constexpr int TS = decltype(TSC)::value, O = decltype(OC)::value;
                for (int j = lane; j < NI / TS; j += 64) {
                    const int i = j * TS + r;
                    float ft1 = 0.f;
#pragma unroll 1
                    for (int m = 0; m < M; m++) {
                        constexpr int R = (TS - O) % TS;                    // this residue class (wave-uniform, == r)
                        constexpr bool PAD = true;                          // rows padded by one element per TS samples (see the mix stage)
                        const v2f *row = (const v2f *)PH + m * Lpad + (PAD ? j * (TS + 1) + R : i);
                        v2f v[TS];
#pragma unroll
                        for (int u = 0; u < TS; u++) v[u] = row[u + ((PAD && R + u >= TS) ? 1 : 0)];
                        v2f acc = {0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < TS; u++) acc = acc + v[(O + u) % TS];
                        FI[m * NI + i] = make_float2(acc.x, acc.y);
                        ft1 += (acc.x * acc.x) + (acc.y * acc.y);           // fsk.c:862-868
                    }
                    const float2 pf = RAW ? cfg.phi_ft[i] : pft_t[i];
                    if (cfg.p_tsum_split) { TPs[i] = ft1 * pf.x; TPs[NIq + i] = ft1 * pf.y; }
                    else TP[i] = make_float2(ft1 * pf.x, ft1 * pf.y);
                }
            };
            auto classes = [&](auto TSC) {
                constexpr int TS = decltype(TSC)::value;
                for (int r = wave - 3; r < TS; r += WP_DSP_WAVES) {
                    switch ((r == 0) ? 0 : TS - r) {
#define WP_ROT(K) case K: residue(r, TSC, std::integral_constant<int, (K) % TS>()); break;
                        WP_ROT(0) WP_ROT(1) WP_ROT(2) WP_ROT(3) WP_ROT(4) WP_ROT(5) WP_ROT(6) WP_ROT(7) WP_ROT(8) WP_ROT(9)
#undef WP_ROT
                    }
                }
            };
#ifdef WR_DBG_SKIP
            if (!(cfg.dbg_skip & 32))
#endif
            if (Ts == 10) classes(std::integral_constant<int, 10>()); else classes(std::integral_constant<int, 8>());
            wave_sync();
            return;
        }
        {
            // one row per (tone, output): sum the Ts circular-buffer slots in slot order (fsk.c:829-840), loads first
            auto integrate_row = [&](int m, int i, auto TSC) {
                constexpr int TS = decltype(TSC)::value;                 // 0 = runtime Ts
                const int ts = TS ? TS : Ts;
                const int base = i * q;
                const int r = base % ts;
                int o = (r == 0) ? 0 : ts - r;
                const v2f *row = (const v2f *)PH + m * Lpad + base;
                v2f acc = {0.f, 0.f};
                if (TS) {
                    v2f v[TS ? TS : 1];
#pragma unroll
                    for (int u = 0; u < TS; u++) { v[u] = row[o]; o++; if (o == ts) o = 0; }
#pragma unroll
                    for (int u = 0; u < TS; u++) acc = acc + v[u];
                } else {
                    for (int j0 = 0; j0 < ts; j0 += 8) {
                        v2f v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { v[u] = row[(j0 + u < ts) ? o : 0]; if (j0 + u < ts) { o++; if (o == ts) o = 0; } }
#pragma unroll
                        for (int u = 0; u < 8; u++) if (j0 + u < ts) acc = acc + v[u];
                    }
                }
                FI[m * NI + i] = make_float2(acc.x, acc.y);
            };
            for (int w = t; w < M * NI; w += WP_DSP_THREADS) {
                const int m = w / NI, i = w - m * NI;
                if (Ts == 10) integrate_row(m, i, std::integral_constant<int, 10>());
                else if (Ts == 8) integrate_row(m, i, std::integral_constant<int, 8>());
                else integrate_row(m, i, std::integral_constant<int, 0>());
            }
        }
        dsp_barrier(&CT[CT_CNT], WP_DSP_WAVES * (++dsp_phase), lane);
        for (int i = t; i < NI; i += WP_DSP_THREADS) {                   // timing products (fsk.c:862-870)
            float ft1 = 0.f;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const float2 v = FI[m * NI + i];
                ft1 += (v.x * v.x) + (v.y * v.y);
            }
            const float2 pf = RAW ? cfg.phi_ft[i] : pft_t[i];
            if (cfg.p_tsum_split) { TPs[i] = ft1 * pf.x; TPs[NIq + i] = ft1 * pf.y; }
            else TP[i] = make_float2(ft1 * pf.x, ft1 * pf.y);
        }
        wave_sync();
    };

    // T(k): ordered sum, timing, nin, decisions (fsk.c:870-993).  Wave 2.  kf = frame index in this launch.
    float norm_rx_timing_st = hdr->norm_rx_timing;                       // T-wave private carried scalars
    float ppm = hdr->ppm;
    auto tstage = [&](int kf, long long frames, int nin_cur) {
        const float2 *FI = FIb + (kf & 1) * M * NI;
        const float2 *TP = TPb + (kf & 1) * NIq;
        float tcr, tci;
        {
            if (cfg.p_tsum_split) {
                // Real part in even lanes, imaginary part in odd lanes: 490 dependent PLAIN adds per frame instead of packed
                // ones (a packed-f32 op occupies the SIMD twice as long, and this wave shares its SIMD with other captures).
                typedef float v4f __attribute__((ext_vector_type(4)));
                const float *TPf = (const float *)TP + (lane & 1) * NIq;    // this lane's row (re or im), four products per 128-bit LDS read
                const v4f *T4 = (const v4f *)TPf;
                float acc = 0.f;
                v4f bufA[4], bufB[4];                                        // ping-pong in batches of 16 products: under load an LDS read takes
                int i = 0;                                                   // longer than eight dependent adds, so the next batch is asked for 16 ahead
#define WP_ADD16(buf) do { _Pragma("unroll") for (int u = 0; u < 4; u++) { acc = acc + buf[u].x; acc = acc + buf[u].y; acc = acc + buf[u].z; acc = acc + buf[u].w; } } while (0)
#define WP_LD16(buf, at) do { _Pragma("unroll") for (int u = 0; u < 4; u++) buf[u] = T4[((at) >> 2) + u]; } while (0)
                if (NI >= 16) {
                    WP_LD16(bufA, 0);
                    for (i = 16; i + 32 <= NI; i += 32) {
                        WP_LD16(bufB, i);
                        WP_ADD16(bufA);
                        asm volatile("" : "+v"(acc) : : "memory");           // keep the reload of A behind its last use
                        WP_LD16(bufA, i + 16);
                        WP_ADD16(bufB);
                        asm volatile("" : "+v"(acc) : : "memory");
                    }
                    if (i + 16 <= NI) {
                        WP_LD16(bufB, i);
                        WP_ADD16(bufA);
                        WP_ADD16(bufB);
                        i += 16;
                    } else {
                        WP_ADD16(bufA);
                    }
                }
#undef WP_ADD16
#undef WP_LD16
                for (; i < NI; i++) acc = acc + TPf[i];
                tcr = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(acc), 0));
                tci = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(acc), 1));
            } else {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f *TP4 = (const v4f *)TP;
                v2f acc = {0.f, 0.f};
                v4f bufA[4], bufB[4];                                        // ping-pong: loads of one batch fly while the other is summed
                int i = 0;
                if (NI >= 8) {
#pragma unroll
                    for (int u = 0; u < 4; u++) bufA[u] = TP4[u];
                    for (i = 8; i + 16 <= NI; i += 16) {
#pragma unroll
                        for (int u = 0; u < 4; u++) bufB[u] = TP4[(i >> 1) + u];
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc = acc + bufA[u].xy; acc = acc + bufA[u].zw; }
                        asm volatile("" : "+v"(acc) : : "memory");           // keep the reload of A behind its last use (else the
#pragma unroll                                                           // scheduler hoists it and pays 8 register copies per round)
                        for (int u = 0; u < 4; u++) bufA[u] = TP4[(i >> 1) + 4 + u];
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc = acc + bufB[u].xy; acc = acc + bufB[u].zw; }
                        asm volatile("" : "+v"(acc) : : "memory");
                    }
                    if (i + 8 <= NI) {
#pragma unroll
                        for (int u = 0; u < 4; u++) bufB[u] = TP4[(i >> 1) + u];
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc = acc + bufA[u].xy; acc = acc + bufA[u].zw; }
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc = acc + bufB[u].xy; acc = acc + bufB[u].zw; }
                        i += 8;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc = acc + bufA[u].xy; acc = acc + bufA[u].zw; }
                    }
                }
                for (; i < NI; i++) { const float2 v = TP[i]; acc = acc + (v2f){v.x, v.y}; }
                tcr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(acc.x)));
                tci = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(acc.y)));
            }
        }
        int nin_next = nin_cur;
        float tr_mean = 0.f, tr_std = 0.f, tr_rxt = 0.f;
        const bool nan_frame = (tcr != tcr) || (tci != tci);             // fsk.c:878-880
        if (!nan_frame) {
            const float at = wg_atan2f(tci, tcr);
            const float norm_rx_timing = (float)((double)at / (2 * 3.14159265358979323846));
            const float rx_timing = norm_rx_timing * cfg.P_f;
            const float d_nrt = norm_rx_timing - norm_rx_timing_st;
            norm_rx_timing_st = norm_rx_timing;
            if ((double)fabsf(d_nrt) < .2) {
                const float appm = (float)(1e6 * (double)d_nrt / (double)cfg.nsym_f);
                ppm = (float)(.9 * (double)ppm + .1 * (double)appm);
            }
            if (norm_rx_timing > 0.25f) nin_next = N + Ts / 2;
            else if (norm_rx_timing < -0.25f) nin_next = N - Ts / 2;
            else nin_next = N;
            nin_next = __builtin_amdgcn_readfirstlane(nin_next);
            if (lane == 0) CT[CT_NIN_NEXT] = nin_next;                   // published early; read after the frame barrier
            const int low_sample = (int)floorf(rx_timing);
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = (int)ceilf(rx_timing);
            const float omf = 1 - fract;
            tr_rxt = rx_timing;
            float mymax = 0.f;
            if (lane < WR_NSYM) {
                const int st = (lane + 1) * P;
                float tmax[M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const float2 a = FI[m * NI + st + low_sample];
                    const float2 b = FI[m * NI + st + high_sample];
                    float tr = omf * a.x, ti = omf * a.y;
                    tr = tr + fract * b.x;
                    ti = ti + fract * b.y;
                    tmax[m] = (tr * tr) + (ti * ti);
                }
                float mx = tmax[0];
                int sym = 0;
#pragma unroll
                for (int m = 0; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
                mymax = mx;
                if (C.bits_out) {
                    uint8_t *bo = C.bits_out + frames * Nbits;
                    if (M == 2) bo[lane] = (uint8_t)(sym == 1);
                    else { bo[lane * 2 + 1] = (uint8_t)(sym & 1); bo[lane * 2] = (uint8_t)((sym & 2) >> 1); }
                }
#pragma unroll
                for (int m = 0; m < M; m++) tmax[m] = sqrtf(tmax[m]);
                if (M == 2) {
                    SDL[lane] = tmax[0] - tmax[1];
                } else {
                    float s1 = -tmax[0], s0 = -tmax[0];
                    s1 += tmax[1 % M];  s0 += -tmax[1 % M];
                    s1 += -tmax[2 % M]; s0 += tmax[2 % M];
                    s1 += tmax[3 % M];  s0 += tmax[3 % M];
                    SDL[lane * 2 + 1] = s1;
                    SDL[lane * 2] = s0;
                }
            }
            if (cfg.stats) {
                if (lane < WR_NSYM) ((float2 *)SC)[lane] = make_float2(mymax, sqrtf(mymax));
                wave_sync();
                if (lane == 0) {
                    // the two running sums of fsk.c:998-1004 are independent chains: one packed add per symbol, loads up front
                    v2f acc = {0.f, 0.f};
                    const v2f *sc2 = (const v2f *)SC;
#pragma unroll
                    for (int i0 = 0; i0 < WR_NSYM; i0 += 16) {
                        v2f w[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) w[u] = sc2[i0 + u];
#pragma unroll
                        for (int u = 0; u < 16; u++) acc = acc + w[u];
                    }
                    float stdebno = acc.x, meanebno = acc.y;
                    meanebno = meanebno / cfg.nsym_f;
                    stdebno = (stdebno / cfg.nsym_f) - (meanebno * meanebno);
                    if ((double)stdebno > 0.0) stdebno = (float)sqrt((double)stdebno); else stdebno = 0.0f;
                    SC[2 * WR_NSYM] = meanebno;
                    SC[2 * WR_NSYM + 1] = stdebno;
                }
                wave_sync();
                tr_mean = SC[2 * WR_NSYM];
                tr_std = SC[2 * WR_NSYM + 1];
            }
            if (C.dump && frames >= C.dump_first && ((frames - C.dump_first) % C.dump_period) == 0) {
                const long long slot = (frames - C.dump_first) / C.dump_period;
                if (slot < C.dump_cap) {
                    float *d = C.dump + slot * cfg.dump_floats;
                    const float *FEk = FEr + (kf & 3) * NH;
                    const int neye = cfg.eye_traces * M * cfg.neyesamp;
                    for (int e = lane; e < neye; e += 64) {
                        const int j = e % cfg.neyesamp;
                        const int tm = e / cfg.neyesamp;
                        const int i = tm / M, m = tm - i * M;
                        const int ind = 2 * P * i + (high_sample + 1) + j * cfg.eye_dec;
                        float v = 0.f;
                        if (ind >= 0 && ind < NI) { const float2 f = FI[m * NI + ind]; v = sqrtf(f.x * f.x + f.y * f.y); }
                        d[e] = v;
                    }
                    for (int i = lane; i < NH; i += 64) d[neye + i] = FEk[i];
                    if (lane == 0) { d[neye + NH] = (float)high_sample; d[neye + NH + 1] = (float)frames; }
                }
            }
        } else if (lane == 0) {
            CT[CT_NIN_NEXT] = nin_next;
        }
        wave_sync();
        if (C.sd_out) {
            float *so = C.sd_out + frames * Nbits;
            for (int i = lane; i < Nbits; i += 64) so[i] = SDL[i];
        }
        if (C.trace && lane == 0) {
            float *tr = C.trace + frames * WR_TRACE_FLOATS;
#pragma unroll
            for (int m = 0; m < WR_M_MAX; m++) tr[WR_TR_FEST + m] = (m < M) ? cfg.bin_freq[CT[CT_FBIN + (kf & 3) * 4 + (m < M ? m : 0)]] : 0.f;
            tr[WR_TR_NIN] = (float)nin_next;
            tr[WR_TR_NRT] = norm_rx_timing_st;
            tr[WR_TR_PPM] = ppm;
            tr[WR_TR_MEAN] = nan_frame ? __int_as_float(0x7fc00000) : tr_mean;   // NaN marks a frame the reference returned early from (fsk.c:878-880): the host leaves EbNodB / snr_est alone
            tr[WR_TR_STD] = tr_std;
            tr[WR_TR_RXT] = tr_rxt;
        }
    };

    // the chain wave is the critical path of every frame: let it win VALU arbitration on its SIMD
    {   // cfg.chain_prio = chain | T << 2 | estimator << 4 (two bits each; s_setprio takes an immediate)
        const int pr = (wave == 0) ? (cfg.chain_prio & 3) : (wave == 2) ? ((cfg.chain_prio >> 2) & 3) : (wave == 1) ? ((cfg.chain_prio >> 4) & 3) : 0;
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    }

    // ================================ pipeline prologue ========================================
    //   E(0) | C(0),E(1) | D(0),C(1),E(2)         (frame 0 with the true nin, later frames speculative)
    long long off = 0, frames = 0;
    const bool any = (off + nin <= C.nsamples) && (C.cap_frames > 0);
    if (any) {
        if (wave == 1) estimate(0, 0, nin);
        lds_barrier();
        if (wave == 0) chain(0, nin);
        if (wave == 1) estimate(1, (long long)nin, N);
        lds_barrier();
        if (wave == 0) chain(1, N);
        if (wave == 1) estimate(2, (long long)nin + N, N);
        if (wave >= 3) dstage(0, 0, nin);
        lds_barrier();
    }
    // D-thread prefetch registers: samples [filled, filled + 2*320)
    uint2 pre[WP_KP];
#pragma unroll
    for (int k = 0; k < WP_KP; k++) pre[k] = make_uint2(0u, 0u);
    if (any && wave >= 3) {
        const long long last = C.nsamples - 1;
#pragma unroll
        for (int k = 0; k < WP_KP; k++) { long long i = filled + t + WP_DSP_THREADS * k; pre[k] = load_raw(C.raw, fmt_k, i < last ? i : last); }
    }

    // ================================ frame loop ===============================================
    // Batch variant (RAW): one copy of the loop per role -- a wavefront never leaves its role, so inside its copy only that
    // role's values are live (no VGPR spills under the 80-register cap, a third fewer SGPR reloads per frame: -4 % at three
    // captures per CU).  One stream is 8 % faster with the single loop for all roles (the timing wave is its critical path
    // and comes out 1.2 k cycles per frame slower in the split form), so the float-ring variant keeps that.
    int kf = 0;                                                          // frame index within this launch
    int nslip = 0;                                                       // frames with nin(k+1) != N (reported to the host: batch kernel choice)
    long long pr_busy = 0, pr_iter = 0, pr_redo = 0, pr_t0 = 0;          // PROF: per-role busy ticks
#ifdef WR_DBG_SKIP                                                       // development build only (tools/gpu_stage_cost.sh): leave stages out
    const int skip = cfg.dbg_skip;                                       // 1 chain, 2 estimator, 4 D, 8 T, 16 mix, 32 integrate, 64 staging -- results are garbage
#else
    constexpr int skip = 0;
#endif
    if constexpr (RAW) {
    // work(off1): this role's stage of the steady pipeline;  redo1/2/3(off1, nin_next): its part of the three re-run steps
    auto frame_loop = [&](auto work, auto redo1, auto redo2, auto redo3, bool is_d) {
        while (off + nin <= C.nsamples && frames < C.cap_frames) {
            if (PROF) pr_t0 = (long long)__builtin_readcyclecounter();
            const long long off1 = off + nin;                            // true start of frame k+1
            work(off1);
            if (PROF) pr_busy += (long long)__builtin_readcyclecounter() - pr_t0;
            lds_barrier();
            if (PROF) pr_iter += (long long)__builtin_readcyclecounter() - pr_t0;
            // ---- commit frame k; verify the speculation nin(k+1) == N ------------------------------
            const int nin_next = __builtin_amdgcn_readfirstlane(CT[CT_NIN_NEXT]);
            if (is_d) filled += nin;
            if (nin_next != N) {
                // Everything computed ahead assumed nin(k+1) == N (window length, nold, sample offsets).  Re-run it
                // from the state of frame k, which the rings still hold:  E(k+1) | C(k+1),E(k+2) | D(k+1),C(k+2),E(k+3)
                if (PROF) pr_redo++;
                redo1(off1, nin_next);
                lds_barrier();
                redo2(off1, nin_next);
                lds_barrier();
                redo3(off1, nin_next);
                lds_barrier();
            }
            off = off1;
            nin = nin_next;
            nslip += (nin_next != N) ? 1 : 0;
            frames++;
            kf++;
        }
    };
    auto nothing = [&](long long, int) {};
    if (wave == 0) {
        frame_loop([&](long long) { if (!(skip & 1)) chain(kf + 2, N); },                               // C(k+2), speculative
                   nothing,
                   [&](long long, int nn) { chain(kf + 1, nn); },
                   [&](long long, int) { chain(kf + 2, N); }, false);
    } else if (wave == 1) {
        frame_loop([&](long long off1) { if (!(skip & 2)) estimate(kf + 3, off1 + 2LL * N, N); },      // E(k+3), speculative
                   [&](long long off1, int nn) { estimate(kf + 1, off1, nn); },
                   [&](long long off1, int nn) { estimate(kf + 2, off1 + nn, N); },
                   [&](long long off1, int nn) { estimate(kf + 3, off1 + nn + N, N); }, false);
    } else if (wave == 2) {
        frame_loop([&](long long) { if (!(skip & 8)) tstage(kf, frames, nin); else if (lane == 0) CT[CT_NIN_NEXT] = N; },   // T(k)
                   nothing, nothing, nothing, false);
    } else {
        frame_loop([&](long long off1) {
                       if (skip & 4) return;
                       if (!(skip & 64)) {
                           // stage the next nin samples into the ring, issue the following prefetch
#pragma unroll
                           for (int k = 0; k < WP_KP; k++) { const int i = t + WP_DSP_THREADS * k; if (i < nin) ring_put_raw(RIDX(filled + i), pre[k], fmt_k); }
                           const long long nf = filled + nin, last = C.nsamples - 1;
#pragma unroll
                           for (int k = 0; k < WP_KP; k++) { long long i = nf + t + WP_DSP_THREADS * k; pre[k] = load_raw(C.raw, fmt_k, i < last ? i : last); }
                       }
                       dstage(kf + 1, off1, N);                          // D(k+1), speculative
                   },
                   nothing, nothing,
                   [&](long long off1, int nn) { dstage(kf + 1, off1, nn); }, true);
    }
    } else {
    while (off + nin <= C.nsamples && frames < C.cap_frames) {
        if (PROF) pr_t0 = (long long)__builtin_readcyclecounter();
        const long long off1 = off + nin;                                // true start of frame k+1
        if (wave == 0) {
            if (!(skip & 1)) chain(kf + 2, N);                           // C(k+2), speculative
        } else if (wave == 1) {
            if (!(skip & 2)) estimate(kf + 3, off1 + 2LL * N, N);        // E(k+3), speculative
        } else if (wave == 2) {
            if (!(skip & 8)) tstage(kf, frames, nin);                    // T(k)
            else if (lane == 0) CT[CT_NIN_NEXT] = N;
        } else if (skip & 4) {
        } else {
            // stage the next nin samples into the ring, issue the following prefetch
#ifdef WR_DBG_SKIP
            if (!(cfg.dbg_skip & 64)) {
#endif
#pragma unroll
            for (int k = 0; k < WP_KP; k++) { const int i = t + WP_DSP_THREADS * k; if (i < nin) ring_put_raw(RIDX(filled + i), pre[k], fmt_k); }
            {
                const long long nf = filled + nin, last = C.nsamples - 1;
#pragma unroll
                for (int k = 0; k < WP_KP; k++) { long long i = nf + t + WP_DSP_THREADS * k; pre[k] = load_raw(C.raw, fmt_k, i < last ? i : last); }
            }
#ifdef WR_DBG_SKIP
            }
#endif
            dstage(kf + 1, off1, N);                                     // D(k+1), speculative
        }
        if (PROF) pr_busy += (long long)__builtin_readcyclecounter() - pr_t0;
        lds_barrier();
        if (PROF) pr_iter += (long long)__builtin_readcyclecounter() - pr_t0;
        // ---- commit frame k; verify the speculation nin(k+1) == N ----------------------------------
        const int nin_next = __builtin_amdgcn_readfirstlane(CT[CT_NIN_NEXT]);
        if (wave >= 3) filled += nin;
        if (nin_next != N) {
            // Everything computed ahead assumed nin(k+1) == N (window length, nold, sample offsets).  Re-run it
            // from the state of frame k, which the rings still hold:  E(k+1) | C(k+1),E(k+2) | D(k+1),C(k+2),E(k+3)
            if (PROF) pr_redo++;
            if (wave == 1) estimate(kf + 1, off1, nin_next);
            lds_barrier();
            if (wave == 0) chain(kf + 1, nin_next);
            if (wave == 1) estimate(kf + 2, off1 + nin_next, N);
            lds_barrier();
            if (wave == 0) chain(kf + 2, N);
            if (wave == 1) estimate(kf + 3, off1 + nin_next + N, N);
            if (wave >= 3) dstage(kf + 1, off1, nin_next);
            lds_barrier();
        }
        off = off1;
        nin = nin_next;
        nslip += (nin_next != N) ? 1 : 0;
        frames++;
        kf++;
    }
    }
    if (PROF && C.prof && lane == 0) {
        // [0] chain busy  [1] estimator busy  [2] T busy  [3] D busy (wave 3)  [4] iteration total  [5] mispredictions  [6] frames
        if (wave == 0) C.prof[0] = pr_busy;
        if (wave == 1) C.prof[1] = pr_busy;
        if (wave == 2) { C.prof[2] = pr_busy; C.prof[4] = pr_iter; C.prof[5] = pr_redo; C.prof[6] = frames; }
        if (wave == 3) C.prof[3] = pr_busy;
    }

    // ================================ save carried state =======================================
    lds_barrier();
    if (frames > 0) {
        const int jl = kf - 1;                                           // last committed frame
        const float *FEk = FEr + (jl & 3) * NH;
        for (int i = tid; i < NH; i += WP_THREADS) st_fft[i] = FEk[i];
        for (int i = tid; i < nstash; i += WP_THREADS) st_old[i] = ring_get(RIDX(off - nstash + i));
        for (int i = tid; i < Nbits; i += WP_THREADS) st_sd[i] = SDL[i];
        if (tid < M) { hdr->phi_c[tid] = PHE[(jl % 3) * 4 + tid]; hdr->f_bin[tid] = CT[CT_FBIN + (jl & 3) * 4 + tid]; }
    }
    if (tid == 128) {                                                    // lane 0 of the T wave owns the timing scalars
        hdr->norm_rx_timing = norm_rx_timing_st;
        hdr->ppm = ppm;
        hdr->nin = nin;
        hdr->frames_total += frames;
        hdr->frames_call = frames;
        hdr->slips_call = nslip;
        hdr->consumed_call = off;
    }
#undef RIDX
}

