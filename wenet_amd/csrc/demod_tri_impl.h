// demod_tri_impl.h -- THREE captures per workgroup: the pipelined demod kernel (demod_pipe_impl.h, raw-cu8-ring batch form) with one
// NCO-chain wavefront shared by the three captures.  The chain is 36 % of the kernel's VALU instructions and uses 4 of 64 lanes;
// here lanes 4c..4c+3 carry capture c, so the same instructions serve three captures.  16 wavefronts:
//     wave 0        C(k+2) of all three captures
//     waves 1..3    E(k+3) of capture 0..2          waves 4..6    T(k) of capture 0..2
//     waves 7..15   D(k+1): three wavefronts per capture
// Each capture keeps its own LDS block (cfg.p_cap_stride bytes, same layout as the one-capture kernel), the configuration tables
// are shared.  The captures advance in lock-step (one workgroup barrier per frame); a timing slip of one capture costs all
// three the three re-run steps.  Statements and their order per capture are those of demod_pipe_impl.h: same bits.
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

#define WP_THREADS 1024
#define WT_CAPS 3
#define WT_CTHREADS 320              // threads that belong to one capture: its E, T and three D waves
#define WP_DSP_THREADS 192          // three D waves per capture
#define WP_KP 3                     // raw samples prefetched per D thread (3*192 >= N+Ts/2 is required)
#define WP_DSP_WAVES 3
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

enum { CT_NIN_NEXT = 0, CT_CNT = 1, CT_CONT = 2 /* this capture has another frame */, CT_FBIN = 4 /* [4 frames][4 tones] */, CT_INTS = 24 };


#include "demod_chain_split.h"

}  // namespace

// RAW: every capture of the launch is cu8 -> the sample ring keeps the raw byte pairs (2 B instead of 8 B per
// sample; (u8-127)/128 is exact, so converting at each read gives the same floats) and the timing-product
// phasors stay in global memory.  That brings the workgroup under a third of a CU's LDS and, with the register
// bound below, lets THREE captures share a CU instead of two.
// LIVE: the instantiation for live ticks whose chunks arrive beside the launch (demod_pipe_arrive.inc; in the batch instantiation none of that is compiled:
// its waits, words and agent-scope loads cost the three-capture kernel 3 % per frame even when unused -- 27 more scalar registers spilled)
template <int M, bool LIVE>
__global__ __launch_bounds__(WP_THREADS, 4) void wenet_demod_tri_kernel(WrDemodCfg cfg, const WrChan *chans, int nchan) {
    constexpr bool RAW = true;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Role and capture of each of the 16 wavefronts: host-built tables (DemodTables::tri_cfg), two bits per wave -- consecutive waves sit on
    // consecutive SIMDs, and which waves share a SIMD is worth up to 25 % (the long serial waves want mix/integrate waves beside them).
    //   role: 0 chain, 1 estimator, 2 timing, 3 mix/integrate
    const int role  = (int)((cfg.tri_role >> (2 * wave)) & 3u);
    const int cap   = (int)((cfg.tri_cap >> (2 * wave)) & 3u);
    const int dwave = (int)((cfg.tri_dw >> (2 * wave)) & 3u);                // D wave index within its capture
    const bool is_chain = role == 0, is_e = role == 1, is_t = role == 2, is_d = role == 3;
    const int ctid = is_e ? lane : is_t ? 64 + lane : is_d ? 128 + dwave * 64 + lane : WT_CTHREADS;    // thread index within the capture's 320
    const int ch = blockIdx.x * WT_CAPS + cap;
    const bool present = ch < nchan;                                   // (the last workgroup may carry fewer than three captures)
    WrChan C = chans[present ? ch : 0];
    if (!present) { C.nsamples = 0; C.cap_frames = 0; C.sd_out = nullptr; C.bits_out = nullptr; C.trace = nullptr; C.dump = nullptr; }
    const int fmt_k = (int)WR_FMT_CU8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *smem = smem_all + cap * cfg.p_cap_stride;        // this capture's block; the shared tables are addressed from smem_all
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
    const float2 *tw_t = (const float2 *)(smem_all + cfg.p_off_TW);
    const float  *hann_t = (const float *)(smem_all + cfg.p_off_HANN);
    const int    *src_t = (const int *)(smem_all + cfg.p_off_SRC);
    const float2 *pft_t = (const float2 *)(smem_all + cfg.p_off_PFT);
    const float2 *dphi_t = (const float2 *)(smem_all + cfg.p_off_DPHI);

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
        float2 *tw_w = (float2 *)(smem_all + cfg.p_off_TW); float *hann_w = (float *)(smem_all + cfg.p_off_HANN);
        int *src_w = (int *)(smem_all + cfg.p_off_SRC);
        float2 *dphi_w = (float2 *)(smem_all + cfg.p_off_DPHI);
        for (int i = tid; i < Ndft; i += WP_THREADS) { tw_w[i] = cfg.tw[i]; hann_w[i] = cfg.hann[i]; src_w[i] = cfg.fft_src[i]; }
        for (int i = tid; i < NH; i += WP_THREADS) dphi_w[i] = cfg.dphi_tab[i];
    }
#define WP_ARRIVE_STAT (is_d && dwave == 0)
#define WP_ARRIVE_ON LIVE
#include "demod_pipe_arrive.inc"
#undef WP_ARRIVE_ON
#undef WP_ARRIVE_STAT
    int nin = N;
    if (!is_chain) {                                                     // every capture is loaded by its own five waves
        for (int i = ctid; i < NH; i += WT_CTHREADS) FEr[3 * NH + i] = present ? st_fft[i] : 0.f;       // "after frame -1" lives in slot 3
        for (int i = ctid; i < Nbits; i += WT_CTHREADS) SDL[i] = present ? st_sd[i] : 0.f;
        for (int i = ctid; i < nstash; i += WT_CTHREADS) ring_put_f(RIDX((long long)(i - nstash)), present ? st_old[i] : make_float2(0.f, 0.f));
        if (ctid < M) {
            PHE[2 * 4 + ctid] = present ? hdr->phi_c[ctid] : make_float2(1.f, 0.f);
            CT[CT_FBIN + 3 * 4 + ctid] = present ? hdr->f_bin[ctid] : 0;                                   // frame -1 -> slots 2 / 3
        }
        nin = present ? __builtin_amdgcn_readfirstlane(hdr->nin) : N;
        if (ctid == 0) { CT[CT_CNT] = 0; CT[30] = 0; CT[CT_NIN_NEXT] = nin; CT[CT_CONT] = (present && (long long)nin <= C.nsamples && C.cap_frames > 0) ? 1 : 0; }
        // first 4*Nmax samples into the ring
        const long long last = C.nsamples - 1;
        if constexpr (LIVE) await_samples(4LL * Nmax);
        for (long long i = ctid; i < 4LL * Nmax; i += WT_CTHREADS)
#define WP_LOAD_SAMPLE(i_) (LIVE ? load_sample(i_) : load_raw(C.raw, fmt_k, (i_)))
#include "demod_pipe_shared_1.inc"
    auto chain = [&](int j, int nin_j, int capmask) {                   // C(j) of the captures in capmask (lanes 4c..4c+3 carry capture c)
        // (the packed form of the one-capture kernel -- one lane per capture and tone, shorter dependent path -- measured 9 % slower
        //  here: with 16 wavefronts on the CU SIMD time still counts for more than the chain's latency)
        nco_chain_split<WT_CAPS>(j, nin_j, lane, M, N, NH, Nmem, L, (lds_i32 *)CT, (lds_f32 *)PHE, (lds_f32 *)CKb, (lds_f32 *)CKD,
                        (const lds_f32 *)dphi_t, cfg.bin_freq, cfg.backoff_tab, capmask, cfg.p_cap_stride / 4);
        wave_sync();
    };

    // D(j): mix + integrate + timing products of frame j.  Waves 3..7 (t = D thread index).
    const int t = dwave * 64 + lane;                                     // D thread index within the capture (0..191)
#include "demod_pipe_shared_2.inc"
                for (int r = dwave; r < TS; r += WP_DSP_WAVES) {
#include "demod_pipe_shared_3.inc"
    auto tstage = [&](int kf, long long frames, int nin_cur, int act_caps) {
        if (C.prof && lane == 0) C.prof[12] = (long long)__builtin_readcyclecounter();
        const float2 *FI = FIb + (kf & 1) * M * NI;
        const float2 *TP = TPb + (kf & 1) * NIq;
        float tcr, tci;
        {
            // ONE timing wave adds the products of all active captures: lanes 2c / 2c+1 carry the real / imaginary row of capture c
            // (490 dependent plain adds per frame, issued once for the three captures instead of once each); the sums go through
            // capture 0's control block, the other timing waves wait for the frame's flag.
            float *G = (float *)(smem_all + cfg.p_off_CT) + 24;              // [3 captures][re, im] sums, then the flag at [6]
            volatile int *gflag = (volatile int *)(smem_all + cfg.p_off_CT) + 30;
            const int summer = __builtin_ctz(act_caps);                      // the lowest capture that still runs does the adding
            if (cap == summer) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                int sc = lane >> 1;                                          // capture of this lane
                if (sc >= WT_CAPS || !((act_caps >> sc) & 1)) sc = cap;      // idle lanes add the wave's own capture again (result unused)
                const float *TPf = (const float *)((const unsigned char *)TP + (sc - cap) * cfg.p_cap_stride) + (lane & 1) * NIq;
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
                if (lane < 2 * WT_CAPS) G[lane] = acc;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *gflag = kf + 1;
            }
            while (*gflag != kf + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tcr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(G[2 * cap])));
            tci = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(G[2 * cap + 1])));
        }
        if (C.prof && lane == 0) C.prof[8] += (long long)__builtin_readcyclecounter() - C.prof[12];     // development: sum (or wait for it)
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
            if (C.prof && lane == 0) C.prof[9] += (long long)__builtin_readcyclecounter() - C.prof[12];   // development: ... + atan2f, nin
#include "demod_pipe_shared_4.inc"
        if (C.prof && lane == 0) C.prof[10] += (long long)__builtin_readcyclecounter() - C.prof[12];      // development: ... + resample/decide
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

    // the serial waves win VALU arbitration against the D waves that share their SIMDs
    {   // cfg.chain_prio = chain | T << 2 | estimator << 4 (two bits each; s_setprio takes an immediate)
        const int pr = is_chain ? (cfg.chain_prio & 3) : is_t ? ((cfg.chain_prio >> 2) & 3) : is_e ? ((cfg.chain_prio >> 4) & 3) : 0;
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    }

    // control words of all three captures (every wave reads them after each barrier: who continues, who slipped)
    const int *CT0 = (const int *)(smem_all + cfg.p_off_CT);
    const int ctw = cfg.p_cap_stride / 4;
    auto cont_mask = [&]() {
        int mk = 0;
#pragma unroll
        for (int c = 0; c < WT_CAPS; c++) mk |= (__builtin_amdgcn_readfirstlane(CT0[c * ctw + CT_CONT]) ? 1 : 0) << c;
        return mk;
    };

    // ================================ pipeline prologue ========================================
    //   E(0) | C(0),E(1) | D(0),C(1),E(2)         (frame 0 with the true nin, later frames speculative)
    long long off = 0, frames = 0;
    int act = cont_mask();                                               // captures that have a frame 0
    const bool mine0 = !is_chain && ((act >> cap) & 1);
    if (act) {
        if (is_e && mine0) estimate(0, 0, nin);
        lds_barrier();
        if (is_chain) {
#pragma unroll
            for (int c = 0; c < WT_CAPS; c++)                            // frame 0 runs with each capture's carried nin: one capture at a time
                if ((act >> c) & 1) chain(0, __builtin_amdgcn_readfirstlane(CT0[c * ctw + CT_NIN_NEXT]), 1 << c);
        }
        if (is_e && mine0) estimate(1, (long long)nin, N);
        lds_barrier();
        if (is_chain) chain(1, N, act);
        if (is_e && mine0) estimate(2, (long long)nin + N, N);
        if (is_d && mine0) dstage(0, 0, nin);
        lds_barrier();
    }
    // D-thread prefetch registers: samples [filled, filled + 3*192)
    uint2 pre[WP_KP];
#pragma unroll
    for (int k = 0; k < WP_KP; k++) pre[k] = make_uint2(0u, 0u);
    if (is_d && mine0) {
        const long long last = C.nsamples - 1;
        if constexpr (LIVE) await_samples(filled + WP_KP * WP_DSP_THREADS);
#pragma unroll
        for (int k = 0; k < WP_KP; k++) { long long i = filled + t + WP_DSP_THREADS * k; pre[k] = LIVE ? load_sample(i < last ? i : last) : load_raw(C.raw, fmt_k, i < last ? i : last); }
    }

    // ================================ frame loop ===============================================
    // One copy of the loop per role (see demod_pipe_impl.h).  The three captures share the barrier cadence: the loop runs while any
    // of them has a frame; a wave works when ITS capture has one.  After the barrier every wave reads which captures continue
    // (CT_CONT, written by each capture's T wave) and which slipped (nin(k+1) != N): a slip of any capture makes all waves
    // pass the three re-run barriers, only that capture's waves (and its lanes of the chain wave) do work in them.
    int kf = 0;                                                          // frame index within this launch (common: lock-step)
    int nslip = 0;                                                       // this capture's frames with nin(k+1) != N
    long long pr_busy = 0, pr_iter = 0, pr_t0 = 0;                       // development (WENET_RX_PROFILE=3): busy / total ticks of this wave's role
    const bool pp = C.prof != nullptr;
    auto frame_loop = [&](auto work, auto redo1, auto redo2, auto redo3) {
        while (act) {
            const bool mine = !is_chain && ((act >> cap) & 1);
            const long long off1 = off + nin;                            // true start of frame k+1 (this wave's capture)
            if (pp) pr_t0 = (long long)__builtin_readcyclecounter();
            work(off1, mine);
            if (pp) pr_busy += (long long)__builtin_readcyclecounter() - pr_t0;
            lds_barrier();
            if (pp) pr_iter += (long long)__builtin_readcyclecounter() - pr_t0;
            // ---- commit frame k; verify the speculation nin(k+1) == N ------------------------------
            int nn[WT_CAPS], slip = 0;
#pragma unroll
            for (int c = 0; c < WT_CAPS; c++) {
                nn[c] = __builtin_amdgcn_readfirstlane(CT0[c * ctw + CT_NIN_NEXT]);
                if (((act >> c) & 1) && nn[c] != N) slip |= 1 << c;
            }
            const int nin_next = is_chain ? N : nn[cap < WT_CAPS ? cap : 0];
            if (is_d && mine) filled += nin;
            if (slip) {
                const bool redo = !is_chain && ((slip >> cap) & 1);
                redo1(off1, nin_next, redo, slip, nn);
                lds_barrier();
                redo2(off1, nin_next, redo, slip, nn);
                lds_barrier();
                redo3(off1, nin_next, redo, slip, nn);
                lds_barrier();
            }
            if (mine) { off = off1; nin = nin_next; frames++; nslip += (nin_next != N) ? 1 : 0; }
            kf++;
            act = cont_mask();                                           // (written by the T waves before the first barrier of this iteration)
        }
    };
    auto nothing = [&](long long, int, bool, int, const int *) {};
#ifdef WR_DBG_SKIP                                                       // development build only (tools/gpu_stage_cost.sh): leave stages out
    const int skip = cfg.dbg_skip;                                       // 1 chain, 2 estimator, 4 D -- results are garbage by construction
#else
    constexpr int skip = 0;
#endif
    if (is_chain) {
        frame_loop([&](long long, bool) { if (!(skip & 1)) chain(kf + 2, N, act); },                                     // C(k+2), speculative, all captures at once
                   nothing,
                   [&](long long, int, bool, int slip, const int *nn) {
#pragma unroll
                       for (int c = 0; c < WT_CAPS; c++) if ((slip >> c) & 1) chain(kf + 1, nn[c], 1 << c);      // slipped captures one at a time
                   },
                   [&](long long, int, bool, int slip, const int *) { chain(kf + 2, N, slip); });
    } else if (is_e) {
        frame_loop([&](long long off1, bool mine) { if (mine && !(skip & 2)) estimate(kf + 3, off1 + 2LL * N, N); },  // E(k+3), speculative
                   [&](long long off1, int nn, bool redo, int, const int *) { if (redo) estimate(kf + 1, off1, nn); },
                   [&](long long off1, int nn, bool redo, int, const int *) { if (redo) estimate(kf + 2, off1 + nn, N); },
                   [&](long long off1, int nn, bool redo, int, const int *) { if (redo) estimate(kf + 3, off1 + nn + N, N); });
    } else if (is_t) {
        frame_loop([&](long long off1, bool mine) {
                       if (!mine) return;
                       tstage(kf, frames, nin, act);                     // T(k): leaves nin(k+1) in CT_NIN_NEXT
                       if (lane == 0) {                                  // does this capture have a frame k+1?
                           const int nn = CT[CT_NIN_NEXT];
                           CT[CT_CONT] = (off1 + nn <= C.nsamples && frames + 1 < C.cap_frames) ? 1 : 0;
                       }
                   },
                   nothing, nothing, nothing);
    } else {
        frame_loop([&](long long off1, bool mine) {
                       if (!mine || (skip & 4)) return;
                       // stage the next nin samples into the ring, issue the following prefetch
#pragma unroll
                       for (int k = 0; k < WP_KP; k++) { const int i = t + WP_DSP_THREADS * k; if (i < nin) ring_put_raw(RIDX(filled + i), pre[k], fmt_k); }
                       const long long nf = filled + nin, last = C.nsamples - 1;
                       if constexpr (LIVE) await_samples(nf + WP_KP * WP_DSP_THREADS);
#pragma unroll
                       for (int k = 0; k < WP_KP; k++) { long long i = nf + t + WP_DSP_THREADS * k; pre[k] = LIVE ? load_sample(i < last ? i : last) : load_raw(C.raw, fmt_k, i < last ? i : last); }
                       dstage(kf + 1, off1, N);                          // D(k+1), speculative
                   },
                   nothing, nothing,
                   [&](long long off1, int nn, bool redo, int, const int *) { if (redo) dstage(kf + 1, off1, nn); });
    }

    if (pp && lane == 0) {   // [0] chain busy  [1] estimator busy  [2] T busy  [3] D busy (first D wave)  [4] iteration total  [6] frames  (capture 0's buffer: chain)
        if (is_chain) C.prof[0] = pr_busy;
        if (is_e) C.prof[1] = pr_busy;
        if (is_t) { C.prof[2] = pr_busy; C.prof[4] = pr_iter; C.prof[5] = nslip; C.prof[6] = frames; }
        if (is_d && dwave == 0) C.prof[3] = pr_busy;
        if (is_d && dwave == 1) C.prof[7] = pr_busy;
    }
    // ================================ save carried state =======================================
    lds_barrier();
    if (!is_chain && present) {
        if (frames > 0) {
            const int jl = (int)frames - 1;                              // last committed frame (frame index == frames processed: lock-step from 0)
            const float *FEk = FEr + (jl & 3) * NH;
            for (int i = ctid; i < NH; i += WT_CTHREADS) st_fft[i] = FEk[i];
            for (int i = ctid; i < nstash; i += WT_CTHREADS) st_old[i] = ring_get(RIDX(off - nstash + i));
            for (int i = ctid; i < Nbits; i += WT_CTHREADS) st_sd[i] = SDL[i];
            if (ctid < M) { hdr->phi_c[ctid] = PHE[(jl % 3) * 4 + ctid]; hdr->f_bin[ctid] = CT[CT_FBIN + (jl & 3) * 4 + ctid]; }
        }
        if (is_t && lane == 0) {                                         // lane 0 of the T wave owns the timing scalars
            hdr->norm_rx_timing = norm_rx_timing_st;
            hdr->ppm = ppm;
            hdr->nin = nin;
            hdr->frames_total += frames;
            hdr->frames_call = frames;
            hdr->slips_call = nslip;
            hdr->consumed_call = off;
        }
    }
#undef RIDX
#undef WP_LOAD_SAMPLE
}

