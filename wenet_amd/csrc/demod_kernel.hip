// demod_kernel.hip -- non-coherent M-FSK demodulator for gfx950 (MI355X).
//
// One 64-lane wavefront per channel walks that channel's modem frames in order, with all
// state that the reference carries in struct FSK (src/fsk.h:43-90) resident in LDS/registers
// for the whole launch.  Inside a frame every step that the reference's arithmetic leaves
// order-free is spread over the 64 lanes; the three float recurrences whose rounding depends on
// evaluation order (NCO phasor chain fsk.c:791-824, integrator slot sums :833-840, spectral-line
// sum :862-874) are evaluated in exactly the reference order, so soft decisions are bit-identical
// to the CPU pipeline.
//
// Reference map (file:line in /root/reference/src):
//   sample conversion            fsk_demod.c:273-296
//   tone estimator               fsk.c:540-677   (window :583-603, FFT kiss_fft.c, IIR :625-628, peaks :633-672)
//   NCO set-up / down-conversion fsk.c:756-842
//   state save                   fsk.c:845-851
//   fine timing, nin             fsk.c:858-907
//   resample / decide / soft out fsk.c:913-993, Eb/N0 accumulators :995-1007
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod_common.h"

#pragma clang fp contract(off)


// PROF: accumulate s_memtime deltas per phase into C.prof (development aid, separate instantiation)
#define WR_PROF_PHASES 12
#define PROF_MARK(k) do { if (PROF) { const long long _t = (long long)__builtin_readcyclecounter(); prof[k] += _t - t_last; t_last = _t; } } while (0)

// NT = threads per capture: 64 (one wavefront) or 512 (eight: the order-free stages are spread over all of them,
// the ordered recurrences run on wavefront 0) -- used for configurations too large for the pipelined kernel.
// BIG = the per-frame sample buffers (input block, mixed rows, phasor checkpoints, integrator outputs, timing products) live
// in a per-capture global scratch block instead of LDS: frame geometries that do not fit 160 KiB, i.e. the one-second
// frames of fsk_create (fsk.c:278-398, `fsk_demod -l`).  Same statements in the same order; workgroup barriers then also
// order global memory (one workgroup runs on one CU and shares its vector L1).
template <bool BIG> __device__ __forceinline__ void wg_barrier() { if constexpr (BIG) __syncthreads(); else lds_barrier(); }

template <int M, bool PROF, bool TLDS, int NT, bool BIG = false>
__global__ __launch_bounds__(NT) void wenet_demod_kernel(WrDemodCfg cfg, const WrChan *chans, int nchan) {
    const int ch = blockIdx.x;
    if (ch >= nchan) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WrChan C = chans[ch];

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define WR_FRAMEBUF(off) ((float2 *)((BIG ? C.big : smem) + (off)))      /* BIG is a compile-time constant: the address space folds */
    float2 *X  = WR_FRAMEBUF(cfg.off_X);          // [nstash + N + Ts/2]  old tail | new block
    float2 *FB = (float2 *)(smem + cfg.off_FB);   // [Ndft]               FFT work buffer
    float2 *PH = WR_FRAMEBUF(cfg.off_PH);         // [M][Lpad]            NCO phasors -> down-converted samples -> timing products
    float2 *FI = WR_FRAMEBUF(cfg.off_FI);         // [M][NI]              integrator outputs
    float  *FE = (float *)(smem + cfg.off_FE);    // [Ndft/2]             IIR-smoothed spectrum (fsk->fft_est)
    float  *FW = (float *)(smem + cfg.off_FW);    // [Ndft/2]             peak-search working copy
    float  *SDL = (float *)(smem + cfg.off_SD);   // [Nbits]              last soft decisions (kept across a NaN frame)
    float  *SC = (float *)(smem + cfg.off_SC);    // [4*Nsym + 16]        scratch
    float2 *CKb = WR_FRAMEBUF(cfg.off_CK);        // [2 segments][M][ckrow] every 8th NCO phasor (checkpoints)
    float2 *CKD = (float2 *)(smem + cfg.off_CKD); // [2 segments][M]        NCO step of each segment
    // configuration tables: LDS copies (TLDS) or the global originals (configurations too big for LDS)
    const float2 *tw_t   = TLDS ? (const float2 *)(smem + cfg.off_TW) : cfg.tw;
    const float  *hann_t = TLDS ? (const float *)(smem + cfg.off_HANN) : cfg.hann;
    const int    *src_t  = TLDS ? (const int *)(smem + cfg.off_SRC) : cfg.fft_src;
    const float2 *pft_t  = TLDS ? (const float2 *)(smem + cfg.off_PFT) : cfg.phi_ft;
    const float2 *dphi_t = TLDS ? (const float2 *)(smem + cfg.off_DPHI) : cfg.dphi_tab;

    const int Ts = cfg.Ts, N = cfg.N, P = cfg.P, Nmem = cfg.Nmem, nstash = cfg.nstash;
    const int Ndft = cfg.Ndft, NH = cfg.Ndft / 2, L = cfg.L, NI = cfg.NI, q = cfg.q, Lpad = cfg.Lpad;
    const int Nbits = cfg.Nbits;
    const int nsym_k = BIG ? cfg.Nsym : WR_NSYM;                          // symbols per frame (48 for fsk_create_hbr, N/Ts for fsk_create)
    const int scc = BIG ? 4 * cfg.Nsym : 120;                             // control slots behind the per-symbol scratch in SC

    // ---- load carried state ---------------------------------------------------------------
    WrChanHdr *hdr = (WrChanHdr *)C.state;
    float *st_fft = C.state + cfg.st_fft_est;
    float2 *st_old = (float2 *)(C.state + cfg.st_samp_old);
    float *st_sd = C.state + cfg.st_sd_last;
    for (int i = tid; i < NH; i += NT) FE[i] = st_fft[i];
    for (int i = tid; i < nstash; i += NT) X[i] = st_old[i];
    for (int i = tid; i < Nbits; i += NT) SDL[i] = st_sd[i];
    if (TLDS) {
        float2 *tw_w = (float2 *)(smem + cfg.off_TW); float *hann_w = (float *)(smem + cfg.off_HANN);
        int *src_w = (int *)(smem + cfg.off_SRC); float2 *pft_w = (float2 *)(smem + cfg.off_PFT);
        float2 *dphi_w = (float2 *)(smem + cfg.off_DPHI);
        for (int i = tid; i < Ndft; i += NT) { tw_w[i] = cfg.tw[i]; hann_w[i] = cfg.hann[i]; src_w[i] = cfg.fft_src[i]; }
        for (int i = tid; i < NI; i += NT) pft_w[i] = cfg.phi_ft[i];
        for (int i = tid; i < NH; i += NT) dphi_w[i] = cfg.dphi_tab[i];
    }
    float2 phi_c = hdr->phi_c[lane % M];   // meaningful in lanes 0..M-1
    int fbin_prev[M];
#pragma unroll
    for (int m = 0; m < M; m++) fbin_prev[m] = __builtin_amdgcn_readfirstlane(hdr->f_bin[m]);
    float norm_rx_timing_st = hdr->norm_rx_timing;
    float ppm = hdr->ppm;
    int nin = __builtin_amdgcn_readfirstlane(hdr->nin);
    if (tid == 0) { ((int *)SC)[126] = 0; ((int *)SC)[127] = 0; ((int *)SC)[128] = 0; ((int *)SC)[129] = 0; }      // progress counters of the streamed frame body
    __syncthreads();

    long long prof[WR_PROF_PHASES];
#pragma unroll
    for (int k = 0; k < WR_PROF_PHASES; k++) prof[k] = 0;
    long long t_last = PROF ? (long long)__builtin_readcyclecounter() : 0;

    // Input prefetch: while frame k is processed, the longest possible window of frame k+1
    // (N + Ts/2 samples from off+nin) is already in flight into registers.
    constexpr int KPRE = 8;
    const bool use_pre = (N + Ts / 2) <= NT * KPRE;
    uint2 pre[KPRE];
#pragma unroll
    for (int k = 0; k < KPRE; k++) pre[k] = make_uint2(0u, 0u);
    if (use_pre) {
        if (C.nsamples > 0) prefetch_raw<KPRE>(pre, C.raw, C.fmt, 0, C.nsamples - 1, tid, NT);
    }

    long long off = 0, frames = 0;
    while (off + nin <= C.nsamples && frames < C.cap_frames) {
        const int nold = Nmem - nin;                                    // fsk.c:698
        PROF_MARK(11);
        // ---- new samples -> X[nstash ..] ---------------------------------------------------
        if (use_pre) {
#pragma unroll
            for (int k = 0; k < KPRE; k++) { const int i = tid + NT * k; if (i < nin) X[nstash + i] = convert_raw(pre[k], C.fmt); }
            prefetch_raw<KPRE>(pre, C.raw, C.fmt, off + nin, C.nsamples - 1, tid, NT);   // clamped: samples past the end are never used
        } else {
            for (int i = tid; i < nin; i += NT) X[nstash + i] = load_sample(C.raw, C.fmt, off + i);
        }
        wg_barrier<BIG>();
        PROF_MARK(0);

        // ---- tone estimator (fsk.c:540-677) ------------------------------------------------
        const int fft_loops = nin / Ndft;
        for (int jl = 0; jl < fft_loops; jl++) {
            const int samps = nin - (jl + 1) * Ndft;                   // fsk.c:583
            const int fft_samps = samps >= Ndft ? Ndft : samps;        // fsk.c:584
            // window + digit-reversed placement (kf_work leaves, kiss_fft.c:273-278)
            for (int n = tid; n < Ndft; n += NT) {
                const int idx = src_t[n];
                float2 v = make_float2(0.f, 0.f);
                if (idx < fft_samps) {
                    const float h = hann_t[idx];
                    const float2 x = X[nstash + idx + Ndft * jl];
                    v = make_float2(h * x.x, h * x.y);
                }
                FB[n] = v;
            }
            wg_barrier<BIG>();
            for (int s = cfg.nstages - 1; s >= 0; s--) {               // innermost butterflies first
                const int m = cfg.mstage[s], p = cfg.radix[s], fs = cfg.fstride[s];
                const int lgm = 31 - __clz(m);
                const int nb = Ndft / p;
                for (int b = tid; b < nb; b += NT) {
                    const int blk = b >> lgm, k = b & (m - 1);           // m is a power of two (Ndft is)
                    float2 *F = FB + blk * m * p + k;
                    if (p == 4) {                                      // kf_bfly4 (kiss_fft.c:44-90), forward
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
                    } else {                                           // kf_bfly2 (kiss_fft.c:21-42)
                        const float2 t = cmul(F[m], tw_t[k * fs]);
                        const float2 f0 = F[0];
                        F[m] = make_float2(f0.x - t.x, f0.y - t.y);
                        F[0] = make_float2(f0.x + t.x, f0.y + t.y);
                    }
                }
                wg_barrier<BIG>();
            }
            // |X|^2, band limits, IIR (fsk.c:612-628)
            for (int i = tid; i < NH; i += NT) {
                const float2 v = FB[i];
                float mag = (v.x * v.x) + (v.y * v.y);
                if (i < cfg.f_min) mag = 0.f;
                if (cfg.f_max - 1 >= 0 && i >= cfg.f_max - 1) mag = 0.f;
                const float e = (FE[i] * cfg.one_minus_tc) + (sqrtf(mag) * cfg.tc);
                FE[i] = e;
                FW[i] = e;
            }
            wg_barrier<BIG>();
        }
        if (fft_loops == 0) {          // not reachable for hbr geometries (nin >= Ndft); defined behaviour anyway
            for (int i = tid; i < NH; i += NT) FW[i] = 0.f;
            wg_barrier<BIG>();
        }
        PROF_MARK(1);
        // M peaks: first-maximum argmax, blank +-f_zero, ascending sort (fsk.c:633-667) -- wavefront 0, then shared
        int fbin[M];
#pragma unroll
        for (int k = 0; k < M; k++) fbin[k] = 0;
        if (wave == 0) {
#pragma unroll
        for (int k = 0; k < M; k++) {
            BestBin best; best.v = 0.f; best.i = 0;
            for (int j = lane; j < NH; j += 64) {
                const float v = FW[j];
                if (v > best.v) { best.v = v; best.i = j; }
            }
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) {
                BestBin o;
                o.v = __shfl_xor(best.v, sh, 64);
                o.i = __shfl_xor(best.i, sh, 64);
                best = better(best, o);
            }
            // all-zero spectrum: best.v stays 0 and lanes disagree on .i only through ties at v==0,
            // where the reference keeps imax=0 (nothing is > 0)
            // (the butterfly leaves the same winner in every lane; readfirstlane tells the compiler it is uniform)
            const int imax = __builtin_amdgcn_readfirstlane((best.v > 0.f) ? best.i : 0);
            int lo = imax - cfg.f_zero; lo = lo < 0 ? 0 : lo;
            int hi = imax + cfg.f_zero; hi = hi > NH ? NH : hi;        // only bins < Ndft/2 are ever read again
            wave_sync();
            for (int j = lo + lane; j < hi; j += 64) FW[j] = 0.f;   // (wavefront 0 only)
            wave_sync();
            fbin[k] = imax;
        }
#pragma unroll
        for (int a = 1; a < M; a++) {                                  // ascending insertion sort of M ints
#pragma unroll
            for (int b = a; b > 0; b--) {
                if (fbin[b - 1] > fbin[b]) { const int t = fbin[b]; fbin[b] = fbin[b - 1]; fbin[b - 1] = t; }
            }
        }
        if (NT > 64 && lane == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) ((int *)SC)[scc + m] = fbin[m];
        }
        }   // wave == 0
        if (NT > 64) {
            wg_barrier<BIG>();
#pragma unroll
            for (int m = 0; m < M; m++) fbin[m] = __builtin_amdgcn_readfirstlane(((const int *)SC)[scc + m]);
        }
        // first run: no valid previous estimate (fsk.c:750-753)
        if (cfg.bin_freq[fbin_prev[0]] < 1.0f) {
#pragma unroll
            for (int m = 0; m < M; m++) fbin_prev[m] = fbin[m];
        }

        float tcr = 0.f, tci = 0.f;
        if (!BIG && NT > 64 && cfg.seq_stream) {
            // ================= streamed frame body (workgroups of eight waves) ===================================
            // The NCO chain is one dependent recurrence on wave 0 (24 cycles per sample); everything downstream of it
            // is consumed as it is produced instead of after it: the chain publishes how many samples have their
            // checkpoint stored, waves 2..7 mix + integrate + form the timing products block by block behind it, and
            // wave 1 adds the products in order as they appear.  Same arithmetic, same order; the frame costs about
            // estimator + chain instead of the sum of all stages.
            volatile int *ctl = (volatile int *)SC + 126;                  // [0] samples mixable, [1] timing products ready, [2] barrier count, [3] samples mixed
            const bool stamp = (C.prof != nullptr) && frames == 50;            // development: cycle stamps of one frame (WENET_RX_PROFILE=3)
            if (stamp && tid == 0) C.prof[0] = (long long)__builtin_readcyclecounter();
            float2 *TP = (float2 *)(smem + cfg.off_TP);
            const float2 *src = X + (nstash - nold);                       // fsk.c:775: old tail then new block, contiguous
            const int nA = (nold + 7) / 8;
            if (wave == 0) {
                __builtin_amdgcn_s_setprio(3);                                     // the chain is the frame's critical path
                if (tid < M) {
                    int bp = fbin_prev[0], bc = fbin[0];
#pragma unroll
                    for (int m = 1; m < M; m++) if (lane == m) { bp = fbin_prev[m]; bc = fbin[m]; }
                    const int ncase = (nin < N) ? 0 : ((nin > N) ? 2 : 1);
                    const float2 bo = cfg.backoff_tab[ncase * NH + bp];
                    v2f phi = cmul_pk((v2f){bo.x, bo.y}, (v2f){phi_c.x, phi_c.y});   // back the phase off (fsk.c:758-759)
                    float2 dd = dphi_t[bp];                                          // step with the PREVIOUS estimate
                    v2f d = {dd.x, dd.y};
                    const int ckrow = cfg.ckrow;
                    v2f *ckA = (v2f *)(CKb + (0 * M + lane) * ckrow);
                    v2f *ckB = (v2f *)(CKb + (1 * M + lane) * ckrow);
                    CKD[0 * M + lane] = dd;
                    int s = 0, c = 0;
                    for (; s + 8 <= nold; s += 8, c++) {
                        ckA[c] = phi;
#pragma unroll
                        for (int u = 0; u < 8; u++) phi = cmul_pk(phi, d);
                    }
                    if (s < nold) { ckA[c] = phi; for (; s < nold; s++) phi = cmul_pk(phi, d); }
                    {                                                              // comp_normalize, new estimate
                        const float av = sqrtf(phi.x * phi.x + phi.y * phi.y);
                        phi = (v2f){phi.x / av, phi.y / av};
                        dd = dphi_t[bc];
                        d = (v2f){dd.x, dd.y};
                    }
                    CKD[1 * M + lane] = dd;
                    asm volatile("" ::: "memory");
                    ctl[0] = nold;                                                 // (LDS executes a wave's stores in order)
                    c = 0;
                    for (; s + 32 <= L; s += 32, c += 4) {                         // four checkpoints per trip, then publish
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            ckB[c + k] = phi;
#pragma unroll
                            for (int u = 0; u < 8; u++) phi = cmul_pk(phi, d);
                        }
                        asm volatile("" ::: "memory");
                        ctl[0] = s + 32;
                    }
                    for (; s + 8 <= L; s += 8, c++) {
                        ckB[c] = phi;
#pragma unroll
                        for (int u = 0; u < 8; u++) phi = cmul_pk(phi, d);
                    }
                    if (s < L) { ckB[c] = phi; for (; s < L; s++) phi = cmul_pk(phi, d); }
                    asm volatile("" ::: "memory");
                    ctl[0] = L;
                    phi_c = make_float2(phi.x, phi.y);                             // saved un-normalised (fsk.c:846)
                }
                __builtin_amdgcn_s_setprio(0);
                if (stamp && tid == 0) C.prof[1] = (long long)__builtin_readcyclecounter();
            } else if (wave == 1) {
                __builtin_amdgcn_s_setprio(2);
                // ordered sum of the timing products (fsk.c:870) as far as they are published
                typedef float v4f __attribute__((ext_vector_type(4)));
                v2f acc = {0.f, 0.f};
                int i = 0;
                while (i < NI) {
                    int ready;
                    while ((ready = ctl[1]) <= i) __builtin_amdgcn_s_sleep(8);
                    for (; i + 16 <= ready; i += 16) {                             // 16 products per round: loads up front
                        const v4f *p4 = (const v4f *)(TP + i);
                        v4f w[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) w[u] = p4[u];
#pragma unroll
                        for (int u = 0; u < 8; u++) { acc = acc + w[u].xy; acc = acc + w[u].zw; }
                    }
                    if (ready >= NI) for (; i < NI; i++) { const float2 v = TP[i]; acc = acc + (v2f){v.x, v.y}; }
                }
                if (lane == 0) { SC[124] = acc.x; SC[125] = acc.y; }
                __builtin_amdgcn_s_setprio(0);
                if (stamp && lane == 0) C.prof[2] = (long long)__builtin_readcyclecounter();
            } else if (wave == 2) {
                // mixer: follows the chain, one lane per (checkpoint, tone), replaying the <= 8 chain steps after the checkpoint
                // (fsk.c:791,817); items are ordered checkpoint-major so that after every pass all tones are complete up to a
                // sample position, which is published for the integrators
                const int ngt = nA + (L - nold + 7) / 8;                           // checkpoints per tone
                int g = 0;                                                         // checkpoints finished (all tones)
                while (g < ngt) {
                    const int gs = g < nA ? g * 8 : nold + (g - nA) * 8;           // first sample of checkpoint g
                    int avail;
                    while ((avail = ctl[0]) <= gs) __builtin_amdgcn_s_sleep(4);
                    int gend = avail >= L ? ngt : (avail >= nold ? nA + (avail - nold) / 8 : avail / 8);     // whole checkpoints the chain has passed
                    if (gend > g + 64 / M) gend = g + 64 / M;                        // one pass: 64 items
                    if (gend <= g) { __builtin_amdgcn_s_sleep(4); continue; }
                    const int w = lane;
                    if (w < (gend - g) * M) {
                        const int gg = g + w / M, m = w - (w / M) * M;
                        const bool segB = gg >= nA;
                        const int cc = segB ? gg - nA : gg;
                        const int s0 = segB ? nold + cc * 8 : cc * 8;
                        const int send = segB ? L : nold;
                        const int cnt = (send - s0) < 8 ? (send - s0) : 8;
                        const float2 dd = CKD[(segB ? 1 : 0) * M + m];
                        const v2f d = {dd.x, dd.y};
                        v2f phi = ((const v2f *)(CKb + ((segB ? 1 : 0) * M + m) * cfg.ckrow))[cc];
                        float2 *row = PH + m * Lpad + s0;
                        for (int u = 0; u < cnt; u++) {
                            row[u] = cmul(src[s0 + u], make_float2(phi.x, -phi.y));
                            phi = cmul_pk(phi, d);
                        }
                    }
                    g = gend;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    ctl[3] = (g >= ngt) ? L : (g <= nA ? (g * 8 < nold ? g * 8 : nold) : nold + (g - nA) * 8);     // samples mixed, all tones
                }
            } else {
                constexpr int DW = NT / 64 - 3, DT = DW * 64;                      // integrator waves / threads
                const int BO = (NI + (NI + DT - 1) / DT - 1) / ((NI + DT - 1) / DT);      // outputs per block: equal blocks of at most one output per thread
                const int dt = tid - 192;
                int phase = 0;
                auto dbar = [&]() {                                                // barrier among the integrator waves (monotone LDS counter)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add((int *)&ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++phase;
                    while (__hip_atomic_load((int *)&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < DW * phase) __builtin_amdgcn_s_sleep(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                };
                for (int i0 = 0; i0 < NI; i0 += BO) {
                    const int i1 = (i0 + BO < NI) ? i0 + BO : NI;
                    int need = (i1 - 1) * q + Ts;                                  // samples the block's windows reach
                    if (need > L) need = L;
                    while (ctl[3] < need) __builtin_amdgcn_s_sleep(8);
                    const long long t_blk = stamp ? (long long)__builtin_readcyclecounter() : 0;
                    // integrate-and-dump of outputs [i0, i1): slot-ordered re-sum (fsk.c:829-840), one thread per output doing all
                    // tones and the timing product right away (fsk.c:862-870)
                    for (int i = i0 + dt; i < i1; i += DT) {
                        const int base = i * q;
                        const int r = base % Ts;
                        const int o0 = (r == 0) ? 0 : Ts - r;
                        float ft1 = 0.f;
                        v2f acc[M];
#pragma unroll
                        for (int m = 0; m < M; m++) acc[m] = (v2f){0.f, 0.f};
                        if ((Ts & (Ts - 1)) == 0 && Ts >= 8) {                         // power-of-two Ts: the slot wrap is a mask
                            const int msk = Ts - 1;
                            for (int j0 = 0; j0 < Ts; j0 += 8) {
                                v2f v[M][8];
#pragma unroll
                                for (int u = 0; u < 8; u++) {
                                    const int idx = base + ((o0 + j0 + u) & msk);
#pragma unroll
                                    for (int m = 0; m < M; m++) v[m][u] = ((const v2f *)PH)[m * Lpad + idx];
                                }
#pragma unroll
                                for (int u = 0; u < 8; u++) {
#pragma unroll
                                    for (int m = 0; m < M; m++) acc[m] = acc[m] + v[m][u];
                                }
                            }
                        } else {
                        int o = o0;
                        for (int j0 = 0; j0 < Ts; j0 += 8) {                           // 8 slots of all tones at a time: loads first, then the ordered adds
                            v2f v[M][8];
                            int oo = o;
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                const int idx = (j0 + u < Ts) ? base + oo : base;
#pragma unroll
                                for (int m = 0; m < M; m++) v[m][u] = ((const v2f *)PH)[m * Lpad + idx];
                                oo++;
                                if (oo == Ts) oo = 0;
                            }
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                if (j0 + u < Ts) {
#pragma unroll
                                    for (int m = 0; m < M; m++) acc[m] = acc[m] + v[m][u];
                                }
                            }
                            o = oo;
                        }
                        }
#pragma unroll
                        for (int m = 0; m < M; m++) {
                            FI[m * NI + i] = make_float2(acc[m].x, acc[m].y);
                            ft1 += (acc[m].x * acc[m].x) + (acc[m].y * acc[m].y);
                        }
                        const float2 pf = pft_t[i];
                        TP[i] = make_float2(ft1 * pf.x, ft1 * pf.y);
                    }
                    dbar();
                    if (dt == 0) ctl[1] = i1;
                    if (stamp && dt == 0 && i0 / BO < 8) { C.prof[4 + 2 * (i0 / BO)] = t_blk; C.prof[5 + 2 * (i0 / BO)] = (long long)__builtin_readcyclecounter(); }
                }
            }
#pragma unroll
            for (int m = 0; m < M; m++) fbin_prev[m] = fbin[m];                // fsk.c:847
            wg_barrier<BIG>();
            for (int i = tid; i < nstash; i += NT) X[i] = X[nstash + nin - nstash + i];     // fsk.c:851
            tcr = SC[124]; tci = SC[125];
            if (stamp && tid == 0) C.prof[3] = (long long)__builtin_readcyclecounter();
            if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; }  // for the next frame (ordered by the barriers below)
        } else {

            PROF_MARK(2);
            // ---- NCO phasor chain, lanes 0..M-1 (fsk.c:756-764, 781-798, 807-824) ---------------
            if (tid < M) {
                // lanes 0..M-1 of wavefront 0 carry one tone each.  Trip counts are wave-uniform (scalar loop control).
                int bp = fbin_prev[0], bc = fbin[0];
    #pragma unroll
                for (int m = 1; m < M; m++) if (lane == m) { bp = fbin_prev[m]; bc = fbin[m]; }
                const int ncase = (nin < N) ? 0 : ((nin > N) ? 2 : 1);
                const float2 bo = cfg.backoff_tab[ncase * NH + bp];
                v2f phi = cmul_pk((v2f){bo.x, bo.y}, (v2f){phi_c.x, phi_c.y});   // back the phase off (fsk.c:758-759)
                float2 dd = dphi_t[bp];                                          // step with the PREVIOUS estimate
                v2f d = {dd.x, dd.y};
                // Only every 8th phasor is stored (a store per step doubles the cost of the dependent chain); the
                // down-conversion threads replay the steps in between with the same instruction sequence.
                const int ckrow = cfg.ckrow;
                v2f *ckA = (v2f *)(CKb + (0 * M + lane) * ckrow);
                v2f *ckB = (v2f *)(CKb + (1 * M + lane) * ckrow);
                CKD[0 * M + lane] = dd;
                int s = 0, c = 0;
                for (; s + 8 <= nold; s += 8, c++) {
                    ckA[c] = phi;
    #pragma unroll
                    for (int u = 0; u < 8; u++) phi = cmul_pk(phi, d);
                }
                if (s < nold) { ckA[c] = phi; for (; s < nold; s++) phi = cmul_pk(phi, d); }
                {                                                              // comp_normalize, new estimate
                    const float av = sqrtf(phi.x * phi.x + phi.y * phi.y);
                    phi = (v2f){phi.x / av, phi.y / av};
                    dd = dphi_t[bc];
                    d = (v2f){dd.x, dd.y};
                }
                CKD[1 * M + lane] = dd;
                c = 0;
                for (; s + 8 <= L; s += 8, c++) {
                    ckB[c] = phi;
    #pragma unroll
                    for (int u = 0; u < 8; u++) phi = cmul_pk(phi, d);
                }
                if (s < L) { ckB[c] = phi; for (; s < L; s++) phi = cmul_pk(phi, d); }
                phi_c = make_float2(phi.x, phi.y);                             // saved un-normalised (fsk.c:846)
            }
    #pragma unroll
            for (int m = 0; m < M; m++) fbin_prev[m] = fbin[m];                // fsk.c:847
            wg_barrier<BIG>();

            PROF_MARK(3);
            // ---- down-convert: sample * conj(phasor) (fsk.c:791,817); one thread per (tone, checkpoint) replays the
            //      <= 8 chain steps after its checkpoint ----------------------------------------------------------------
            {
                const float2 *src = X + (nstash - nold);                       // fsk.c:775: old tail then new block, contiguous
                const int nA = (nold + 7) / 8, nB = (L - nold + 7) / 8;
                const int per_tone = nA + nB;
                for (int w = tid; w < M * per_tone; w += NT) {
                    const int m = w / per_tone, c = w - m * per_tone;
                    const bool segB = c >= nA;
                    const int cc = segB ? c - nA : c;
                    const int s0 = segB ? nold + cc * 8 : cc * 8;
                    const int send = segB ? L : nold;
                    const int cnt = (send - s0) < 8 ? (send - s0) : 8;
                    const float2 dd = CKD[(segB ? 1 : 0) * M + m];
                    const v2f d = {dd.x, dd.y};
                    v2f phi = ((const v2f *)(CKb + ((segB ? 1 : 0) * M + m) * cfg.ckrow))[cc];
                    float2 *row = PH + m * Lpad + s0;
                    for (int u = 0; u < cnt; u++) {
                        row[u] = cmul(src[s0 + u], make_float2(phi.x, -phi.y));
                        phi = cmul_pk(phi, d);
                    }
                }
            }
            wg_barrier<BIG>();

            PROF_MARK(4);
            // ---- integrate-and-dump: every output re-sums the Ts circular-buffer slots in slot order
            //      (fsk.c:829-840).  Output i covers samples [i*q, i*q+Ts); sample s sits in slot s % Ts.
            for (int i = tid; i < NI; i += NT) {
                const int base = i * q;
                const int r = base % Ts;
                int o = (r == 0) ? 0 : Ts - r;                                 // window offset of slot 0
                v2f acc[M];
    #pragma unroll
                for (int m = 0; m < M; m++) acc[m] = (v2f){0.f, 0.f};
                for (int j0 = 0; j0 < Ts; j0 += 8) {                           // 8 slots at a time: loads first, then the ordered adds
                    v2f v[M][8];
                    int oo = o;
    #pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int idx = (j0 + u < Ts) ? base + oo : base;      // padding reads a valid address, value unused
    #pragma unroll
                        for (int m = 0; m < M; m++) v[m][u] = ((const v2f *)PH)[m * Lpad + idx];
                        oo++;
                        if (oo == Ts) oo = 0;
                    }
    #pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (j0 + u < Ts) {
    #pragma unroll
                            for (int m = 0; m < M; m++) acc[m] = acc[m] + v[m][u];
                        }
                    }
                    o = oo;
                }
    #pragma unroll
                for (int m = 0; m < M; m++) FI[m * NI + i] = make_float2(acc[m].x, acc[m].y);
            }
            wg_barrier<BIG>();

            PROF_MARK(5);
            // ---- stash the tail of the new block for the next frame (fsk.c:851) ------------------
            for (int i = tid; i < nstash; i += NT) X[i] = X[nstash + nin - nstash + i];

            // ---- fine timing: sum_i (sum_m |f_int|^2) * phi_ft[i]  (fsk.c:858-874) ---------------
            float2 *TP = BIG ? WR_FRAMEBUF(cfg.off_TP) : PH;                   // (LDS form: the down-converted samples are dead now)
            for (int i = tid; i < NI; i += NT) {
                float ft1 = 0.f;
    #pragma unroll
                for (int m = 0; m < M; m++) {
                    const float2 v = FI[m * NI + i];
                    ft1 += (v.x * v.x) + (v.y * v.y);
                }
                const float2 pf = pft_t[i];
                TP[i] = make_float2(ft1 * pf.x, ft1 * pf.y);
            }
            wg_barrier<BIG>();
            PROF_MARK(6);
            if (NT == 64 || wave == 0) {
                // sequential float accumulation in index order (fsk.c:870): one packed add per product
                // (re and im sums are independent chains); every lane runs the uniform loop, lane 0's value
                // is used.  The next 8 products are loaded (128-bit LDS reads) before the current 8 are added.
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
                        asm volatile("" : "+v"(acc) : : "memory");           // keep the reload of A behind its last use (no register copies)
    #pragma unroll
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
                tcr = acc.x; tci = acc.y;
            }
            if (NT > 64) {                                                     // wavefront 0's sums to everyone
                if (tid == 0) { SC[scc + 4] = tcr; SC[scc + 5] = tci; }
                wg_barrier<BIG>();
                tcr = SC[scc + 4]; tci = SC[scc + 5];
            }
        }
        tcr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(tcr)));   // lane 0's sums, as wave-uniform values
        tci = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(tci)));
        PROF_MARK(7);

        int nin_next = nin;
        float tr_mean = 0.f, tr_std = 0.f, tr_rxt = 0.f;
        const bool nan_frame = (tcr != tcr) || (tci != tci);               // fsk.c:878-880: return, outputs untouched
        if (!nan_frame) {
            // fsk.c:883-907 (double-typed sub-expressions written out)
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

            // ---- resample, decide, soft decisions (fsk.c:913-993) ---------------------------
            const int low_sample = (int)floorf(rx_timing);
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = (int)ceilf(rx_timing);
            const float omf = 1 - fract;
            tr_rxt = rx_timing;
            for (int sy = tid; sy < nsym_k; sy += NT) {                        // (one trip for the 48-symbol frames: NT >= 64)
                const int st = (sy + 1) * P;
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
                if (cfg.stats) { SC[sy] = mx; SC[nsym_k + sy] = sqrtf(mx); }   // Eb/N0 accumulators (fsk.c:984-993)
                if (C.bits_out) {
                    uint8_t *bo = C.bits_out + frames * Nbits;
                    if (M == 2) bo[sy] = (uint8_t)(sym == 1);
                    else { bo[sy * 2 + 1] = (uint8_t)(sym & 1); bo[sy * 2] = (uint8_t)((sym & 2) >> 1); }
                }
#pragma unroll
                for (int m = 0; m < M; m++) tmax[m] = sqrtf(tmax[m]);
                if (M == 2) {
                    SDL[sy] = tmax[0] - tmax[1];
                } else {                                                   // fsk.c:969-980, same accumulation order
                    float s1 = -tmax[0], s0 = -tmax[0];
                    s1 += tmax[1 % M];  s0 += -tmax[1 % M];
                    s1 += -tmax[2 % M]; s0 += tmax[2 % M];
                    s1 += tmax[3 % M];  s0 += tmax[3 % M];
                    SDL[sy * 2 + 1] = s1;
                    SDL[sy * 2] = s0;
                }
            }
            if (cfg.stats) {                                               // Eb/N0 accumulators (fsk.c:984-1007)
                wg_barrier<BIG>();
                if (tid == 0) {
                    float stdebno = 0.f, meanebno = 0.f;
                    for (int i = 0; i < nsym_k; i++) { stdebno += SC[i]; meanebno += SC[nsym_k + i]; }
                    meanebno = meanebno / cfg.nsym_f;
                    stdebno = (stdebno / cfg.nsym_f) - (meanebno * meanebno);
                    if ((double)stdebno > 0.0) stdebno = (float)sqrt((double)stdebno); else stdebno = 0.0f;
                    SC[2 * nsym_k] = meanebno;
                    SC[2 * nsym_k + 1] = stdebno;
                }
                wg_barrier<BIG>();
                tr_mean = SC[2 * nsym_k];
                tr_std = SC[2 * nsym_k + 1];
            }
            // ---- stats snapshot for the JSON side channel (fsk.c:1037-1066, fsk_demod.c:366-385):
            //      raw eye traces |f_int[m][ind]| and the smoothed spectrum; normalisation is host work
            if (C.dump && frames >= C.dump_first && ((frames - C.dump_first) % C.dump_period) == 0) {
                const long long slot = (frames - C.dump_first) / C.dump_period;
                if (slot < C.dump_cap) {
                    float *d = C.dump + slot * cfg.dump_floats;
                    const int neye = cfg.eye_traces * M * cfg.neyesamp;
                    for (int e = tid; e < neye; e += NT) {
                        const int j = e % cfg.neyesamp;
                        const int tm = e / cfg.neyesamp;                 // = i*M + m
                        const int i = tm / M, m = tm - i * M;
                        const int ind = 2 * P * i + (high_sample + 1) + j * cfg.eye_dec;
                        float v = 0.f;                                   // reference reads out of bounds when ind<0
                        if (ind >= 0 && ind < NI) { const float2 f = FI[m * NI + ind]; v = sqrtf(f.x * f.x + f.y * f.y); }
                        d[e] = v;
                    }
                    for (int i = tid; i < NH; i += NT) d[neye + i] = FE[i];
                    if (tid == 0) { d[neye + NH] = (float)high_sample; d[neye + NH + 1] = (float)frames; }
                }
            }
        }
        wg_barrier<BIG>();
        PROF_MARK(8);
        // ---- emit the frame's outputs (fsk_demod.c:403-407): a NaN frame re-emits the previous buffer
        if (C.sd_out) {
            float *so = C.sd_out + frames * Nbits;
            for (int i = tid; i < Nbits; i += NT) so[i] = SDL[i];
        }
        if (C.trace && tid == 0) {
            float *tr = C.trace + frames * WR_TRACE_FLOATS;
#pragma unroll
            for (int m = 0; m < WR_M_MAX; m++) tr[WR_TR_FEST + m] = (m < M) ? cfg.bin_freq[fbin[m < M ? m : 0]] : 0.f;
            tr[WR_TR_NIN] = (float)nin_next;
            tr[WR_TR_NRT] = norm_rx_timing_st;
            tr[WR_TR_PPM] = ppm;
            tr[WR_TR_MEAN] = nan_frame ? __int_as_float(0x7fc00000) : tr_mean;   // NaN marks a frame the reference returned early from (fsk.c:878-880): the host leaves EbNodB / snr_est alone
            tr[WR_TR_STD] = tr_std;
            tr[WR_TR_RXT] = tr_rxt;
        }
        off += nin;
        nin = nin_next;
        frames++;
        wg_barrier<BIG>();
        PROF_MARK(9);
    }
    if (PROF && C.prof && tid == 0) {
#pragma unroll
        for (int k = 0; k < WR_PROF_PHASES; k++) C.prof[k] = prof[k];
    }

    // ---- save carried state ---------------------------------------------------------------
    for (int i = tid; i < NH; i += NT) st_fft[i] = FE[i];
    for (int i = tid; i < nstash; i += NT) st_old[i] = X[i];
    for (int i = tid; i < Nbits; i += NT) st_sd[i] = SDL[i];
    if (tid < M) hdr->phi_c[tid] = phi_c;
    if (tid == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) hdr->f_bin[m] = fbin_prev[m];
        hdr->norm_rx_timing = norm_rx_timing_st;
        hdr->ppm = ppm;
        hdr->nin = nin;
        hdr->frames_total += frames;
        hdr->frames_call = frames;
        hdr->slips_call = 0;                                               // (no speculation in this kernel)
        hdr->consumed_call = off;
    }
}

// explicit instantiations + launcher
extern "C" hipError_t wr_launch_demod_pipe(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream, int prof);
extern "C" hipError_t wr_launch_demod_ex(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream, int prof) {
    if (nchan <= 0) return hipSuccess;
    if (cfg->pipe_ok && !cfg->big && prof != 2) return wr_launch_demod_pipe(cfg, d_chans, nchan, stream, prof);   // 8 waves per capture, pipelined
#define WR_LAUNCH(MM, PP, TT, NN)                                                                                        \
    do {                                                                                                                   \
        wr_attr_ok(hipFuncSetAttribute((const void *)wenet_demod_kernel<MM, PP, TT, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  cfg->lds_bytes));                                                                          \
        hipLaunchKernelGGL((wenet_demod_kernel<MM, PP, TT, NN>), dim3(nchan), dim3(NN), cfg->lds_bytes, stream, *cfg, d_chans, nchan); \
    } while (0)
#define WR_LAUNCH5(MM, PP, TT, NN, BB)                                                                                   \
    do {                                                                                                                   \
        wr_attr_ok(hipFuncSetAttribute((const void *)wenet_demod_kernel<MM, PP, TT, NN, BB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  cfg->lds_bytes));                                                                          \
        hipLaunchKernelGGL((wenet_demod_kernel<MM, PP, TT, NN, BB>), dim3(nchan), dim3(NN), cfg->lds_bytes, stream, *cfg, d_chans, nchan); \
    } while (0)
#define WR_LAUNCH_T(MM, PP, NN) do { if (cfg->tables_in_lds) WR_LAUNCH(MM, PP, true, NN); else WR_LAUNCH(MM, PP, false, NN); } while (0)
    if (cfg->big) {                                                    // frame buffers in global memory (fsk_create geometries); never profiled
        if (cfg->M == 2) { if (cfg->tables_in_lds) WR_LAUNCH5(2, false, true, 512, true); else WR_LAUNCH5(2, false, false, 512, true); }
        else             { if (cfg->tables_in_lds) WR_LAUNCH5(4, false, true, 512, true); else WR_LAUNCH5(4, false, false, 512, true); }
        return hipGetLastError();
    }
    // profiling (prof) keeps the one-wavefront form whose phase timings the instrumentation was written for
#ifdef WR_WITH_PROF
    if (cfg->M == 2) { if (prof) WR_LAUNCH_T(2, true, 64); else WR_LAUNCH_T(2, false, 512); }
    else             { if (prof) WR_LAUNCH_T(4, true, 64); else WR_LAUNCH_T(4, false, 512); }
#else                                                                   // (the instrumented one-wavefront instantiations: make PROF=1)
    if (cfg->M == 2) WR_LAUNCH_T(2, false, 512); else WR_LAUNCH_T(4, false, 512);
#endif
#undef WR_LAUNCH_T
#undef WR_LAUNCH5
#undef WR_LAUNCH
    return hipGetLastError();
}
extern "C" hipError_t wr_launch_demod(const WrDemodCfg *cfg, const WrChan *d_chans, int nchan, hipStream_t stream) {
    return wr_launch_demod_ex(cfg, d_chans, nchan, stream, 0);
}
