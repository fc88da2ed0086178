// demod_oct_impl.h -- the BATCH demodulator: eight captures per workgroup, ONE wavefront per capture, no speculation.
//
// Why (round 2).  The pipelined kernels (demod_pipe_impl.h, demod_tri_impl.h) overlap the stages of neighbouring frames of one
// capture; that is what makes a single stream fast, but for batches it pays twice: the order-dependent recurrences of the
// reference (NCO chain fsk.c:798,824 -- 4 lanes per capture; timing sum fsk.c:870-874 -- 2 lanes) are issued once per one or
// three captures, every stage needs rings in LDS (45 KB per capture), and each timing slip costs three re-run steps.  A batch
// has its parallelism ACROSS captures, so here a capture is processed strictly frame after frame (nin(k+1) is known before
// frame k+1 starts -- slips are free) by one wavefront that does all the WIDE stages of its capture, and the two NARROW
// stages are done for all eight captures of the workgroup at once by two extra wavefronts:
//
//     waves 0..7   capture wave c:  E(k) estimator | mix + slot-ordered integrate + timing products | atan2f, nin, decisions
//     wave  8      NCO chain of the eight captures (lanes 4c..4c+3 = tone x {re, im} of capture c), checkpoint every Ts/2 steps
//     wave  9      ordered timing sums of the eight captures (lanes 2c, 2c+1 = re, im)
//
// Per frame: [E(k)] barrier [chain(k)] barrier [mix/integrate(k)] barrier [sums(k)] barrier [decide(k), E(k+1)] ...  Two such
// workgroups share a CU (62 KB of LDS each), so one group's narrow phases run under the other's wide ones.
//
// Data movement: no sample ring.  A capture wave reads its frame straight from HBM -- lane l owns the Ts samples of symbol
// slot l (buffer positions Ts*l .. Ts*l+Ts-1 of the reference's Nmem-sample window), loaded one frame ahead -- mixes them with
// the phasors replayed from the chain's checkpoints (same instruction sequence => same bits), and integrates WITHOUT going
// through LDS: the reference sums the Ts circular-buffer slots in slot order (fsk.c:829-840), which for output i = Ts*l + r is
//        (prefix of length r of block l+1, summed left to right)  then  + d[Ts*l+r] + ... + d[Ts*l+Ts-1]
// i.e. the neighbour lane's running prefix sum (one DPP read) continued with the lane's own tail: 55 adds per component and
// block instead of 100, no index arithmetic, no bank conflicts.  The integrator outputs stay in registers; after the timing
// estimate the two outputs a symbol is resampled from (fsk.c:913-934) sit in the symbol's own lane or its neighbour.
//
// FAST = true (parity-ladder rung P3): same skeleton without waves 8 and 9 and without barriers.  The NCO
// phasor of sample s is read from the FFT twiddle table (tone frequencies are bin centres: e^{-j 2 pi bin s / Ndft} exactly
// periodic), the window sums use block prefix differences, the timing sum is a lane-local sum plus a wave reduction.  Tone bins
// are computed by the same estimator from the same samples (identical while nin is), nin from the fast timing estimate; a frame
// whose estimate lands within WO_GUARD of a decision threshold is counted in the state header (uncertain_call) so that the host
// can re-run that capture through the exact kernel.
#pragma once
#include <type_traits>

#include "demod_common.h"

#pragma clang fp contract(off)

#define WO_CAPS_MAX 15               // capture waves per workgroup (cfg.o_caps of them) + the duty wave (NCO chains, timing sums) <= 16 wavefronts
#define WO_GUARD 2e-5f              // |norm_rx_timing -+ 0.25| below this: the fast estimate does not decide nin(k+1) safely

namespace {

enum { OC_NIN = 0, OC_ALIVE = 1, OC_FBIN = 2 /* [2] this frame */, OC_FBINP = 4 /* [2] previous frame, first-run rule applied (fsk.c:750-753) */,
       OC_FBINN = 6 /* [2] next frame (estimated ahead, see the frame loop) */, OC_TC = 10 /* float re, im: timing sum */, OC_INTS = 16 };

typedef __attribute__((address_space(3))) float oct_lds_f32;

// value of lane + 1 (DPP wave shift; the last lane reads 0)
__device__ __forceinline__ float lane_up(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ v2f lane_up(v2f v) { return (v2f){lane_up(v.x), lane_up(v.y)}; }

template <int N>
__device__ __forceinline__ float nco_steps(float own, float k1, float k2) {          // N steps of nco_step_split in one asm block
    float t1, t2;
#define WO_NCO1 "v_mul_f32 %1, %0, %3\n\ts_nop 0\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %2\n\t"
    static_assert(N == 4 || N == 5, "half a symbol of Ts 8 or 10");
    if (N == 4) asm(WO_NCO1 WO_NCO1 WO_NCO1 WO_NCO1 : "+v"(own), "=&v"(t1), "=&v"(t2) : "v"(k1), "v"(k2));
    else asm(WO_NCO1 WO_NCO1 WO_NCO1 WO_NCO1 WO_NCO1 : "+v"(own), "=&v"(t1), "=&v"(t2) : "v"(k1), "v"(k2));
#undef WO_NCO1
    return own;
}

}  // namespace

template <int M, int TS, bool FAST>
__global__ __launch_bounds__(1024, 4) void wenet_demod_oct_kernel(WrDemodCfg cfg, const WrChan *chans, int nchan) {
    static_assert(M == 2, "two tones (four would need two soft decisions per lane)");
    constexpr int H = TS / 2;                                            // checkpoint spacing = the unit of a timing slip
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = cfg.o_caps;                                            // captures (= capture waves) of this workgroup
    const bool is_cap = wave < G, is_chain = !FAST && wave == G, is_sum = is_chain;      // one duty wave: the chain, later the sums
    const int cap = is_cap ? wave : 0;
    const int ch = blockIdx.x * G + cap;
    const bool present = is_cap && ch < nchan;
    WrChan C = chans[ch < nchan ? ch : 0];
    if (!present) { C.nsamples = 0; C.cap_frames = 0; C.sd_out = nullptr; C.trace = nullptr; }

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned char *smem = smem_all + cap * cfg.o_cap_stride;
    float2 *FB = (float2 *)(smem + cfg.o_off_FB);                        // [Ndft] estimator FFT buffer ...
    float  *TPf = (float *)(smem + cfg.o_off_FB);                        // ... later the frame's timing products: a row of re, a row of im
    float  *FE2 = (float *)(smem + cfg.o_off_FE);                        // [2][Ndft/2] smoothed spectrum after this frame's estimator run | after the next one's
    float  *FW = (float *)(smem + cfg.o_off_FW);                         // [Ndft/2]
    float2 *CK = (float2 *)(smem + cfg.o_off_CK);                        // [M][o_nhb] phasor at the start of every half symbol
    int    *CT = (int *)(smem + cfg.o_off_CT);
    const float2 *tw_t = (const float2 *)(smem_all + cfg.o_off_TW);
    const float  *hann_t = (const float *)(smem_all + cfg.o_off_HANN);
    const int    *src_t = (const int *)(smem_all + cfg.o_off_SRC);
    const float2 *dphi_t = (const float2 *)(smem_all + cfg.o_off_DPHI);
    const float2 *pft_t = (const float2 *)(smem_all + cfg.o_off_PFT);
    const float2 *back_t = (const float2 *)(smem_all + cfg.o_off_BACK);
    const int ctw = cfg.o_cap_stride / 4;
    const int *CT0 = (const int *)(smem_all + cfg.o_off_CT);

    const int N = cfg.N, Nmem = cfg.Nmem, nstash = cfg.nstash, Ndft = cfg.Ndft, NH = cfg.Ndft / 2, L = cfg.L, NI = cfg.NI;
    const int NIq = (NI + 3) & ~3, NHB = cfg.o_nhb;
    const int NOUT = NI / TS;                                            // lanes that own integrator outputs (NI = (Nsym+1)*TS)
    constexpr int NE = 4;                                                // estimator samples per lane (Ndft = 256)

    WrChanHdr *hdr = (WrChanHdr *)C.state;
    float *st_fft = C.state + cfg.st_fft_est;
    float2 *st_old = (float2 *)(C.state + cfg.st_samp_old);
    float *st_sd = C.state + cfg.st_sd_last;
    const unsigned short *raw16 = (const unsigned short *)C.raw;
    const long long last_smp = C.nsamples > 0 ? C.nsamples - 1 : 0;

    // ---- shared tables and carried state -> LDS / registers ------------------------------------
    {
        float2 *tw_w = (float2 *)(smem_all + cfg.o_off_TW); float *hann_w = (float *)(smem_all + cfg.o_off_HANN);
        int *src_w = (int *)(smem_all + cfg.o_off_SRC); float2 *dphi_w = (float2 *)(smem_all + cfg.o_off_DPHI);
        float2 *pft_w = (float2 *)(smem_all + cfg.o_off_PFT);
        const int nt = blockDim.x;
        for (int i = tid; i < Ndft; i += nt) { tw_w[i] = cfg.tw[i]; hann_w[i] = cfg.hann[i]; src_w[i] = cfg.fft_src[i]; }
        for (int i = tid; i < NH; i += nt) dphi_w[i] = cfg.dphi_tab[i];
        for (int i = tid; i < NI; i += nt) pft_w[i] = cfg.phi_ft[i];
        float2 *back_w = (float2 *)(smem_all + cfg.o_off_BACK);
        for (int i = tid; i < 3 * NH; i += nt) back_w[i] = cfg.backoff_tab[i];
    }
    int nin = N;
    float sdl = 0.f;                                                     // this lane's last soft decision (re-emitted by a NaN frame, fsk.c:878-880)
    float norm_rx_timing_st = 0.f, ppm = 0.f;
    long long off = 0, frames = 0;
    int nslip = 0, nuncertain = 0;
    bool alive = false;
    if (is_cap) {
        for (int i = lane; i < NH; i += 64) FE2[i] = present ? st_fft[i] : 0.f;
        if (present) {
            if (lane < cfg.Nbits) sdl = st_sd[lane];
            nin = __builtin_amdgcn_readfirstlane(hdr->nin);
            norm_rx_timing_st = hdr->norm_rx_timing; ppm = hdr->ppm;
        }
        alive = present && (long long)nin <= C.nsamples && C.cap_frames > 0;
        if (lane < M) CT[OC_FBIN + lane] = present ? hdr->f_bin[lane] : 0;           // bins of the frame before this launch
        if (lane == 0) { CT[OC_NIN] = nin; CT[OC_ALIVE] = alive ? 1 : 0; }
    }
    // chain wave: lane 4c + 2m + part carries one component of phi_c[m] of capture c, in a register, across the frames
    float own = 0.f;
    if (is_chain) {
        const int cc = lane / (2 * M), m = (lane >> 1) % M, part = lane & 1;
        const int chc = blockIdx.x * G + cc;
        if (cc < G && chc < nchan) {
            const WrChanHdr *h = (const WrChanHdr *)chans[chc].state;
            own = part ? h->phi_c[m].y : h->phi_c[m].x;
        } else own = part ? 0.f : 1.f;
    }
    lds_barrier();

    auto cvt = [](unsigned w) -> float2 {                                // fsk_demod.c:283-284, exact in float
        return make_float2(((float)(w & 0xffu) - 127.0f) / 128.0f, ((float)((w >> 8) & 0xffu) - 127.0f) / 128.0f);
    };

    // ================================ capture-wave stages ======================================
    unsigned epre[NE];                                                   // estimator samples of the NEXT frame (its start is known a frame ahead)
    unsigned xr[TS];                                                     // this lane's symbol slot of the frame about to be mixed
    auto prefetch_est = [&](long long off_j) {
#pragma unroll
        for (int j = 0; j < NE; j++) { long long a = off_j + src_t[lane + 64 * j]; epre[j] = raw16[a < last_smp ? a : last_smp]; }
    };
    auto prefetch_slot = [&](long long off_j, int nin_j) {
        const long long a0 = off_j - (Nmem - nin_j) + TS * lane;         // (negative only in a launch's first frame: patched from samp_old)
#pragma unroll
        for (int u = 0; u < TS; u++) { long long a = a0 + u; a = a < 0 ? 0 : (a < last_smp ? a : last_smp); xr[u] = raw16[a]; }
    };

    // E(j): tone estimator (fsk.c:540-677) on the prefetched samples; one FFT (Ndft <= nin < 2 Ndft)
    int fecur = 0;                                                       // FE2[fecur]: spectrum after the estimator run of the frame in work
    auto estimate = [&](int nin_j) {                                     // reads FE2[fecur], leaves FE2[fecur ^ 1] and the bins in OC_FBINN
        const float *FEin = FE2 + fecur * NH;
        float *FEout = FE2 + (fecur ^ 1) * NH;
        const int fft_samps = nin_j - Ndft;                              // fsk.c:583-584 with fft_loops == 1
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int n = lane + 64 * j, idx = src_t[n];
            float2 v = make_float2(0.f, 0.f);
            if (idx < fft_samps) { const float h = hann_t[idx]; const float2 x = cvt(epre[j]); v = make_float2(h * x.x, h * x.y); }
            FB[n] = v;
        }
        wave_sync();
        for (int s = cfg.nstages - 1; s >= 0; s--) {
            const int m = cfg.mstage[s], p = cfg.radix[s], fs = cfg.fstride[s];
            const int lgm = 31 - __clz(m);
            const int nb = Ndft / p;
            for (int b = lane; b < nb; b += 64) {
                const int blk = b >> lgm, k = b & (m - 1);
                float2 *F = FB + blk * m * p + k;
                if (p == 4) {                                            // kf_bfly4 (kiss_fft.c:44-90)
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
                } else {                                                 // kf_bfly2 (kiss_fft.c:21-42)
                    const float2 t = cmul(F[m], tw_t[k * fs]);
                    const float2 f0 = F[0];
                    F[m] = make_float2(f0.x - t.x, f0.y - t.y);
                    F[0] = make_float2(f0.x + t.x, f0.y + t.y);
                }
            }
            wave_sync();
        }
        for (int i = lane; i < NH; i += 64) {                            // fsk.c:612-628
            const float2 v = FB[i];
            float mag = (v.x * v.x) + (v.y * v.y);
            if (i < cfg.f_min) mag = 0.f;
            if (cfg.f_max - 1 >= 0 && i >= cfg.f_max - 1) mag = 0.f;
            const float e = (FEin[i] * cfg.one_minus_tc) + (sqrtf(mag) * cfg.tc);
            FEout[i] = e;
            FW[i] = e;
        }
        wave_sync();
        int fbin[M];
#pragma unroll
        for (int k = 0; k < M; k++) {                                    // fsk.c:633-654
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
        if (fbin[0] > fbin[1]) { const int t = fbin[0]; fbin[0] = fbin[1]; fbin[1] = t; }     // fsk.c:658-667 (M == 2)
        if (lane == 0) { CT[OC_FBINN] = fbin[0]; CT[OC_FBINN + 1] = fbin[1]; }
        wave_sync();
    };
    // the frame whose estimator run is in OC_FBINN / FE2[fecur ^ 1] becomes the frame in work.  First-run rule (fsk.c:750-753): while
    // the stored estimate of tone 0 is below 1 Hz the old part of the frame is mixed with the NEW estimates
    auto commit_estimate = [&]() {
        if (lane == 0) {
            const bool first = CT[OC_FBIN] < cfg.o_first_bins;                // bin_freq[stored bin of tone 0] < 1.0f
#pragma unroll
            for (int m = 0; m < M; m++) { const int nb = CT[OC_FBINN + m]; CT[OC_FBINP + m] = first ? nb : CT[OC_FBIN + m]; CT[OC_FBIN + m] = nb; }
        }
        fecur ^= 1;
        wave_sync();
    };

    v2f F[M][TS];                                                        // integrator outputs of this lane's symbol slot (fsk.c:803-841)
    // D(j): mix, integrate, timing products
    auto dstage = [&](long long off_j, int nin_j) {
        const int nold = Nmem - nin_j;
        float2 x[TS];
#pragma unroll
        for (int u = 0; u < TS; u++) x[u] = cvt(xr[u]);
        if (off_j < (long long)nold) {                                   // first frame of a launch: the window starts in the carried samp_old[]
#pragma unroll
            for (int u = 0; u < TS; u++) { const long long a = off_j - nold + TS * lane + u; if (a < 0 && present) x[u] = st_old[nstash + a]; }
        }
        float ft1[TS];
#pragma unroll
        for (int m = 0; m < M; m++) {
            v2f d[TS];
            if (!FAST) {
                const float2 dA2 = dphi_t[CT[OC_FBINP + m]], dB2 = dphi_t[CT[OC_FBIN + m]];
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const int hb = 2 * lane + hh;
                    const float2 p2 = CK[m * NHB + (hb < NHB ? hb : NHB - 1)];
                    v2f phi = {p2.x, p2.y};
                    const bool segA = hb * H < nold;
                    const v2f dd = {segA ? dA2.x : dB2.x, segA ? dA2.y : dB2.y};
#pragma unroll
                    for (int u = 0; u < H; u++) {
                        const float2 mx = cmul(x[hh * H + u], make_float2(phi.x, -phi.y));       // fsk.c:796 / :822
                        d[hh * H + u] = (v2f){mx.x, mx.y};
                        if (u < H - 1) phi = cmul_pk(phi, dd);                                     // fsk.c:798 / :824 (replayed from the checkpoint)
                    }
                }
                // slot-ordered window sums (fsk.c:829-840), see the header
                v2f P[TS + 1];
                P[1] = (v2f){0.f, 0.f} + d[0];
#pragma unroll
                for (int n = 1; n < TS; n++) P[n + 1] = P[n] + d[n];
                F[m][0] = P[TS];
#pragma unroll
                for (int r = 1; r < TS; r++) {
                    v2f acc = lane_up(P[r]) + d[r];
#pragma unroll
                    for (int n = r + 1; n < TS; n++) acc = acc + d[n];
                    F[m][r] = acc;
                }
            } else {
#pragma clang fp contract(fast)
                // phasor of buffer position s: the old part of the frame turns with the previous bin, the new part with this frame's,
                // phase-continuous at s = nold (fsk.c:756-764,785-788); angles are multiples of 2 pi / Ndft
                const int bp = CT[OC_FBINP + m], bc = CT[OC_FBIN + m];
                const int s0 = TS * lane;
#pragma unroll
                for (int u = 0; u < TS; u++) {
                    const int s = s0 + u;
                    const int k = (s < nold) ? bp * s : bp * nold + bc * (s - nold);
                    const float2 w = tw_t[k & (Ndft - 1)];                // e^{-j 2 pi k / Ndft} = conj(phasor)
                    d[u] = (v2f){x[u].x * w.x - x[u].y * w.y, x[u].x * w.y + x[u].y * w.x};
                }
                v2f P[TS + 1];
                P[1] = d[0];
#pragma unroll
                for (int n = 1; n < TS; n++) P[n + 1] = P[n] + d[n];
                F[m][0] = P[TS];
#pragma unroll
                for (int r = 1; r < TS; r++) F[m][r] = (P[TS] - P[r]) + lane_up(P[r]);
            }
#pragma unroll
            for (int r = 0; r < TS; r++) {                               // fsk.c:862-868
                const float a = (F[m][r].x * F[m][r].x) + (F[m][r].y * F[m][r].y);
                ft1[r] = (m == 0) ? a : ft1[r] + a;
            }
        }
        if (!FAST) {
            if (lane < NOUT) {
#pragma unroll
                for (int r = 0; r < TS; r++) {
                    const float2 pf = pft_t[TS * lane + r];
                    TPf[TS * lane + r] = ft1[r] * pf.x;                  // fsk.c:870-871: the products; wave 9 adds them in order
                    TPf[NIq + TS * lane + r] = ft1[r] * pf.y;
                }
            }
        } else {
            float sr = 0.f, si = 0.f;
            if (lane < NOUT) {
#pragma unroll
                for (int r = 0; r < TS; r++) { const float2 pf = pft_t[TS * lane + r]; sr += ft1[r] * pf.x; si += ft1[r] * pf.y; }
            }
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) { sr += __shfl_xor(sr, sh, 64); si += __shfl_xor(si, sh, 64); }
            if (lane == 0) { ((float *)CT)[OC_TC] = sr; ((float *)CT)[OC_TC + 1] = si; }
        }
        wave_sync();
    };

    // T(j): timing estimate, nin of the next frame, resampling and decisions (fsk.c:876-993); returns nin(j+1)
    auto tstage = [&](long long fr) -> int {
        const float tcr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(((const float *)CT)[OC_TC])));
        const float tci = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(((const float *)CT)[OC_TC + 1])));
        int nin_next = nin;
        float tr_rxt = 0.f;
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
            if (norm_rx_timing > 0.25f) nin_next = N + TS / 2;
            else if (norm_rx_timing < -0.25f) nin_next = N - TS / 2;
            else nin_next = N;
            if (FAST && (fabsf(norm_rx_timing - 0.25f) < WO_GUARD || fabsf(norm_rx_timing + 0.25f) < WO_GUARD)) nuncertain++;
            nin_next = __builtin_amdgcn_readfirstlane(nin_next);
            const int low_sample = __builtin_amdgcn_readfirstlane((int)floorf(rx_timing));
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = __builtin_amdgcn_readfirstlane((int)ceilf(rx_timing));
            const float omf = 1 - fract;
            tr_rxt = rx_timing;
            // symbol `lane` is resampled between f_int[.][(lane+1)*P + low_sample] and [.. + high_sample]: for an offset o >= 0
            // that is output o of the NEXT lane's slot, for o < 0 output TS + o of this lane's
            auto pick = [&](int o, int m) -> v2f {
                const int r = o >= 0 ? o : TS + o;
                v2f v = F[m][0];
#pragma unroll
                for (int q = 1; q < TS; q++) if (r == q) v = F[m][q];
                return o >= 0 ? lane_up(v) : v;
            };
            float tmax[M];
#pragma unroll
            for (int m = 0; m < M; m++) {
                const v2f a = pick(low_sample, m), b = pick(high_sample, m);
                float tr = omf * a.x, ti = omf * a.y;
                tr = tr + fract * b.x;
                ti = ti + fract * b.y;
                tmax[m] = (tr * tr) + (ti * ti);
            }
            if (lane < WR_NSYM) sdl = sqrtf(tmax[0]) - sqrtf(tmax[1]);   // fsk.c:955-966
        }
        if (C.sd_out && lane < WR_NSYM) C.sd_out[fr * WR_NSYM + lane] = sdl;
        if (C.trace && lane == 0) {
            float *tr = C.trace + fr * WR_TRACE_FLOATS;
#pragma unroll
            for (int m = 0; m < WR_M_MAX; m++) tr[WR_TR_FEST + m] = (m < M) ? cfg.bin_freq[CT[OC_FBIN + (m < M ? m : 0)]] : 0.f;
            tr[WR_TR_NIN] = (float)nin_next;
            tr[WR_TR_NRT] = norm_rx_timing_st;
            tr[WR_TR_PPM] = ppm;
            tr[WR_TR_MEAN] = 0.f;                                        // (Eb/N0 accumulators: the stats path runs the pipelined kernels)
            tr[WR_TR_STD] = 0.f;
            tr[WR_TR_RXT] = tr_rxt;
        }
        return nin_next;
    };

    // ================================ narrow stages (exact mode) ===============================
    // C(j) of the captures in `mask`: lanes 2M c .. 2M c + 2M - 1
    auto chain = [&](int mask) {
        const int cc = lane / (2 * M);
        if (cc >= G || !((mask >> cc) & 1)) return;
        const int m = (lane >> 1) % M, part = lane & 1;
        const int *CTc = CT0 + cc * ctw;
        oct_lds_f32 *ck = (oct_lds_f32 *)(smem_all + cc * cfg.o_cap_stride + cfg.o_off_CK) + m * NHB * 2 + part;
        const int nin_j = CTc[OC_NIN];
        const int nold = Nmem - nin_j;
        const int bc = CTc[OC_FBIN + m], bp = CTc[OC_FBINP + m];
        const int ncase = (nin_j < N) ? 0 : ((nin_j > N) ? 2 : 1);
        const float2 bo = back_t[ncase * NH + bp];
        {
            const float oth = __shfl_xor(own, 1, 64);
            const v2f pc = {part ? oth : own, part ? own : oth};
            const v2f phi0 = cmul_pk((v2f){bo.x, bo.y}, pc);             // fsk.c:758-759
            own = part ? phi0.y : phi0.x;
        }
        const float2 d0 = dphi_t[bp], d1 = dphi_t[bc];
        float k1 = d0.x, k2 = part ? d0.y : -d0.y;
        const int hsw = nold / H;                                        // 3, 4 or 5: the half symbol that starts with the new samples
        int hb = 0;
        auto blocks = [&](int upto) {
            for (; hb < upto; hb++) { ck[2 * hb] = own; own = nco_steps<H>(own, k1, k2); }
        };
        auto swtch = [&]() {                                             // fsk.c:785-788: normalise, continue with this frame's estimate
            if (hb == hsw) {
                const float oth = __shfl_xor(own, 1, 64);
                const float re = part ? oth : own, im = part ? own : oth;
                const float av = sqrtf(re * re + im * im);
                own = own / av;
                k1 = d1.x; k2 = part ? d1.y : -d1.y;
            }
        };
        blocks(3); swtch(); blocks(4); swtch(); blocks(5); swtch();
        const int full = L / H;
        for (; hb + 4 <= full; hb += 4) {                                // four checkpoints per trip: a taken branch costs ~16 cycles
#pragma unroll
            for (int k = 0; k < 4; k++) { ck[2 * (hb + k)] = own; own = nco_steps<H>(own, k1, k2); }
        }
        blocks(full);
        if (full * H < L) {
            ck[2 * hb] = own;
            for (int s = full * H; s < L; s++) own = nco_step_split(own, k1, k2);
        }
    };

    // ordered timing sums (fsk.c:870-874) of the captures in `mask`: lanes 2c / 2c+1 add the re / im products of capture c
    auto tsum = [&](int mask) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        int sc = lane >> 1;
        const bool mine = sc < G && ((mask >> sc) & 1);
        if (!mine) sc = __builtin_ctz(mask);
        const float *row = (const float *)(smem_all + sc * cfg.o_cap_stride + cfg.o_off_FB) + (lane & 1) * NIq;
        const v4f *T4 = (const v4f *)row;
        float acc = 0.f;
        v4f bufA[4], bufB[4];
        int i = 0;
#define WO_ADD16(buf) do { _Pragma("unroll") for (int u = 0; u < 4; u++) { acc = acc + buf[u].x; acc = acc + buf[u].y; acc = acc + buf[u].z; acc = acc + buf[u].w; } } while (0)
#define WO_LD16(buf, at) do { _Pragma("unroll") for (int u = 0; u < 4; u++) buf[u] = T4[((at) >> 2) + u]; } while (0)
        if (NI >= 16) {
            WO_LD16(bufA, 0);
            for (i = 16; i + 32 <= NI; i += 32) {
                WO_LD16(bufB, i);
                WO_ADD16(bufA);
                asm volatile("" : "+v"(acc) : : "memory");
                WO_LD16(bufA, i + 16);
                WO_ADD16(bufB);
                asm volatile("" : "+v"(acc) : : "memory");
            }
            if (i + 16 <= NI) { WO_LD16(bufB, i); WO_ADD16(bufA); WO_ADD16(bufB); i += 16; }
            else WO_ADD16(bufA);
        }
#undef WO_ADD16
#undef WO_LD16
        for (; i < NI; i++) acc = acc + row[i];
        if (mine) ((float *)(smem_all + sc * cfg.o_cap_stride + cfg.o_off_CT))[OC_TC + (lane & 1)] = acc;
    };

    auto alive_mask = [&]() {
        int mk = 0;
        for (int c = 0; c < G; c++) mk |= (__builtin_amdgcn_readfirstlane(CT0[c * ctw + OC_ALIVE]) ? 1 : 0) << c;
        return mk;
    };

    // ================================ frame loop ===============================================
    // Exact mode, per frame k (four workgroup barriers):
    //   A  duty wave: NCO chains of frame k          | capture waves: E(k+1) AHEAD, assuming nin(k+1) = N (they would idle otherwise)
    //   B  capture waves: mix / integrate / timing products of frame k
    //   C  duty wave: ordered timing sums of frame k
    //   D  capture waves: timing estimate, nin(k+1), decisions; if nin(k+1) != N the estimator run of frame k+1 is repeated with the
    //      true nin (it reads the untouched spectrum of frame k); then frame k+1 becomes the frame in work
    int ran = 0;                                                         // duty wave: captures that demodulated at least one frame
    if (is_cap && alive) { prefetch_est(0); prefetch_slot(0, nin); estimate(nin); commit_estimate(); if (!FAST) prefetch_est(nin); }
    if (FAST) {
        // no shared stages: every capture wave runs on its own
        while (alive) {
            const long long off1 = off + nin;
            prefetch_est(off1);
            dstage(off, nin);
            const int nn = tstage(frames);
            const bool more = off1 + nn <= C.nsamples && frames + 1 < C.cap_frames;
            nslip += (nn != N) ? 1 : 0;
            off = off1; nin = nn; frames++;
            alive = more;
            if (alive) { prefetch_slot(off, nin); estimate(nin); commit_estimate(); }
        }
    } else {
        // development (WENET_RX_PROFILE=4): cycles of wave 0 and of the duty wave per phase, summed over the frames:
        //   [0] phase D (to its barrier)  [1] phase A  [2] phase B  [3] phase C  [5] duty wave: chain busy  [6] frames; duty wave at +8
        const bool pp = C.prof != nullptr && lane == 0 && (wave == 0 || is_chain);
        long long *pr = C.prof + (is_chain ? 8 : 0);
        long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = pp ? (long long)__builtin_readcyclecounter() : 0;
#define WO_STAMP(k) do { if (pp) { const long long t1 = (long long)__builtin_readcyclecounter(); pt[k] += t1 - t0; t0 = t1; } } while (0)
        for (;;) {
            lds_barrier();                                               // nin(k), bins(k), alive published
            WO_STAMP(0);
            const int mask = alive_mask();
            if (!mask) break;
            const long long off1 = off + nin;
            if (is_chain) { chain(mask); ran |= mask; }
            if (pp && is_chain) { pt[5] += (long long)__builtin_readcyclecounter() - t0; }
            if (is_cap && alive) { estimate(N); prefetch_est(off1 + N); }   // E(k+1) ahead; then the samples of E(k+2), one frame ahead again
            if (pp && !is_chain) { pt[4] += (long long)__builtin_readcyclecounter() - t0; }
            lds_barrier();                                               // checkpoints of frame k
            WO_STAMP(1);
            if (is_cap && alive) dstage(off, nin);
            lds_barrier();                                               // timing products
            WO_STAMP(2);
            if (is_sum) tsum(mask);
            lds_barrier();                                               // timing sums
            WO_STAMP(3);
            if (is_cap && alive) {
                const int nn = tstage(frames);
                const bool more = off1 + nn <= C.nsamples && frames + 1 < C.cap_frames;
                if (more) {
                    prefetch_slot(off1, nn);
                    if (nn != N) { prefetch_est(off1); estimate(nn); prefetch_est(off1 + nn); }      // (a timing slip: E(k+1) again)
                    commit_estimate();
                }
                if (lane == 0) { CT[OC_NIN] = nn; CT[OC_ALIVE] = more ? 1 : 0; }
                nslip += (nn != N) ? 1 : 0;
                off = off1; nin = nn; frames++;
                alive = more;
            }
        }
        if (pp) { for (int k = 0; k < 6; k++) pr[k] = pt[k]; if (!is_chain) pr[6] = frames; }
#undef WO_STAMP
    }

    // ================================ save carried state =======================================
    if (is_cap && present) {
        if (frames > 0) {
            for (int i = lane; i < NH; i += 64) st_fft[i] = FE2[fecur * NH + i];
            for (int i = lane; i < nstash; i += 64) st_old[i] = cvt(raw16[off - nstash + i]);     // fsk.c:851 (off >= nin > nstash)
            if (lane < cfg.Nbits) st_sd[lane] = sdl;
            if (lane < M) hdr->f_bin[lane] = CT[OC_FBIN + lane];
        }
        if (lane == 0) {
            hdr->norm_rx_timing = norm_rx_timing_st;
            hdr->ppm = ppm;
            hdr->nin = nin;
            hdr->frames_total += frames;
            hdr->frames_call = frames;
            hdr->slips_call = nslip;
            hdr->uncertain_call = nuncertain;
            hdr->consumed_call = off;
        }
    }
    if (is_chain) {                                                      // un-normalised, as saved at fsk.c:846
        const int cc = lane / (2 * M), m = (lane >> 1) % M, part = lane & 1;
        const int chc = blockIdx.x * G + cc;
        if (cc < G && chc < nchan && ((ran >> cc) & 1)) {
            WrChanHdr *h = (WrChanHdr *)chans[chc].state;
            if (part) h->phi_c[m].y = own; else h->phi_c[m].x = own;
        }
    }
}
