/*
 * wenet_oracle.c -- TEST INFRASTRUCTURE ONLY (see wenet_oracle.h).
 *
 * Plain-C restatement of the reference algorithm for the receive hot path.
 * Each function cites the reference file:line it follows.  The arithmetic is
 * written so that every float/double/long-double rounding happens exactly where
 * the reference's C expressions round (x86-64 SSE2 evaluation, no FMA: built
 * with -ffp-contract=off); libm calls (cosf sinf atan2f log10f sqrtf) go to the
 * same glibc the reference would use on this host.
 *
 * It is deliberately structured differently from the reference (iterative FFT,
 * static Tanner graph with flat edge arrays, table-driven phi0, bit-mask UW
 * matcher) -- the same structure the HIP kernels use -- so that passing the
 * bit-exact comparison against oracle/_ref validates that structure.
 */
#include "wenet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_tables.inc"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================= */
/* complex helpers  (src/comp_prim.h:57-141)                                 */
/* ======================================================================= */
static inline ora_comp c_mul(ora_comp a, ora_comp b) {          /* comp_prim.h:57-65 */
    ora_comp r;
    r.real = a.real * b.real - a.imag * b.imag;
    r.imag = a.real * b.imag + a.imag * b.real;
    return r;
}
static inline ora_comp c_conj(ora_comp a) { a.imag = -a.imag; return a; } /* :47-55 */
static inline ora_comp c_expj(float phi) {                       /* comp_prim.h:95-100 */
    ora_comp r; r.real = cosf(phi); r.imag = sinf(phi); return r;
}
static inline float c_abs(ora_comp a) {                          /* comp_prim.h:87-90 */
    /* powf(x,2.0) is folded to x*x by gcc -O3 in the reference build; x*x is also the
       correctly rounded value, so the two agree (checked by tests against oracle/_ref). */
    return sqrtf(a.real * a.real + a.imag * a.imag);
}
static inline ora_comp c_normalize(ora_comp a) {                 /* comp_prim.h:133-139 */
    float av = c_abs(a);
    ora_comp b; b.real = a.real / av; b.imag = a.imag / av; return b;
}

/* ======================================================================= */
/* FFT: iterative form of kiss_fft's radix-4/2 decimation-in-time recursion   */
/*   src/kiss_fft.c:237-302 (kf_work), :44-90 (kf_bfly4), :21-42 (kf_bfly2),  */
/*   :308-330 (kf_factor), :339-368 (twiddles)                                */
/* ======================================================================= */
#define ORA_MAXSTAGES 16
typedef struct {
    int nfft, nstages;
    int radix[ORA_MAXSTAGES];    /* outermost first, as kf_factor emits them */
    int m[ORA_MAXSTAGES];
    ora_comp *tw;                /* nfft twiddles */
    int *src;                    /* leaf n reads input src[n] (digit reversal) */
} ora_fft;

static int ora_fft_init(ora_fft *st, int nfft) {
    int n = nfft, i, s;
    if (nfft < 2 || (nfft & (nfft - 1))) return -1;   /* Ndft is always a power of two (fsk.c:169-173) */
    st->nfft = nfft;
    st->nstages = 0;
    /* kf_factor (kiss_fft.c:308-330): 4s first, then a 2 */
    while (n > 1) {
        int p = (n % 4 == 0) ? 4 : 2;
        n /= p;
        st->radix[st->nstages] = p;
        st->m[st->nstages] = n;
        st->nstages++;
    }
    st->tw = (ora_comp *)malloc(sizeof(ora_comp) * nfft);
    st->src = (int *)malloc(sizeof(int) * nfft);
    for (i = 0; i < nfft; i++) {                      /* kiss_fft.c:356-364, _kiss_fft_guts.h:136-145 */
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        st->tw[i].real = (float)cosf((float)phase);
        st->tw[i].imag = (float)sinf((float)phase);
    }
    /* leaf index n = sum_s q_s*m_s  <-  input index sum_s q_s*fstride_s, fstride_s = prod_{t<s} radix_t
       (kf_work: child q of a stage reads f + q*fstride and writes Fout + q*m) */
    for (i = 0; i < nfft; i++) {
        int rem = i, idx = 0, fstride = 1;
        for (s = 0; s < st->nstages; s++) {
            int q = rem / st->m[s];
            rem -= q * st->m[s];
            idx += q * fstride;
            fstride *= st->radix[s];
        }
        st->src[i] = idx;
    }
    return 0;
}
static void ora_fft_free(ora_fft *st) { free(st->tw); free(st->src); }

static void ora_fft_forward(const ora_fft *st, const ora_comp *in, ora_comp *out) {
    int s, i, k, blk;
    int fstride[ORA_MAXSTAGES];
    for (i = 0; i < st->nfft; i++) out[i] = in[st->src[i]];       /* kf_work m==1 leaves */
    fstride[0] = 1;
    for (s = 1; s < st->nstages; s++) fstride[s] = fstride[s - 1] * st->radix[s - 1];
    for (s = st->nstages - 1; s >= 0; s--) {                         /* innermost butterflies first */
        const int m = st->m[s], p = st->radix[s], fs = fstride[s];
        const int span = m * p;
        for (blk = 0; blk < st->nfft; blk += span) {
            ora_comp *F = out + blk;
            if (p == 2) {                                            /* kf_bfly2 */
                for (k = 0; k < m; k++) {
                    ora_comp t = c_mul(F[m + k], st->tw[k * fs]);
                    F[m + k].real = F[k].real - t.real;  F[m + k].imag = F[k].imag - t.imag;
                    F[k].real += t.real;                 F[k].imag += t.imag;
                }
            } else {                                                 /* kf_bfly4, forward */
                for (k = 0; k < m; k++) {
                    ora_comp s0 = c_mul(F[k + m],     st->tw[k * fs]);
                    ora_comp s1 = c_mul(F[k + 2 * m], st->tw[k * fs * 2]);
                    ora_comp s2 = c_mul(F[k + 3 * m], st->tw[k * fs * 3]);
                    ora_comp s3, s4, s5, f0 = F[k];
                    s5.real = f0.real - s1.real;  s5.imag = f0.imag - s1.imag;
                    f0.real += s1.real;           f0.imag += s1.imag;
                    s3.real = s0.real + s2.real;  s3.imag = s0.imag + s2.imag;
                    s4.real = s0.real - s2.real;  s4.imag = s0.imag - s2.imag;
                    F[k + 2 * m].real = f0.real - s3.real;  F[k + 2 * m].imag = f0.imag - s3.imag;
                    f0.real += s3.real;           f0.imag += s3.imag;
                    F[k] = f0;
                    F[k + m].real     = s5.real + s4.imag;  F[k + m].imag     = s5.imag - s4.real;
                    F[k + 3 * m].real = s5.real - s4.imag;  F[k + 3 * m].imag = s5.imag + s4.real;
                }
            }
        }
    }
}

/* ======================================================================= */
/* FSK demodulator state  (src/fsk.h:43-90, src/modem_stats.h:46-72)          */
/* ======================================================================= */
#define ORA_M_MAX 4
#define ORA_EYE_TR 8      /* MODEM_STATS_ET_MAX      (modem_stats.h:40) */
#define ORA_EYE_IND 160   /* MODEM_STATS_EYE_IND_MAX (modem_stats.h:41) */

struct ora_fsk {
    int Ndft, Fs, N, Rs, Ts, Nmem, P, Nsym, Nbits, mode;
    int f1_tx, fs_tx;
    int est_min, est_max, est_space;
    float *hann;
    ora_comp phi_c[ORA_M_MAX];
    ora_fft fft;
    float norm_rx_timing;
    ora_comp *samp_old;
    int nstash;
    float *fft_est;
    float EbNodB;
    float f_est[ORA_M_MAX];
    float ppm;
    int nin;
    /* stats */
    float snr_est, st_rx_timing, foff, clock_offset;
    float rx_eye[ORA_EYE_TR][ORA_EYE_IND];
    int neyetr, neyesamp;
    float st_f_est[ORA_M_MAX];
    /* scratch */
    ora_comp *fftin, *fftout, *f_intbuf, *f_int[ORA_M_MAX];
};

/* Both constructors: lbr == 0 is fsk_create_hbr (fsk.c:128-259), lbr != 0 is fsk_create (fsk.c:278-398), which differs in
 * the frame length (one second, N = Fs), the oversampling (horus_P = 8, fsk.c:35), a fixed 1024-point estimator and the
 * estimator band (HORUS_MIN/MAX/MIN_SPACING, fsk.c:262-264). */
static ora_fsk *ora_fsk_new(int Fs, int Rs, int P, int M, int lbr) {
    ora_fsk *f;
    int i, m, Ndft = 0;
    int nsyms = 48;                                                  /* fsk.c:135 */
    if (lbr) P = 8;                                                  /* horus_P, fsk.c:35,308 */
    if (Fs <= 0 || Rs <= 0 || P <= 0) return NULL;                   /* asserts fsk.c:137-141 / 286-290 */
    if (Fs % Rs != 0) return NULL;                                   /* fsk.c:143 / 292 */
    if ((Fs / Rs) % P != 0) return NULL;                             /* fsk.c:145 / 294 */
    if (M != 2 && M != 4) return NULL;                               /* fsk.c:146 / 295 */
    if (lbr) nsyms = Fs / (Fs / Rs);                                 /* fsk.c:306,309: N = Fs, Nsym = N/Ts */
    f = (ora_fsk *)calloc(1, sizeof(*f));
    f->Fs = Fs; f->Rs = Rs; f->Ts = Fs / Rs;
    f->N = f->Ts * nsyms; f->P = P; f->Nsym = nsyms;
    f->Nmem = f->N + 2 * f->Ts;
    f->f1_tx = 1200; f->fs_tx = 400;                                 /* fsk_demod.c:214 */
    f->nin = f->N;
    f->mode = M;
    f->Nbits = (M == 2) ? f->Nsym : f->Nsym * 2;
    if (lbr) {
        Ndft = 1024;                                                 /* fsk.c:300 */
        f->est_min = 800; f->est_max = 2500; f->est_space = 100;     /* fsk.c:262-264,317-319 */
    } else {
        for (i = 1; i; i <<= 1) if (f->N & i) Ndft = i;              /* fsk.c:169-171: highest set bit */
        f->est_min = Rs / 4; if (f->est_min < 0) f->est_min = 0;     /* fsk.c:175-176 */
        f->est_max = (Fs / 2) - Rs / 4;                              /* fsk.c:178 */
        f->est_space = Rs - (Rs / 5);                                /* fsk.c:180 */
    }
    f->Ndft = Ndft;
    for (m = 0; m < M; m++) f->phi_c[m] = c_expj(0);                 /* fsk.c:184-185 */
    f->nstash = 4 * f->Ts;                                           /* fsk.c:187-189 */
    f->samp_old = (ora_comp *)calloc(f->nstash, sizeof(ora_comp));
    ora_fft_init(&f->fft, Ndft);
    f->fft_est = (float *)calloc(Ndft / 2, sizeof(float));
    f->hann = (float *)malloc(sizeof(float) * Ndft);
    {                                                                /* fsk.c:94-111 */
        ora_comp dphi = c_expj((2 * M_PI) / ((float)Ndft - 1));
        ora_comp rphi = {.5, 0};
        rphi = c_mul(c_conj(dphi), rphi);
        for (i = 0; i < Ndft; i++) {
            rphi = c_mul(dphi, rphi);
            f->hann[i] = .5 - rphi.real;
        }
    }
    f->norm_rx_timing = 0; f->EbNodB = 0; f->ppm = 0;
    /* stats_init fsk.c:402-434 */
    {
        int neyesamp_dec = ceil(((float)P * 2) / ORA_EYE_IND);
        f->neyesamp = (P * 2) / neyesamp_dec;
        f->neyetr = M * (ORA_EYE_TR / M);
    }
    f->snr_est = 0; f->st_rx_timing = 0;
    f->fftin = (ora_comp *)malloc(sizeof(ora_comp) * Ndft);
    f->fftout = (ora_comp *)malloc(sizeof(ora_comp) * Ndft);
    f->f_intbuf = (ora_comp *)malloc(sizeof(ora_comp) * f->Ts);
    for (m = 0; m < M; m++) f->f_int[m] = (ora_comp *)malloc(sizeof(ora_comp) * (nsyms + 1) * P);
    return f;
}

ora_fsk *ora_fsk_create_hbr(int Fs, int Rs, int P, int M) { return ora_fsk_new(Fs, Rs, P, M, 0); }   /* fsk.c:128-259 */
ora_fsk *ora_fsk_create(int Fs, int Rs, int M) { return ora_fsk_new(Fs, Rs, 8, M, 1); }             /* fsk.c:278-398 */

void ora_fsk_destroy(ora_fsk *f) {
    int m;
    if (!f) return;
    for (m = 0; m < f->mode; m++) free(f->f_int[m]);
    free(f->f_intbuf); free(f->fftout); free(f->fftin); free(f->hann); free(f->fft_est);
    ora_fft_free(&f->fft); free(f->samp_old); free(f);
}

void ora_fsk_set_est_limits(ora_fsk *f, int est_min, int est_max) {  /* fsk.c:522-528 */
    f->est_min = est_min; if (f->est_min < 0) f->est_min = 0;
    f->est_max = est_max;
}
int ora_fsk_nin(const ora_fsk *f) { return f->nin; }

int ora_fsk_geom(const ora_fsk *f, int what) {
    switch (what) {
    case 0: return f->Ndft; case 1: return f->N; case 2: return f->Ts; case 3: return f->Nmem;
    case 4: return f->P; case 5: return f->Nsym; case 6: return f->Nbits; case 7: return f->nstash;
    case 8: return f->mode; case 9: return f->est_min; case 10: return f->est_max; case 11: return f->est_space;
    }
    return -1;
}
void ora_fsk_get_f_est(const ora_fsk *f, float out[4]) { memcpy(out, f->f_est, 4 * sizeof(float)); }
void ora_fsk_get_phi_c(const ora_fsk *f, float out[8]) { memcpy(out, f->phi_c, 8 * sizeof(float)); }
void ora_fsk_get_fft_est(const ora_fsk *f, float *out) { memcpy(out, f->fft_est, sizeof(float) * f->Ndft / 2); }
void ora_fsk_get_hann(const ora_fsk *f, float *out) { memcpy(out, f->hann, sizeof(float) * f->Ndft); }
void ora_fsk_get_samp_old(const ora_fsk *f, float *out) { memcpy(out, f->samp_old, sizeof(ora_comp) * f->nstash); }
float ora_fsk_get_scalar(const ora_fsk *f, int what) {
    switch (what) {
    case 0: return f->norm_rx_timing; case 1: return f->ppm; case 2: return f->EbNodB;
    case 3: return f->snr_est; case 4: return f->st_rx_timing; case 5: return f->foff;
    }
    return 0;
}
int ora_fsk_get_eye(const ora_fsk *f, float *out, int *neyetr, int *neyesamp) {
    memcpy(out, f->rx_eye, sizeof(f->rx_eye));
    *neyetr = f->neyetr; *neyesamp = f->neyesamp;
    return 0;
}

/* ---- tone frequency estimator: fsk.c:540-677 --------------------------- */
static void ora_freq_est(ora_fsk *f, const ora_comp *in, float *freqs, int M) {
    const int Ndft = f->Ndft, Fs = f->Fs, nin = f->nin;
    int i, j, k;
    int freqi[ORA_M_MAX];
    int f_min = (f->est_min * Ndft) / Fs;                            /* fsk.c:568-570, int division */
    int f_max = (f->est_max * Ndft) / Fs;
    int f_zero = (f->est_space * Ndft) / Fs;
    float tc = 0.95 * Ndft / Fs;                                     /* fsk.c:573: double expr -> float */
    int fft_loops = nin / Ndft;                                      /* fsk.c:577 */
    ora_comp *fftin = f->fftin, *fftout = f->fftout;

    for (j = 0; j < fft_loops; j++) {
        int samps = nin - ((j + 1) * Ndft);                          /* fsk.c:583 */
        int fft_samps = (samps >= Ndft) ? Ndft : samps;              /* fsk.c:584 */
        for (i = 0; i < fft_samps; i++) {                            /* fsk.c:587-597 */
            float hann = f->hann[i];
            fftin[i].real = hann * in[i + Ndft * j].real;
            fftin[i].imag = hann * in[i + Ndft * j].imag;
        }
        for (; i < Ndft; i++) { fftin[i].real = 0; fftin[i].imag = 0; } /* fsk.c:600-603 */
        ora_fft_forward(&f->fft, fftin, fftout);                     /* fsk.c:606 */
        for (i = 0; i < Ndft / 2; i++)                               /* fsk.c:612-614 */
            fftout[i].real = (fftout[i].real * fftout[i].real) + (fftout[i].imag * fftout[i].imag);
        for (i = 0; i < f_min; i++) fftout[i].real = 0;              /* fsk.c:617-619 */
        if (f_max - 1 >= 0)   /* the reference's index is size_t: f_max==0 wraps and the loop body never runs */
            for (i = f_max - 1; i < Ndft / 2; i++) fftout[i].real = 0;   /* fsk.c:620-622 */
        for (i = 0; i < Ndft / 2; i++) {                             /* fsk.c:625-628 */
            f->fft_est[i] = (f->fft_est[i] * (1 - tc)) + (sqrtf(fftout[i].real) * tc);
            fftout[i].imag = f->fft_est[i];
        }
    }
    for (k = 0; k < M; k++) {                                        /* fsk.c:635-654 */
        int imax = 0, lo, hi;
        float max = 0;
        for (j = 0; j < Ndft / 2; j++)
            if (fftout[j].imag > max) { max = fftout[j].imag; imax = j; }
        lo = imax - f_zero; lo = lo < 0 ? 0 : lo;
        hi = imax + f_zero; hi = hi > Ndft ? Ndft : hi;
        for (j = lo; j < hi; j++) fftout[j].imag = 0;
        freqi[k] = imax;
    }
    /* gnome sort fsk.c:658-667 == ascending sort of M small ints (stable order irrelevant: ints) */
    for (i = 1; i < M; i++) {
        int v = freqi[i];
        for (j = i; j > 0 && freqi[j - 1] > v; j--) freqi[j] = freqi[j - 1];
        freqi[j] = v;
    }
    for (i = 0; i < M; i++)                                          /* fsk.c:670-672 */
        freqs[i] = (float)(freqi[i]) * ((float)Fs / (float)Ndft);
}

/* ---- one modem frame: fsk.c:679-1108 ----------------------------------- */
void ora_fsk_demod_frame(ora_fsk *f, uint8_t *rx_bits, float *rx_sd, const ora_comp *fsk_in) {
    const int N = f->N, Ts = f->Ts, Rs = f->Rs, Fs = f->Fs, nsym = f->Nsym, nin = f->nin;
    const int P = f->P, Nmem = f->Nmem, M = f->mode, nstash = f->nstash;
    const int nold = Nmem - nin;                                     /* fsk.c:698 */
    int i, j, m;
    ora_comp phi_c[ORA_M_MAX], dphi[ORA_M_MAX], t[ORA_M_MAX], t_c, phi_ft, dphift;
    float f_est[ORA_M_MAX], tmax[ORA_M_MAX];
    float ft1, rx_timing, norm_rx_timing, old_norm_rx_timing, d_norm_rx_timing, appm;
    float meanebno, stdebno, fc_avg, fc_tx, eye_max;
    ora_comp *f_intbuf = f->f_intbuf;

    for (m = 0; m < M; m++) phi_c[m] = f->phi_c[m];                  /* fsk.c:721-722 */
    ora_freq_est(f, fsk_in, f_est, M);                               /* fsk.c:725 */
    if (f->f_est[0] < 1)                                             /* fsk.c:750-753 */
        for (m = 0; m < M; m++) f->f_est[m] = f_est[m];

    for (m = 0; m < M; m++) {                                        /* fsk.c:756-764 */
        dphi[m] = c_expj(-2 * (Nmem - nin - (Ts / P)) * M_PI * ((f->f_est[m]) / (float)(Fs)));
        phi_c[m] = c_mul(dphi[m], phi_c[m]);
        dphi[m] = c_expj(2 * M_PI * ((f->f_est[m]) / (float)(Fs)));
    }

    for (m = 0; m < M; m++) {                                        /* fsk.c:767-842 */
        const float f_est_m = f_est[m];
        ora_comp *f_int_m = f->f_int[m];
        ora_comp dphi_m = dphi[m];
        const ora_comp *src = &f->samp_old[nstash - nold];           /* fsk.c:775 */
        int using_old = 1, dc_i, cbuf_i;
        for (dc_i = 0; dc_i < Ts - (Ts / P); dc_i++) {               /* prefill fsk.c:779-799 */
            if (dc_i >= nold && using_old) {
                src = fsk_in; dc_i = 0; using_old = 0;
                phi_c[m] = c_normalize(phi_c[m]);
                dphi_m = c_expj(2 * M_PI * ((f_est_m) / (float)(Fs)));
            }
            f_intbuf[dc_i] = c_mul(src[dc_i], c_conj(phi_c[m]));
            phi_c[m] = c_mul(phi_c[m], dphi_m);
        }
        cbuf_i = dc_i;
        for (i = 0; i < (nsym + 1) * P; i++) {                       /* fsk.c:803-841 */
            float it_r = 0, it_i = 0;
            for (j = 0; j < (Ts / P); j++, dc_i++) {
                if (dc_i >= nold && using_old) {
                    src = fsk_in; dc_i = 0; using_old = 0;
                    phi_c[m] = c_normalize(phi_c[m]);
                    dphi_m = c_expj(2 * M_PI * ((f_est_m) / (float)(Fs)));
                }
                f_intbuf[cbuf_i + j] = c_mul(src[dc_i], c_conj(phi_c[m]));
                phi_c[m] = c_mul(phi_c[m], dphi_m);
            }
            cbuf_i += Ts / P;
            if (cbuf_i >= Ts) cbuf_i = 0;
            for (j = 0; j < Ts; j++) { it_r += f_intbuf[j].real; it_i += f_intbuf[j].imag; }
            f_int_m[i].real = it_r; f_int_m[i].imag = it_i;
        }
    }

    for (m = 0; m < M; m++) { f->phi_c[m] = phi_c[m]; f->f_est[m] = f_est[m]; }  /* fsk.c:845-848 */
    memcpy(f->samp_old, &fsk_in[nin - nstash], sizeof(ora_comp) * nstash);        /* fsk.c:851 */

    /* fine timing fsk.c:858-874 */
    dphift = c_expj(2 * M_PI * ((float)(Rs) / (float)(P * Rs)));
    phi_ft.real = 1; phi_ft.imag = 0;
    t_c.real = 0; t_c.imag = 0;
    for (i = 0; i < (nsym + 1) * P; i++) {
        ft1 = 0;
        for (m = 0; m < M; m++)
            ft1 += (f->f_int[m][i].real * f->f_int[m][i].real) + (f->f_int[m][i].imag * f->f_int[m][i].imag);
        t_c.real = t_c.real + ft1 * phi_ft.real;
        t_c.imag = t_c.imag + ft1 * phi_ft.imag;
        phi_ft = c_mul(phi_ft, dphift);
    }
    if (isnan(t_c.real) || isnan(t_c.imag)) return;                  /* fsk.c:878-880 */

    norm_rx_timing = atan2f(t_c.imag, t_c.real) / (2 * M_PI);        /* fsk.c:883 */
    rx_timing = norm_rx_timing * (float)P;
    old_norm_rx_timing = f->norm_rx_timing;
    f->norm_rx_timing = norm_rx_timing;
    d_norm_rx_timing = norm_rx_timing - old_norm_rx_timing;          /* fsk.c:890 */
    if (fabsf(d_norm_rx_timing) < .2) {                              /* fsk.c:893-896 */
        appm = 1e6 * d_norm_rx_timing / (float)nsym;
        f->ppm = .9 * f->ppm + .1 * appm;
    }
    if (norm_rx_timing > 0.25) f->nin = N + Ts / 2;                  /* fsk.c:900-907 (burst_mode==0) */
    else if (norm_rx_timing < -0.25) f->nin = N - Ts / 2;
    else f->nin = N;

    {                                                                /* fsk.c:913-993 */
        int low_sample = (int)floorf(rx_timing);
        float fract = rx_timing - (float)low_sample;
        int high_sample = (int)ceilf(rx_timing);
        meanebno = 0; stdebno = 0;
        for (i = 0; i < nsym; i++) {
            int st = (i + 1) * P, sym = 0;
            float max, min;
            for (m = 0; m < M; m++) {
                ora_comp a = f->f_int[m][st + low_sample], b = f->f_int[m][st + high_sample];
                t[m].real = (1 - fract) * a.real;  t[m].imag = (1 - fract) * a.imag;
                t[m].real = t[m].real + fract * b.real;
                t[m].imag = t[m].imag + fract * b.imag;
                tmax[m] = (t[m].real * t[m].real) + (t[m].imag * t[m].imag);
            }
            max = tmax[0]; min = tmax[0];
            for (m = 0; m < M; m++) {
                if (tmax[m] > max) { max = tmax[m]; sym = m; }
                if (tmax[m] < min) min = tmax[m];
            }
            if (rx_bits != NULL) {
                if (M == 2) rx_bits[i] = sym == 1;
                else { rx_bits[(i * 2) + 1] = (sym & 0x1); rx_bits[(i * 2)] = (sym & 0x2) >> 1; }
            }
            if (rx_sd != NULL) {
                for (m = 0; m < M; m++) tmax[m] = sqrtf(tmax[m]);
                if (M == 2) rx_sd[i] = tmax[0] - tmax[1];
                else {                                               /* fsk.c:969-980 */
                    rx_sd[(i * 2) + 1] = -tmax[0];
                    rx_sd[(i * 2)]     = -tmax[0];
                    rx_sd[(i * 2) + 1] += tmax[1];
                    rx_sd[(i * 2)]     += -tmax[1];
                    rx_sd[(i * 2) + 1] += -tmax[2];
                    rx_sd[(i * 2)]     += tmax[2];
                    rx_sd[(i * 2) + 1] += tmax[3];
                    rx_sd[(i * 2)]     += tmax[3];
                }
            }
            ft1 = max;                                               /* fsk.c:986-990 */
            stdebno += ft1;
            meanebno += sqrtf(ft1);
        }
        meanebno = meanebno / (float)nsym;                           /* fsk.c:998-1009 */
        stdebno = (stdebno / (float)nsym) - (meanebno * meanebno);
        if (stdebno > 0.0) stdebno = sqrt(stdebno); else stdebno = 0.0;
        f->EbNodB = -6 + (20 * log10f((1e-6 + meanebno) / (1e-6 + stdebno)));

        f->clock_offset = f->ppm;                                    /* fsk.c:1017 */
        f->snr_est = .5 * f->snr_est + .5 * f->EbNodB;               /* fsk.c:1021 */
        f->st_rx_timing = (float)rx_timing;
        fc_avg = (f_est[0] + f_est[1]) / 2;                          /* fsk.c:1027-1029 */
        fc_tx = (f->f1_tx + f->f1_tx + f->fs_tx) / 2;
        f->foff = fc_tx - fc_avg;
        {                                                            /* eye fsk.c:1037-1079 */
            int neyesamp_dec = ceil(((float)P * 2) / ORA_EYE_IND);
            int neyesamp = (P * 2) / neyesamp_dec;
            int neyeoffset = high_sample + 1;
            int eye_traces = ORA_EYE_TR / M;
            int total = (nsym + 1) * P;
            f->neyesamp = neyesamp;
            f->neyetr = M * eye_traces;
            for (i = 0; i < eye_traces; i++)
                for (m = 0; m < M; m++)
                    for (j = 0; j < neyesamp; j++) {
                        int ind = 2 * P * i + neyeoffset + j * neyesamp_dec;
                        /* the reference reads f_int[m][ind] with ind<0 when high_sample<-1
                           (out of bounds, undefined); we substitute 0 there. */
                        f->rx_eye[i * M + m][j] = (ind >= 0 && ind < total) ? c_abs(f->f_int[m][ind]) : 0.0f;
                    }
            eye_max = 0;
            for (i = 0; i < M * eye_traces; i++)
                for (j = 0; j < neyesamp; j++)
                    if (fabsf(f->rx_eye[i][j]) > eye_max) eye_max = fabsf(f->rx_eye[i][j]);
            for (i = 0; i < M * eye_traces; i++)
                for (j = 0; j < neyesamp; j++) f->rx_eye[i][j] = f->rx_eye[i][j] / eye_max;
        }
        for (i = 0; i < M; i++) f->st_f_est[i] = f_est[i];
    }
}

/* ---- sample conversion: fsk_demod.c:273-296 ---------------------------- */
void ora_convert_samples(int fmt, const void *raw, long n, ora_comp *out) {
    long i;
    if (fmt == ORA_FMT_S16_REAL) {
        const int16_t *r = (const int16_t *)raw;
        for (i = 0; i < n; i++) { out[i].real = ((float)r[i]) / 1000; out[i].imag = 0.0; } /* FDMDV_SCALE codec2_fdmdv.h:67 */
    } else if (fmt == ORA_FMT_CU8) {
        const uint8_t *r = (const uint8_t *)raw;
        for (i = 0; i < n; i++) {
            out[i].real = ((float)r[2 * i] - 127.0) / 128.0;
            out[i].imag = ((float)r[2 * i + 1] - 127.0) / 128.0;
        }
    } else if (fmt == ORA_FMT_CS16) {
        const int16_t *r = (const int16_t *)raw;
        for (i = 0; i < n; i++) {
            out[i].real = ((float)r[2 * i]) / 1000;
            out[i].imag = ((float)r[2 * i + 1] / 1000);
        }
    } else {
        memcpy(out, raw, sizeof(ora_comp) * n);
    }
}

long ora_demod_capture(int fmt, const void *raw, long nsamples, int Fs, int Rs, int P, int M,
                       int est_lo, int est_hi, float *sd_out, uint8_t *bits_out, long cap_frames,
                       float *trace) {
    static const int bps[4] = {2, 4, 2, 8};
    ora_fsk *f = (P == -1) ? ora_fsk_create(Fs, Rs, M) : ora_fsk_create_hbr(Fs, Rs, P, M);   /* P == -1: the -l/--lbr geometry (fsk_demod.c:210-212) */
    long off = 0, nframes = 0;
    ora_comp *modbuf;
    float *sdbuf;
    uint8_t *bitbuf;
    if (!f) return -1;
    if (est_lo > 0 && est_hi > est_lo) ora_fsk_set_est_limits(f, est_lo, est_hi);  /* fsk_demod.c:215-218 */
    modbuf = (ora_comp *)malloc(sizeof(ora_comp) * (f->N + f->Ts * 2));
    sdbuf = (float *)calloc(f->Nbits, sizeof(float));
    bitbuf = (uint8_t *)calloc(f->Nbits, 1);
    while (off + f->nin <= nsamples && nframes < cap_frames) {       /* fsk_demod.c:270 */
        int nin = f->nin;
        ora_convert_samples(fmt, (const char *)raw + off * bps[fmt], nin, modbuf);
        ora_fsk_demod_frame(f, bits_out ? bitbuf : NULL, sd_out ? sdbuf : NULL, modbuf);
        if (sd_out) memcpy(sd_out + nframes * f->Nbits, sdbuf, sizeof(float) * f->Nbits);
        if (bits_out) memcpy(bits_out + nframes * f->Nbits, bitbuf, f->Nbits);
        if (trace) {
            float *tr = trace + nframes * 8;
            memcpy(tr, f->f_est, 4 * sizeof(float));
            tr[4] = (float)f->nin; tr[5] = f->norm_rx_timing; tr[6] = f->ppm; tr[7] = f->EbNodB;
        }
        off += nin;
        nframes++;
    }
    free(bitbuf); free(sdbuf); free(modbuf);
    ora_fsk_destroy(f);
    return nframes;
}

/* ======================================================================= */
/* phi0: src/phi0.c:13-218, table-driven                                     */
/* ======================================================================= */
#define SI16(fl) ((int32_t)((fl) * (1 << 16)))                      /* phi0.c:10 */
static const float PHI0_5_10[10] = {   /* phi0.c:19-28, index 19-(x>>15) */
    0.000116589f, 0.000192223f, 0.000316923f, 0.000522517f, 0.000861485f,
    0.001420349f, 0.002341760f, 0.003860913f, 0.006365583f, 0.010495133f};
static const float PHI0_1_5[64] = {    /* phi0.c:35-98, index 79-(x>>12) */
    0.013903889f, 0.014800644f, 0.015755242f, 0.016771414f, 0.017853133f, 0.019004629f, 0.020230403f, 0.021535250f,
    0.022924272f, 0.024402903f, 0.025976926f, 0.027652501f, 0.029436184f, 0.031334956f, 0.033356250f, 0.035507982f,
    0.037798579f, 0.040237016f, 0.042832850f, 0.045596260f, 0.048538086f, 0.051669874f, 0.055003924f, 0.058553339f,
    0.062332076f, 0.066355011f, 0.070637993f, 0.075197917f, 0.080052790f, 0.085221814f, 0.090725463f, 0.096585578f,
    0.102825462f, 0.109469985f, 0.116545700f, 0.124080967f, 0.132106091f, 0.140653466f, 0.149757747f, 0.159456024f,
    0.169788027f, 0.180796343f, 0.192526667f, 0.205028078f, 0.218353351f, 0.232559308f, 0.247707218f, 0.263863255f,
    0.281099022f, 0.299492155f, 0.319127030f, 0.340095582f, 0.362498271f, 0.386445235f, 0.412057648f, 0.439469363f,
    0.468828902f, 0.500301872f, 0.534073947f, 0.570354566f, 0.609381573f, 0.651427083f, 0.696805010f, 0.745880827f};
/* the comparison tree phi0.c:101-213 is a sorted threshold search: value PHI0_LT1_V[k] applies when
   x > PHI0_LT1_T[k] and (k==0 or x <= PHI0_LT1_T[k-1]); below every threshold the result is 10. */
static const float PHI0_LT1_T[27] = {
    0.707107f, 0.500000f, 0.353553f, 0.250000f, 0.176777f, 0.125000f, 0.088388f, 0.062500f, 0.044194f,
    0.031250f, 0.022097f, 0.015625f, 0.011049f, 0.007812f, 0.005524f, 0.003906f, 0.002762f, 0.001953f,
    0.001381f, 0.000977f, 0.000691f, 0.000488f, 0.000345f, 0.000244f, 0.000173f, 0.000122f, 0.000086f};
static const float PHI0_LT1_V[27] = {
    0.922449644f, 1.241248638f, 1.573515241f, 1.912825912f, 2.255740095f, 2.600476919f, 2.946130351f,
    3.292243417f, 3.638586634f, 3.985045009f, 4.331560985f, 4.678105767f, 5.024664952f, 5.371231340f,
    5.717801329f, 6.064373119f, 6.410945809f, 6.757518949f, 7.104092314f, 7.450665792f, 7.797239326f,
    8.143812888f, 8.490386464f, 8.836960047f, 9.183533634f, 9.530107222f, 9.876680812f};

float ora_phi0(float xf) {
    /* (int32_t)(float) on x86-64 is cvttss2si: NaN and out-of-range give INT32_MIN */
    float y = xf * (1 << 16);
    int32_t x = (y >= -2147483648.0f && y < 2147483648.0f) ? (int32_t)y : INT32_MIN;
    int k;
    if (x >= SI16(10.0f)) return 0.0f;
    if (x >= SI16(5.0f)) return PHI0_5_10[19 - (x >> 15)];
    if (x >= SI16(1.0f)) return PHI0_1_5[79 - (x >> 12)];
    for (k = 0; k < 27; k++)
        if (x > SI16(PHI0_LT1_T[k])) return PHI0_LT1_V[k];
    return 10.0f;
}

/* ======================================================================= */
/* LDPC: static Tanner graph of the Wenet code                               */
/*   init_c_v_nodes with H1=1, shift=0 (mpdecode_core.c:152-379 as called    */
/*   from run_ldpc_decoder :519-538 since NumberRowsHcols != CodeLength)     */
/* ======================================================================= */
#define NPAR 516
#define NDATA 2064
#define NCODE 2580
#define ROWW 12
#define MAXCDEG 14
#define NEDGE (13 + 14 * (NPAR - 1))         /* 7223 */

static int g_ready = 0;
static int c_deg[NPAR], c_off[NPAR];         /* edges of check j: c_off[j] .. c_off[j]+c_deg[j]-1, sub order */
static int v_deg[NCODE];
static int v_edge[NCODE][3];                 /* edge ids of variable i in socket order j */

static void graph_build(void) {
    int i, j, e = 0;
    static int e_var[NEDGE];
    if (g_ready) return;
    for (i = 0; i < NPAR; i++) {                                     /* mpdecode_core.c:171-187, 214-236 */
        c_deg[i] = ROWW + (i == 0 ? 1 : 2);
        c_off[i] = e;
        for (j = 0; j < ROWW; j++) e_var[e++] = ORA_HROWS[i * ROWW + j];
        if (i > 0) e_var[e++] = NDATA + i - 1;
        e_var[e++] = NDATA + i;
    }
    /* variable side: data bit i is connected to the checks listed in H_cols in ascending
       order (mpdecode_core.c:343; H_cols == ascending row list, tools/gen_tables.py);
       parity bit i to checks i-2064, i-2064+1 (:334-341), the last one only to 515 (:296-303). */
    for (i = 0; i < NCODE; i++) v_deg[i] = 0;
    for (i = 0; i < NPAR; i++)          /* ascending check order == H_cols order */
        for (j = 0; j < c_deg[i]; j++) {
            int v = e_var[c_off[i] + j];
            v_edge[v][v_deg[v]++] = c_off[i] + j;
        }
    g_ready = 1;
}

int ora_ldpc_decode(const float *llr, int max_iter, uint8_t *bits, int *pcc) {
    /* SumProduct, mpdecode_core.c:385-489.  vmsg/vsign/cmsg live on the edge. */
    static float vmsg[NEDGE], cmsg[NEDGE];
    static uint8_t vsign[NEDGE];
    int i, j, iter, result;
    graph_build();
    for (i = 0; i < NCODE; i++)                                      /* mpdecode_core.c:353-359 */
        for (j = 0; j < v_deg[i]; j++) {
            int e = v_edge[i][j];
            vmsg[e] = ora_phi0(fabs(llr[i]));
            vsign[e] = (llr[i] < 0) ? 1 : 0;
        }
    result = max_iter;
    for (iter = 0; iter < max_iter; iter++) {
        int ssum = 0, any_data_bit = 0;
        for (i = 0; i < NCODE; i++) bits[i] = 0;
        for (j = 0; j < NPAR; j++) {                                 /* update r :414-436 */
            const int e0 = c_off[j];
            int sign = vsign[e0];
            float phi_sum = vmsg[e0];
            for (i = 1; i < c_deg[j]; i++) { phi_sum += vmsg[e0 + i]; sign ^= vsign[e0 + i]; }
            if (sign == 0) ssum++;
            for (i = 0; i < c_deg[j]; i++) {
                if (sign ^ vsign[e0 + i]) cmsg[e0 + i] = -ora_phi0(phi_sum - vmsg[e0 + i]);
                else                      cmsg[e0 + i] =  ora_phi0(phi_sum - vmsg[e0 + i]);
            }
        }
        for (i = 0; i < NCODE; i++) {                                /* update q :439-464 */
            float Qi = llr[i];
            for (j = 0; j < v_deg[i]; j++) Qi += cmsg[v_edge[i][j]];
            if (Qi < 0) bits[i] = 1;
            for (j = 0; j < v_deg[i]; j++) {
                int e = v_edge[i][j];
                float temp_sum = Qi - cmsg[e];
                vmsg[e] = ora_phi0(fabs(temp_sum));
                vsign[e] = (temp_sum > 0) ? 0 : 1;
            }
        }
        for (i = 0; i < NDATA; i++) if (bits[i] != 0) any_data_bit = 1;  /* :467-469 vs all-zero data[] */
        if (!any_data_bit) { result = iter + 1; break; }             /* :473-476 */
        *pcc = ssum;                                                 /* :479 */
        if (ssum == NPAR) { result = iter + 1; break; }              /* :480-483 */
    }
    return result;
}

void ora_ldpc_encode(const uint8_t *ibits, uint8_t *pbits) {        /* mpdecode_core.c:72-91 */
    unsigned p, i, par, prev = 0;
    for (p = 0; p < NPAR; p++) {
        par = 0;
        for (i = 0; i < ROWW; i++) par += ibits[ORA_HROWS[p * ROWW + i]];
        prev = (par + prev) & 1;
        pbits[p] = (uint8_t)prev;
    }
}

void ora_sd_to_llr(float *llr, const double *sd, int n) {           /* mpdecode_core.c:569-595 */
    double sum, mean, sign, sumsq, estvar, estEsN0, x;
    int i;
    sum = 0.0;
    for (i = 0; i < n; i++) sum += fabs(sd[i]);
    mean = sum / n;
    sum = sumsq = 0.0;
    for (i = 0; i < n; i++) {
        sign = (sd[i] > 0.0L) - (sd[i] < 0.0L);
        x = (sd[i] / mean - sign);
        sum += x;
        sumsq += x * x;
    }
    estvar = (n * sumsq - sum * sum) / (n * (n - 1));
    estEsN0 = 1.0 / (2.0L * estvar + 1E-3);     /* long double (x87 80-bit) sub-expression */
    for (i = 0; i < n; i++) llr[i] = 4.0L * estEsN0 * sd[i];   /* long double product -> float */
}

uint16_t ora_crc16(const uint8_t *data, int len) {                  /* drs232_ldpc.c:91-102: CRC-16/CCITT-FALSE */
    uint16_t crc = 0xFFFF;
    int i, b;
    for (i = 0; i < len; i++) {
        crc ^= (uint16_t)data[i] << 8;
        for (b = 0; b < 8; b++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
    }
    return crc;
}

/* ======================================================================= */
/* deframer + decode: main() of drs232_ldpc.c:171-274 (mode 1) and            */
/* wenet_ldpc.c:166-258 (mode 2)                                             */
/* ======================================================================= */
long ora_deframe_decode(int mode, const float *sd, long nsym, int max_iter, long cap,
                        long *pkt_start, int *pkt_iter, uint8_t *pkt_crc_ok,
                        uint8_t *pkt_bytes, float *llr_dump) {
    /* UW as a bit mask, oldest bit = most significant (drs232_ldpc.c:77-86, wenet_ldpc.c:77-82) */
    static const uint8_t uw1[40] = {0,1,1,0,1,0,1,0,1,1, 0,1,0,1,1,0,0,1,1,1, 0,1,1,1,1,0,1,1,1,1, 0,1,0,0,0,0,0,0,0,1};
    static const uint8_t uw2[32] = {1,0,1,0,1,0,1,1, 1,1,0,0,1,1,0,1, 1,1,1,0,1,1,1,1, 0,0,0,0,0,0,0,1};
    const int uw_bits = (mode == 1) ? 40 : 32;
    const int allowed = (mode == 1) ? 5 : 4;
    const int bits_per_byte = (mode == 1) ? 10 : 8;
    const int spp = (256 + 2 + 65) * bits_per_byte;                  /* SYMBOLS_PER_PACKET */
    const uint8_t *uw = (mode == 1) ? uw1 : uw2;
    uint64_t uwmask = 0, window = 0, wmask = (uw_bits == 64) ? ~0ULL : ((1ULL << uw_bits) - 1);
    double *symbol_buf = (double *)malloc(sizeof(double) * spp);
    double *nors = (double *)malloc(sizeof(double) * spp);
    float *llr = (float *)malloc(sizeof(float) * spp);
    uint8_t bits[NCODE];
    long s, npk = 0, start = 0;
    int state = 0, ind = 0, i, j, k;
    for (i = 0; i < uw_bits; i++) uwmask = (uwmask << 1) | uw[i];
    for (s = 0; s < nsym; s++) {
        float symbol = sd[s];
        int bit = symbol < 0;
        int next_state = state;
        if (state == 0) {                                            /* LOOK_FOR_UW :183-209 */
            window = ((window << 1) | (uint64_t)bit) & wmask;
            int score = uw_bits - __builtin_popcountll((window ^ uwmask) & wmask);
            if (score >= uw_bits - allowed) { ind = 0; next_state = 1; start = s + 1; }
        }
        if (state == 1) {                                            /* COLLECT_PACKET :211-268 */
            if (mode == 2) {                                         /* wenet_ldpc.c:207 */
                int kbit = ind % 1000;
                double code = ((ORA_SCRAMBLE[kbit >> 3] >> (7 - (kbit & 7))) & 1) ? -1.0 : 1.0;
                symbol_buf[ind] = symbol * code;
            } else symbol_buf[ind] = symbol;
            ind++;
            if (ind == spp) {
                const double *dec_in = symbol_buf;
                int iter, pcc = 0;
                uint8_t packet[258];
                uint16_t rx, tx;
                if (mode == 1) {                                     /* drs232_ldpc.c:220-225 */
                    for (i = 0, k = 0; i < spp; i += 10) {
                        for (j = 0; j < 8; j++) nors[k + j] = symbol_buf[i + 7 - j + 1];
                        k += 8;
                    }
                    dec_in = nors;
                }
                ora_sd_to_llr(llr, dec_in, NCODE);
                iter = ora_ldpc_decode(llr, max_iter, bits, &pcc);
                for (i = 0; i < 258; i++) {                          /* :234-239 */
                    uint8_t a = 0;
                    for (j = 0; j < 8; j++) a |= bits[8 * i + j] << (7 - j);
                    packet[i] = a;
                }
                rx = ora_crc16(packet, 256);
                tx = packet[256] + (packet[257] << 8);
                if (npk < cap) {
                    if (pkt_start) pkt_start[npk] = start;
                    if (pkt_iter) pkt_iter[npk] = iter;
                    if (pkt_crc_ok) pkt_crc_ok[npk] = (rx == tx);
                    if (pkt_bytes) memcpy(pkt_bytes + npk * 258, packet, 258);
                    if (llr_dump) memcpy(llr_dump + npk * NCODE, llr, sizeof(float) * NCODE);
                }
                npk++;
                next_state = 0;
            }
        }
        state = next_state;
    }
    free(llr); free(nors); free(symbol_buf);
    return npk < cap ? npk : cap;
}
