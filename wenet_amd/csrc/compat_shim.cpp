// compat_shim.cpp -- libwenet_fsk_compat.so: the reference's link names and structure layouts (include/wenet_fsk_compat.h) forwarded to libwenet_rx.so.
// Host code only, no arithmetic of the receive path: every number comes out of the wenet_* calls.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/wenet_fsk_compat.h"
#include "../../include/wenet_rx.h"

namespace {
struct Priv { wenet_fsk *h; };
wenet_fsk *handle(struct FSK *f) { return ((Priv *)f->wenet_private)->h; }

struct FSK *wrap(wenet_fsk *h, int tx_f1, int tx_fs) {
    if (!h) return nullptr;
    struct FSK *f = (struct FSK *)calloc(1, sizeof(struct FSK));
    Priv *p = (Priv *)calloc(1, sizeof(Priv));
    struct MODEM_STATS *st = (struct MODEM_STATS *)calloc(1, sizeof(struct MODEM_STATS));
    if (!f || !p || !st) { free(f); free(p); free(st); wenet_fsk_destroy(h); return nullptr; }
    p->h = h;
    f->wenet_private = p;
    f->Ndft = wenet_fsk_info(h, 0); f->N = wenet_fsk_info(h, 1); f->Ts = wenet_fsk_info(h, 2); f->Nmem = wenet_fsk_info(h, 3); f->P = wenet_fsk_info(h, 4);
    f->Nsym = wenet_fsk_info(h, 5); f->Nbits = wenet_fsk_info(h, 6); f->nstash = wenet_fsk_info(h, 7); f->mode = wenet_fsk_info(h, 8);
    f->est_min = wenet_fsk_info(h, 9); f->est_max = wenet_fsk_info(h, 10); f->est_space = wenet_fsk_info(h, 11); f->Fs = wenet_fsk_info(h, 12); f->Rs = wenet_fsk_info(h, 13);
    f->f1_tx = tx_f1; f->fs_tx = tx_fs;
    for (int m = 0; m < 4; m++) { f->phi_c[m].real = 1.f; f->phi_c[m].imag = 0.f; }        // fsk.c:182-185 (the live phasors are on the GPU)
    f->tx_phase_c.real = 1.f;
    f->fft_est = (float *)calloc((size_t)f->Ndft / 2, sizeof(float));                      // fsk.c:222-233: zeros until the first frame
    f->nin = (int)wenet_fsk_nin(h);
    f->normalise_eye = 1;                                                                  // fsk.c:256
    f->stats = st;
    {   // fsk.c:405-432 (stats_init): eye geometry known from the start, traces zero
        const int eye_dec = (f->P * 2 + 159) / 160;
        st->neyesamp = (f->P * 2) / (eye_dec > 0 ? eye_dec : 1);
        st->neyetr = f->mode * (8 / f->mode);
    }
    wenet_fsk_enable_stats(h, 0, 1);                                                       // a snapshot per frame: the members below follow every call
    return f;
}
void refresh(struct FSK *f) {
    wenet_fsk *h = handle(f);
    wenet_modem_stats s;
    wenet_fsk_get_demod_stats(h, &s);
    f->nin = (int)wenet_fsk_nin(h);
    f->ppm = s.ppm;
    f->EbNodB = wenet_fsk_last_ebnodb(h);
    for (int m = 0; m < 4; m++) f->f_est[m] = s.f_est[m];
    if (s.nfft_est > 0) memcpy(f->fft_est, s.fft_est, sizeof(float) * (size_t)s.nfft_est);
    struct MODEM_STATS *st = f->stats;
    st->snr_est = s.snr_est; st->rx_timing = s.rx_timing; st->foff = s.foff; st->clock_offset = s.ppm;                 // fsk.c:1017-1029
    if (s.neyetr > 0) { st->neyetr = s.neyetr; st->neyesamp = s.neyesamp; memcpy(st->rx_eye, s.rx_eye, sizeof(st->rx_eye)); }
    for (int m = 0; m < f->mode && m < 4; m++) st->f_est[m] = s.f_est[m];                                               // fsk.c:1084-1086
    st->nr = 0; st->Nc = 0;
}
}  // namespace

extern "C" {

struct FSK *fsk_create_hbr(int Fs, int Rs, int P, int M, int tx_f1, int tx_fs) {
    struct FSK *f = wrap(wenet_fsk_create_hbr(Fs, Rs, P, M, tx_f1, tx_fs), tx_f1, tx_fs);
    if (!f) { fprintf(stderr, "fsk_create_hbr: illegal parameters or no GPU (the reference asserts here, src/fsk.c:137-146)\n"); abort(); }
    return f;
}
struct FSK *fsk_create(int Fs, int Rs, int M, int tx_f1, int tx_fs) {
    struct FSK *f = wrap(wenet_fsk_create(Fs, Rs, M, tx_f1, tx_fs), tx_f1, tx_fs);
    if (!f) { fprintf(stderr, "fsk_create: illegal parameters or no GPU (the reference asserts here, src/fsk.c:286-295)\n"); abort(); }
    return f;
}
void fsk_destroy(struct FSK *f) {
    if (!f) return;
    wenet_fsk_destroy(handle(f));
    free(f->wenet_private); free(f->fft_est); free(f->stats); free(f);
}
void fsk_set_est_limits(struct FSK *f, int fmin, int fmax) {
    wenet_fsk_set_est_limits(handle(f), fmin, fmax);
    f->est_min = wenet_fsk_info(handle(f), 9); f->est_max = wenet_fsk_info(handle(f), 10);
}
uint32_t fsk_nin(struct FSK *f) { return wenet_fsk_nin(handle(f)); }
void fsk_demod(struct FSK *f, uint8_t rx_bits[], COMP fsk_in[]) { wenet_fsk_demod(handle(f), rx_bits, (const wenet_comp *)fsk_in); refresh(f); }
void fsk_demod_sd(struct FSK *f, float rx_sd[], COMP fsk_in[]) { wenet_fsk_demod_sd(handle(f), rx_sd, (const wenet_comp *)fsk_in); refresh(f); }
void fsk_get_demod_stats(struct FSK *f, struct MODEM_STATS *stats) {              // the copy fsk.c:496-517 makes
    const struct MODEM_STATS *s = f->stats;
    stats->clock_offset = s->clock_offset; stats->snr_est = s->snr_est; stats->rx_timing = s->rx_timing; stats->foff = s->foff;
    stats->neyesamp = s->neyesamp; stats->neyetr = s->neyetr;
    memcpy(stats->rx_eye, s->rx_eye, sizeof(stats->rx_eye));
    memcpy(stats->f_est, s->f_est, (size_t)f->mode * sizeof(float));
    stats->sync = 0; stats->nr = s->nr; stats->Nc = s->Nc;
}
void fsk_stats_normalise_eye(struct FSK *f, int normalise_enable) {
    if (!normalise_enable) fprintf(stderr, "fsk_stats_normalise_eye(0): only the normalised eye diagram is provided\n");
    f->normalise_eye = 1;
}

int run_ldpc_decoder(struct LDPC *ldpc, uint8_t out_char[], float input[], int *parityCheckCount) {
    static_assert(sizeof(struct LDPC) == sizeof(struct wenet_ldpc), "struct LDPC layout");
    return wenet_run_ldpc_decoder((struct wenet_ldpc *)ldpc, out_char, input, parityCheckCount);
}
void sd_to_llr(float llr[], double sd[], int n) { wenet_sd_to_llr(llr, sd, n); }

}  // extern "C"
