/*
 * wenet_fsk_compat.h -- libwenet_fsk_compat.so: the reference's OWN link names and structure layouts over libwenet_rx.so, so that a C caller written
 * against /root/reference/src/fsk.h, modem_stats.h and mpdecode_core.h links and runs unchanged.  The proof is the reference's own
 * mains: src/fsk_demod.c, src/drs232_ldpc.c and src/wenet_ldpc.c compiled UNMODIFIED against the reference headers and linked with this library
 * instead of fsk.c / kiss_fft.c / mpdecode_core.c / phi0.c (oracle/Makefile `make ref_on_shim`; tests/test_gpu_compat.py: stdout byte-identical).
 *
 * The declarations below RESTATE the reference's interface (they are not a copy of its headers): same names, argument meaning, structure member order and
 * types.  A caller includes either this header or the reference's own -- never both (the names are the same on purpose).
 * One launch per modem frame: this is the compatibility path, not the fast one (wenet_fsk_demod_stream / wenet_rx_* in wenet_rx.h are).
 */
#ifndef WENET_FSK_COMPAT_H
#define WENET_FSK_COMPAT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float real, imag; } COMP;                          /* src/comp.h:33-36 */

/* src/modem_stats.h:38-72 (member order and sizes; the FFT configuration is an opaque pointer here) */
struct MODEM_STATS {
    int   Nc;
    float snr_est;
    COMP  rx_symbols[8][21];
    int   nr, sync;
    float foff, rx_timing, clock_offset, sync_metric;
    float rx_eye[8][160];
    int   neyetr, neyesamp;
    float f_est[4];
    float fft_buf[2 * 512];
    void *fft_cfg;
};

/* src/fsk.h:42-90: the members callers read (src/fsk_demod.c:255-260,354-382) are kept current after every fsk_demod / fsk_demod_sd; hann_table, fft_cfg
 * and samp_old are NULL (the state they stand for lives on the GPU); what follows `normalise_eye` is private to the library. */
struct FSK {
    int Ndft, Fs, N, Rs, Ts, Nmem, P, Nsym, Nbits, f1_tx, fs_tx, mode, est_min, est_max, est_space;
    float *hann_table;
    COMP phi_c[4];
    void *fft_cfg;
    float norm_rx_timing;
    COMP *samp_old;
    int nstash;
    float *fft_est;
    COMP tx_phase_c;
    float EbNodB;
    float f_est[4];
    float ppm;
    int nin;
    int burst_mode;
    struct MODEM_STATS *stats;
    int normalise_eye;
    void *wenet_private;
};

struct FSK *fsk_create(int Fs, int Rs, int M, int tx_f1, int tx_fs);                 /* src/fsk.h:100 */
struct FSK *fsk_create_hbr(int Fs, int Rs, int P, int M, int tx_f1, int tx_fs);      /* src/fsk.h:110; illegal parameters abort as the reference's asserts do */
void fsk_set_est_limits(struct FSK *fsk, int fmin, int fmax);                         /* src/fsk.h:120 */
void fsk_get_demod_stats(struct FSK *fsk, struct MODEM_STATS *stats);                 /* src/fsk.h:130 */
void fsk_destroy(struct FSK *fsk);                                                    /* src/fsk.h:137 */
uint32_t fsk_nin(struct FSK *fsk);                                                    /* src/fsk.h:173 */
void fsk_demod(struct FSK *fsk, uint8_t rx_bits[], COMP fsk_in[]);                    /* src/fsk.h:184 */
void fsk_demod_sd(struct FSK *fsk, float rx_sd[], COMP fsk_in[]);                     /* src/fsk.h:194 */
void fsk_stats_normalise_eye(struct FSK *fsk, int normalise_enable);                  /* src/fsk.h:198 (only the normalised form exists here: 0 is refused with a message) */
/* not on the receive path, not provided (a caller that needs them fails at link time): fsk_mod, fsk_mod_c, fsk_mod_ext_vco, fsk_set_nsym, fsk_enable_burst_mode, fsk_clear_estimators */

/* src/mpdecode_core.h:18-39 */
struct LDPC {
    int max_iter, dec_type, q_scale_factor, r_scale_factor, CodeLength, NumberParityBits, NumberRowsHcols, max_row_weight, max_col_weight,
        data_bits_per_frame, coded_bits_per_frame, coded_syms_per_frame;
    uint16_t *H_rows, *H_cols;
};
int run_ldpc_decoder(struct LDPC *ldpc, uint8_t out_char[], float input[], int *parityCheckCount);
void sd_to_llr(float llr[], double sd[], int n);

#ifdef __cplusplus
}
#endif
#endif
