/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY (never shipped, never linked into the product).
 *
 * A tiny accessor layer that is compiled TOGETHER WITH the unmodified reference
 * sources (where they lie under $(REF), see oracle/Makefile) into
 * oracle/_ref/libwenet_ref.so.  It exposes the pieces of reference state that
 * have no public getter, so that tests can pin oracle/wenet_oracle.c and the
 * HIP kernels against the reference itself:
 *
 *   - struct FSK fields carried between frames      (src/fsk.h:43-90)
 *   - the LDPC code tables and the embedded known-answer vector
 *     (src/H2064_516_sparse.h:9-33)
 *   - a run_ldpc_decoder() wrapper that fills struct LDPC the way the CLI mains do
 *     (src/drs232_ldpc.c:128-138)
 *
 * No reference code is copied here: the reference headers are #included from
 * $(REF) at build time.
 */
#include <stdint.h>
#include <string.h>

#include "fsk.h"
#include "mpdecode_core.h"
#include "H2064_516_sparse.h"   /* defines H_rows, H_cols, input, detected_data */
#include "wenet_scramble.h"      /* defines scramble_code[1000] (src/wenet_scramble.h:22-146) */

extern float phi0(float xf);

/* ---- FSK state getters ------------------------------------------------- */
int   ref_fsk_Ndft(struct FSK *f)            { return f->Ndft; }
int   ref_fsk_N(struct FSK *f)               { return f->N; }
int   ref_fsk_Ts(struct FSK *f)              { return f->Ts; }
int   ref_fsk_Nmem(struct FSK *f)            { return f->Nmem; }
int   ref_fsk_P(struct FSK *f)               { return f->P; }
int   ref_fsk_Nsym(struct FSK *f)            { return f->Nsym; }
int   ref_fsk_Nbits(struct FSK *f)           { return f->Nbits; }
int   ref_fsk_nstash(struct FSK *f)          { return f->nstash; }
int   ref_fsk_mode(struct FSK *f)            { return f->mode; }
int   ref_fsk_est_min(struct FSK *f)         { return f->est_min; }
int   ref_fsk_est_max(struct FSK *f)         { return f->est_max; }
int   ref_fsk_est_space(struct FSK *f)       { return f->est_space; }
int   ref_fsk_nin_field(struct FSK *f)       { return f->nin; }
float ref_fsk_norm_rx_timing(struct FSK *f)  { return f->norm_rx_timing; }
float ref_fsk_ppm(struct FSK *f)             { return f->ppm; }
float ref_fsk_EbNodB(struct FSK *f)          { return f->EbNodB; }
void  ref_fsk_f_est(struct FSK *f, float out[4])   { memcpy(out, f->f_est, 4*sizeof(float)); }
void  ref_fsk_phi_c(struct FSK *f, float out[8])   { memcpy(out, f->phi_c, 8*sizeof(float)); }
void  ref_fsk_fft_est(struct FSK *f, float *out)   { memcpy(out, f->fft_est, sizeof(float)*f->Ndft/2); }
void  ref_fsk_hann(struct FSK *f, float *out)      { memcpy(out, f->hann_table, sizeof(float)*f->Ndft); }
void  ref_fsk_samp_old(struct FSK *f, float *out)  { memcpy(out, f->samp_old, sizeof(COMP)*f->nstash); }
float ref_fsk_snr_est(struct FSK *f)         { return f->stats->snr_est; }
float ref_fsk_stats_rx_timing(struct FSK *f) { return f->stats->rx_timing; }
float ref_fsk_foff(struct FSK *f)            { return f->stats->foff; }
int   ref_fsk_neyesamp(struct FSK *f)        { return f->stats->neyesamp; }
int   ref_fsk_neyetr(struct FSK *f)          { return f->stats->neyetr; }
void  ref_fsk_rx_eye(struct FSK *f, float *out) {
    memcpy(out, f->stats->rx_eye, sizeof(f->stats->rx_eye));
}

/* ---- LDPC code tables + KAT ------------------------------------------- */
int   ref_ldpc_codelength(void)      { return CODELENGTH; }
int   ref_ldpc_nparity(void)         { return NUMBERPARITYBITS; }
int   ref_ldpc_nrows_hcols(void)     { return NUMBERROWSHCOLS; }
int   ref_ldpc_max_row_weight(void)  { return MAX_ROW_WEIGHT; }
int   ref_ldpc_max_col_weight(void)  { return MAX_COL_WEIGHT; }
int   ref_ldpc_max_iter(void)        { return MAX_ITER; }
const uint16_t *ref_ldpc_H_rows(void){ return H_rows; }
const uint16_t *ref_ldpc_H_cols(void){ return H_cols; }
int   ref_ldpc_H_rows_len(void)      { return (int)(sizeof(H_rows)/sizeof(H_rows[0])); }
int   ref_ldpc_H_cols_len(void)      { return (int)(sizeof(H_cols)/sizeof(H_cols[0])); }
void  ref_ldpc_kat(float *llr_out, uint8_t *bits_out) {
    int i;
    for (i = 0; i < CODELENGTH; i++) { llr_out[i] = (float)input[i]; bits_out[i] = (uint8_t)detected_data[i]; }
}

/* run_ldpc_decoder with the struct LDPC the CLI mains build (drs232_ldpc.c:128-138),
   max_iter overridable (BASELINE config 4 asks for 50). */
int ref_ldpc_decode(float *llr, int max_iter, uint8_t *out_bits, int *parityCheckCount) {
    struct LDPC ldpc;
    memset(&ldpc, 0, sizeof(ldpc));
    ldpc.max_iter = max_iter;
    ldpc.dec_type = 0;
    ldpc.q_scale_factor = 1;
    ldpc.r_scale_factor = 1;
    ldpc.CodeLength = CODELENGTH;
    ldpc.NumberParityBits = NUMBERPARITYBITS;
    ldpc.NumberRowsHcols = NUMBERROWSHCOLS;
    ldpc.max_row_weight = MAX_ROW_WEIGHT;
    ldpc.max_col_weight = MAX_COL_WEIGHT;
    ldpc.H_rows = H_rows;
    ldpc.H_cols = H_cols;
    return run_ldpc_decoder(&ldpc, out_bits, llr, parityCheckCount);
}

float ref_phi0(float x) { return phi0(x); }

int ref_scramble_len(void) { return (int)(sizeof(scramble_code)/sizeof(scramble_code[0])); }
const double *ref_scramble_code(void) { return scramble_code; }
