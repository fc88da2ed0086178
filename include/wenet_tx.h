/*
 * wenet_tx.h -- C ABI of the batched Wenet frame builder / test-signal generator in libwenet_rx.so
 * (the callers' data format on the INPUT side of the receive path).
 *
 * The reference transmitter is Python + one C helper driving radio hardware; nothing of it runs on the
 * receive hot path.  What the receive path depends on is the on-air FORMAT, and this header builds it on
 * the GPU so that large batches of synthetic captures never leave HBM:
 *
 *   frame layout   tx/PacketTX.py:65-66,123-137   16 x 0x55, unique word 0xABCDEF01, 256-byte payload,
 *                                                 CRC-16/CCITT-FALSE little-endian, 65 parity bytes
 *   RA encoder     tx/ldpc_enc.c:33-48 + tx/Hrow2064.txt  (== src/mpdecode_core.c:72-91)
 *   v2 scramble    tx/radio_wrappers.py:385-405   XOR of payload+crc+parity with the 125-byte code
 *   v2 bit order   tx/radio_wrappers.py:407-417   MSB first
 *   v1 bit order   tx/radio_wrappers.py:553-560   RS-232: start 0, 8 data bits LSB first, stop 1
 *   tone keying    bit 1 = upper tone; 4-FSK keys tone 3-(b0<<1|b1) so that the reference's 4-FSK soft
 *                  decisions (src/fsk.c:969-980) come out with the polarity drs232_ldpc expects
 *   noise model    benchmarking/generate_lowsnr.py:70-89  sigma^2 = var(x) Fs/(Rs EbN0 bits_per_symbol),
 *                  complex Gaussian, then division by max|x|
 *   cu8            csdr convert_f_u8 restated as (uint8)(x*127.5+128) (external tool, parity unpinned)
 *
 * The modulator itself has no reference implementation on the Wenet path (the radio chip does it); it
 * is a continuous-phase M-FSK phase accumulator (32-bit phase), deterministic given the seed.
 * Plain pointers and sizes only.  A GPU is mandatory.
 */
#ifndef WENET_TX_H
#define WENET_TX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wenet_tx wenet_tx;

/* framing: 1 = v1/RS-232 (drs232_ldpc), 2 = v2/I2S (wenet_ldpc).  Tone m is at f_low + m*f_space Hz.
 * Fs/Rs must be an integer (src/fsk.c:143).  NULL on illegal parameters or without a GPU. */
wenet_tx *wenet_tx_create(int Fs, int Rs, int M, int framing, double f_low, double f_space);
void wenet_tx_destroy(wenet_tx *tx);

/* symbols one framed packet occupies on air: 343 bytes x 10 (v1) or x 8 (v2) bits, halved for 4-FSK */
long long wenet_tx_symbols_per_packet(const wenet_tx *tx);

/* payloads: npackets x 256 bytes -> symbols: npackets x symbols_per_packet tone indices (one byte each,
 * frames back to back).  device != 0: both pointers are DEVICE pointers and the call only enqueues on
 * `stream` (hipStream_t or NULL); otherwise host pointers, synchronous.  Returns 0, <0 on error. */
int wenet_tx_frame_packets(wenet_tx *tx, const uint8_t *payloads, long long npackets, uint8_t *symbols,
                           int device, void *stream);

/* Modulate ncap independent symbol streams into IQ captures resident in HBM.
 *   symbols[c]  DEVICE pointer to nsym[c] tone indices        iq_out[c]  DEVICE pointer, nsym[c]*Ts samples
 *   ebno_db[c]  Eb/N0 of the added noise (>= 200 : no noise)  ppm[c]     transmitter symbol-clock error
 *   seed[c]     noise seed (Philox-4x32-10 counter-based)     fmt        WENET_FMT_CU8 (2) or WENET_FMT_CS16 (1)
 * Sample n carries symbol min(floor(n (1+ppm 1e-6)/Ts), nsym-1); the capture is divided by its own
 * max|x| before conversion (generate_lowsnr.py:85-87).  Enqueues on `stream`; returns 0, <0 on error. */
int wenet_tx_modulate(wenet_tx *tx, int ncap, const uint8_t *const *symbols, const long long *nsym,
                      const double *ebno_db, const double *ppm, const uint64_t *seed, int fmt,
                      void *const *iq_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
