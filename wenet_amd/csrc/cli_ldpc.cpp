// cli_ldpc.cpp -- drop-in replacements for the reference's `drs232_ldpc` (WENET_FRAMING=1) and
// `wenet_ldpc` (WENET_FRAMING=2) executables (src/drs232_ldpc.c:105-285, src/wenet_ldpc.c): same argv,
// float32 symbols in, CRC-valid 256-byte packets out with a flush after each, same stderr lines.
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "../../include/wenet_rx.h"

#ifndef WENET_FRAMING
#define WENET_FRAMING 1
#endif

static unsigned short gen_crc16(const unsigned char *p, int length) {    /* CRC-16/CCITT-FALSE, for the -vv message only */
    unsigned short crc = 0xFFFF;
    while (length--) {
        crc ^= (unsigned short)(*p++) << 8;
        for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? (unsigned short)((crc << 1) ^ 0x1021) : (unsigned short)(crc << 1);
    }
    return crc;
}

int main(int argc, char *argv[]) {
    FILE *fin, *fout;
    int verbose = 0;
    uint16_t packet_errors = 0, packets = 0;                                /* uint16_t as the reference (wraps at 65536) */
    if (argc < 3) {                                                          /* drs232_ldpc.c:142-145 */
        fprintf(stderr, "usage: drs232 InputOneSymbolPerFloat OutputPackets [-v[v]]\n");
        exit(1);
    }
    if (strcmp(argv[1], "-") == 0) fin = stdin;
    else if ((fin = fopen(argv[1], "rb")) == NULL) {
        fprintf(stderr, "Error opening input file: %s: %s.\n", argv[1], strerror(errno));
        exit(1);
    }
    if (strcmp(argv[2], "-") == 0) fout = stdout;
    else if ((fout = fopen(argv[2], "wb")) == NULL) {
        fprintf(stderr, "Error opening output file: %s: %s.\n", argv[2], strerror(errno));
        exit(1);
    }
    if (argc > 3) {
        if (strcmp(argv[3], "-v") == 0) verbose = 1;
        if (strcmp(argv[3], "-vv") == 0) verbose = 2;
    }
    wenet_deframer *d = wenet_deframer_create(WENET_FRAMING, 10 /* MAX_ITER, src/H2064_516_sparse.h:15 */);
    if (!d) { fprintf(stderr, "wenet_rx: no GPU available\n"); exit(1); }

    // One GPU call per block.  A pipe is drained: the first read blocks (live streams keep their latency), then
    // whatever else is already waiting is taken along, so that a fast upstream (a file through cat) gives big blocks.
    const size_t block = (1u << 20);                                         /* symbols per block, at most */
    std::vector<float> buf(block);
    std::vector<uint8_t> pk(((block / 2584) + 4) * 258);
    std::vector<wenet_packet_info> info((block / 2584) + 4);
    const int fd = fileno(fin);
#ifdef F_SETPIPE_SZ
    (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);                                  // a pipe: let the upstream run 1 MiB ahead (bigger blocks per GPU call)
#endif
    size_t partial = 0;                                                      /* bytes of an incomplete float */
    while (true) {
        ssize_t got = read(fd, (char *)buf.data() + partial, block * sizeof(float) - partial);
        if (got < 0) { if (errno == EINTR) continue; break; }
        if (got == 0) break;
        size_t bytes = partial + (size_t)got;
        while (bytes < block * sizeof(float)) {                              /* drain what is already there */
            struct pollfd pf = {fd, POLLIN, 0};
            if (poll(&pf, 1, 0) <= 0 || !(pf.revents & POLLIN)) break;
            ssize_t more = read(fd, (char *)buf.data() + bytes, block * sizeof(float) - bytes);
            if (more <= 0) break;
            bytes += (size_t)more;
        }
        const size_t nsym = bytes / sizeof(float);
        long n = wenet_deframer_push(d, buf.data(), (long)nsym, pk.data(), info.data(), (long)info.size());
        if (n < 0) { fprintf(stderr, "wenet_rx: GPU decode failed (%ld)\n", n); exit(1); }
        for (long i = 0; i < n; i++) {
            const uint8_t *packet = &pk[(size_t)i * 258];
            packets++;
            if (verbose == 2 && !info[i].crc_ok) {                           /* drs232_ldpc.c:246-251 */
                unsigned rx_checksum = gen_crc16(packet, 256), tx_checksum = packet[256] + (packet[257] << 8);
                fprintf(stderr, "tx_checksum: 0x%02x rx_checksum: 0x%02x\n", tx_checksum, rx_checksum);
            }
            if (info[i].crc_ok) {
                fwrite(packet, sizeof(char), 256, fout);                      /* drs232_ldpc.c:254-257 */
                fflush(fout);
            } else packet_errors++;
            if (verbose)
                fprintf(stderr, "packets: %d packet_errors: %d PER: %4.3f iter: %d\n", packets, packet_errors,
                        (float)packet_errors / packets, info[i].iter);
        }
        partial = bytes - nsym * sizeof(float);
        if (partial) memmove(buf.data(), (char *)buf.data() + nsym * sizeof(float), partial);
    }
    fclose(fin);
    fclose(fout);
    fprintf(stderr, "packets: %d packet_errors: %d PER: %4.3f\n", packets, packet_errors, (float)packet_errors / packets);
    wenet_deframer_destroy(d);
    return 0;
}
