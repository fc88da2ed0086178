"""wenet_amd -- MI355X-native receive hot path of Wenet (fsk_demod | drs232_ldpc / wenet_ldpc).

The product is wenet_amd/libwenet_rx.so (hand-written gfx950 kernels behind the C ABI of
include/wenet_rx.h) plus the drop-in executables in wenet_amd/bin/.  The modules here are the thin
host-side mirror of the reference's interfaces:

    wenet_amd.fsk   -- fsk_create_hbr / fsk_nin / fsk_demod_sd ...   (src/fsk.h)
    wenet_amd.ldpc  -- run_ldpc_decoder / sd_to_llr / deframer       (src/mpdecode_core.h, drs232_ldpc.c)
    wenet_amd.rx    -- batch chain: many captures -> packets
    wenet_amd.siggen-- synthetic Wenet transmit signals for tests and benchmarks
"""
