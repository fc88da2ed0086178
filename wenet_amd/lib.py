"""ctypes binding of libwenet_rx.so (include/wenet_rx.h).

The shared library is the product; this module only loads it and declares prototypes.
There is no Python or CPU implementation behind these calls: if the library is missing the
import fails loudly, and if no GPU is present the create functions return NULL.

Process-level note: PyTorch wheels bundle their own libamdhip64.  In a process that uses BOTH torch and
this library (bench.py, some tests), import torch FIRST so that one HIP runtime serves both; loading
libwenet_rx.so first binds the system runtime and torch then reports "no ROCm-capable device".
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WENET_RX_LIB") or os.path.join(_HERE, "libwenet_rx.so")     # (WENET_RX_LIB: a development build, e.g. tools/prof_build.sh)

# every symbol include/wenet_rx.h declares
EXPORTS = [
    "wenet_fsk_create_hbr", "wenet_fsk_create", "wenet_fsk_destroy", "wenet_fsk_set_est_limits", "wenet_fsk_nin",
    "wenet_fsk_demod", "wenet_fsk_demod_sd", "wenet_fsk_info", "wenet_fsk_demod_stream",
    "wenet_fsk_enable_stats", "wenet_fsk_get_stats", "wenet_fsk_get_demod_stats",
    "wenet_run_ldpc_decoder", "wenet_sd_to_llr", "wenet_ldpc_decode_batch",
    "wenet_deframer_create", "wenet_deframer_destroy", "wenet_deframer_push",
    "wenet_rx_create", "wenet_rx_destroy", "wenet_rx_process", "wenet_rx_enqueue", "wenet_rx_collect",
    "wenet_rx_frames", "wenet_rx_packets", "wenet_rx_get_packets", "wenet_rx_packet_census", "wenet_rx_get_soft",
    "wenet_rx_enable_trace", "wenet_rx_get_trace", "wenet_rx_enable_llr_dump", "wenet_rx_get_llrs",
    "wenet_rx_last_ms", "wenet_rx_device_info", "wenet_rx_version", "wenet_rx_last_kernel", "wenet_rx_get_device", "wenet_rx_channel_counter", "wenet_rx_set_cf32_quantise",
    "wenet_packet_type_class", "wenet_ssdv_packet_info", "wenet_rx_get_packets_of_class", "wenet_rx_ssdv_images",
    "wenet_phi0_eval", "wenet_rx_source_id", "wenet_rx_decoder_repeats", "wenet_rx_result_digest", "wenet_rx_push", "wenet_rx_flush", "wenet_rx_live_gathered", "wenet_rx_pin_host", "wenet_rx_unpin_host", "wenet_fsk_last_ebnodb",
]
# every symbol include/wenet_tx.h declares
EXPORTS_TX = [
    "wenet_tx_create", "wenet_tx_destroy", "wenet_tx_symbols_per_packet", "wenet_tx_frame_packets",
    "wenet_tx_modulate",
]


class PacketInfo(C.Structure):
    _fields_ = [("iter", C.c_int), ("crc_ok", C.c_int), ("start_symbol", C.c_longlong)]


class SsdvInfo(C.Structure):            # wenet_ssdv_info
    _fields_ = [("callsign", C.c_char * 8), ("fec", C.c_int), ("image_id", C.c_int), ("packet_id", C.c_int),
                ("width", C.c_int), ("height", C.c_int)]


class SsdvImage(C.Structure):           # wenet_ssdv_image
    _fields_ = [("first", SsdvInfo), ("npackets", C.c_longlong), ("first_index", C.c_longlong)]


class LdpcStruct(C.Structure):          # struct wenet_ldpc == reference struct LDPC (mpdecode_core.h:18-33)
    _fields_ = [(n, C.c_int) for n in (
        "max_iter", "dec_type", "q_scale_factor", "r_scale_factor", "CodeLength", "NumberParityBits",
        "NumberRowsHcols", "max_row_weight", "max_col_weight", "data_bits_per_frame",
        "coded_bits_per_frame", "coded_syms_per_frame")] + [("H_rows", C.c_void_p), ("H_cols", C.c_void_p)]


class ModemStats(C.Structure):
    _fields_ = [("snr_est", C.c_float), ("ppm", C.c_float), ("f_est", C.c_float * 4),
                ("rx_timing", C.c_float), ("foff", C.c_float), ("neyetr", C.c_int), ("neyesamp", C.c_int),
                ("rx_eye", (C.c_float * 160) * 8), ("nfft_est", C.c_int), ("fft_est", C.c_float * 2048)]


_lib = None


def load():
    """Load libwenet_rx.so (raises OSError with a build hint if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C wenet_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i, l, ll, f = C.c_void_p, C.c_int, C.c_long, C.c_longlong, C.c_float
    L.wenet_fsk_create_hbr.restype = vp; L.wenet_fsk_create_hbr.argtypes = [i] * 6
    L.wenet_fsk_create.restype = vp; L.wenet_fsk_create.argtypes = [i] * 5
    L.wenet_fsk_destroy.argtypes = [vp]
    L.wenet_fsk_set_est_limits.argtypes = [vp, i, i]
    L.wenet_fsk_nin.restype = C.c_uint32; L.wenet_fsk_nin.argtypes = [vp]
    L.wenet_fsk_demod.argtypes = [vp, vp, vp]
    L.wenet_fsk_demod_sd.argtypes = [vp, vp, vp]
    L.wenet_fsk_info.argtypes = [vp, i]
    L.wenet_fsk_demod_stream.restype = l
    L.wenet_fsk_demod_stream.argtypes = [vp, i, vp, l, i, vp, l, C.POINTER(l), vp]
    L.wenet_fsk_enable_stats.argtypes = [vp, l, l]
    L.wenet_fsk_get_stats.argtypes = [vp, C.POINTER(ModemStats), i]
    L.wenet_fsk_get_demod_stats.argtypes = [vp, C.POINTER(ModemStats)]
    L.wenet_fsk_last_ebnodb.restype = f; L.wenet_fsk_last_ebnodb.argtypes = [vp]
    L.wenet_run_ldpc_decoder.argtypes = [C.POINTER(LdpcStruct), vp, vp, C.POINTER(i)]
    L.wenet_sd_to_llr.argtypes = [vp, vp, i]
    L.wenet_ldpc_decode_batch.argtypes = [vp, i, i, vp, vp, vp]
    L.wenet_deframer_create.restype = vp; L.wenet_deframer_create.argtypes = [i, i]
    L.wenet_deframer_destroy.argtypes = [vp]
    L.wenet_deframer_push.restype = l; L.wenet_deframer_push.argtypes = [vp, vp, l, vp, vp, l]
    L.wenet_rx_create.restype = vp; L.wenet_rx_create.argtypes = [i] * 8
    L.wenet_rx_destroy.argtypes = [vp]
    L.wenet_rx_process.argtypes = [vp, i, vp, vp, i, i, vp]
    L.wenet_rx_enqueue.argtypes = [vp, i, vp, vp, i, vp]
    L.wenet_rx_collect.argtypes = [vp]
    L.wenet_rx_push.restype = ll; L.wenet_rx_push.argtypes = [vp, i, vp, vp, i]
    L.wenet_rx_flush.argtypes = [vp]
    L.wenet_rx_live_gathered.argtypes = [vp]
    L.wenet_rx_pin_host.argtypes = [vp, C.c_size_t]
    L.wenet_rx_unpin_host.argtypes = [vp]
    L.wenet_rx_frames.restype = ll; L.wenet_rx_frames.argtypes = [vp, i]
    L.wenet_rx_packets.restype = ll; L.wenet_rx_packets.argtypes = [vp, i]
    L.wenet_rx_get_packets.restype = ll; L.wenet_rx_get_packets.argtypes = [vp, i, vp, vp, ll]
    L.wenet_rx_packet_census.argtypes = [vp, i, C.POINTER(ll)]
    L.wenet_rx_get_soft.restype = ll; L.wenet_rx_get_soft.argtypes = [vp, i, vp, ll]
    L.wenet_rx_enable_trace.argtypes = [vp, i]
    L.wenet_rx_get_trace.restype = ll; L.wenet_rx_get_trace.argtypes = [vp, i, vp, ll]
    L.wenet_rx_enable_llr_dump.argtypes = [vp, i]
    L.wenet_rx_get_llrs.restype = ll; L.wenet_rx_get_llrs.argtypes = [vp, i, vp, ll]
    L.wenet_rx_last_ms.restype = f; L.wenet_rx_last_ms.argtypes = [vp, i]
    L.wenet_rx_last_kernel.restype = C.c_char_p; L.wenet_rx_last_kernel.argtypes = [vp]
    L.wenet_rx_get_device.restype = i; L.wenet_rx_get_device.argtypes = [vp]
    L.wenet_rx_channel_counter.restype = ll; L.wenet_rx_channel_counter.argtypes = [vp, i, i]
    L.wenet_rx_set_cf32_quantise.restype = i; L.wenet_rx_set_cf32_quantise.argtypes = [vp, i]
    L.wenet_packet_type_class.argtypes = [vp]
    L.wenet_ssdv_packet_info.argtypes = [vp, C.POINTER(SsdvInfo)]
    L.wenet_rx_get_packets_of_class.restype = ll; L.wenet_rx_get_packets_of_class.argtypes = [vp, i, i, vp, ll]
    L.wenet_rx_ssdv_images.restype = ll; L.wenet_rx_ssdv_images.argtypes = [vp, i, C.POINTER(SsdvImage), ll]
    L.wenet_rx_device_info.argtypes = [i]
    L.wenet_rx_version.restype = C.c_char_p
    L.wenet_rx_source_id.restype = C.c_char_p
    L.wenet_phi0_eval.argtypes = [vp, vp, l]
    L.wenet_rx_decoder_repeats.restype = ll; L.wenet_rx_decoder_repeats.argtypes = [vp]
    L.wenet_rx_result_digest.restype = C.c_ulonglong; L.wenet_rx_result_digest.argtypes = [vp, C.POINTER(ll), C.POINTER(ll)]
    d = C.c_double
    L.wenet_tx_create.restype = vp; L.wenet_tx_create.argtypes = [i, i, i, i, d, d]
    L.wenet_tx_destroy.argtypes = [vp]
    L.wenet_tx_symbols_per_packet.restype = ll; L.wenet_tx_symbols_per_packet.argtypes = [vp]
    L.wenet_tx_frame_packets.argtypes = [vp, vp, ll, vp, i, vp]
    L.wenet_tx_modulate.argtypes = [vp, i, vp, vp, vp, vp, vp, i, vp, vp]
    _lib = L
    return L
