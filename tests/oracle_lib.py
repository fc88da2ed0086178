"""ctypes bindings for the TEST-ONLY oracle libraries.

  oracle/libwenet_oracle.so    this repository's plain-C restatement (oracle/wenet_oracle.c)
  oracle/_ref/libwenet_ref.so  the UNMODIFIED reference sources compiled by oracle/Makefile
                               (+ oracle/ref_shim.c accessors); optional -- present wherever
                               `make -C oracle ref` has run (it travels to the GPU box prebuilt)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

FMT = {"s16": 0, "cs16": 1, "cu8": 2, "cf32": 3}
BYTES_PER_SAMPLE = {"s16": 2, "cs16": 4, "cu8": 2, "cf32": 8}

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_ora = None
_ref = None


def oracle():
    global _ora
    if _ora is None:
        path = os.path.join(ORACLE_DIR, "libwenet_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.ora_fsk_create_hbr.restype = C.c_void_p
        L.ora_fsk_create_hbr.argtypes = [C.c_int] * 4
        L.ora_fsk_create.restype = C.c_void_p
        L.ora_fsk_create.argtypes = [C.c_int] * 3
        L.ora_fsk_destroy.argtypes = [C.c_void_p]
        L.ora_fsk_set_est_limits.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ora_fsk_nin.argtypes = [C.c_void_p]
        L.ora_fsk_demod_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_fsk_geom.argtypes = [C.c_void_p, C.c_int]
        for n in ("f_est", "phi_c", "fft_est", "hann", "samp_old"):
            getattr(L, "ora_fsk_get_" + n).argtypes = [C.c_void_p, _f32p]
        L.ora_fsk_get_scalar.restype = C.c_float
        L.ora_fsk_get_scalar.argtypes = [C.c_void_p, C.c_int]
        L.ora_fsk_get_eye.argtypes = [C.c_void_p, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ora_convert_samples.argtypes = [C.c_int, C.c_void_p, C.c_long, _f32p]
        L.ora_demod_capture.restype = C.c_long
        L.ora_demod_capture.argtypes = [C.c_int, C.c_void_p, C.c_long] + [C.c_int] * 6 + \
            [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.ora_phi0.restype = C.c_float
        L.ora_phi0.argtypes = [C.c_float]
        L.ora_sd_to_llr.argtypes = [_f32p, _f64p, C.c_int]
        L.ora_ldpc_decode.argtypes = [_f32p, C.c_int, _u8p, C.POINTER(C.c_int)]
        L.ora_ldpc_encode.argtypes = [_u8p, _u8p]
        L.ora_crc16.restype = C.c_uint16
        L.ora_crc16.argtypes = [_u8p, C.c_int]
        L.ora_deframe_decode.restype = C.c_long
        L.ora_deframe_decode.argtypes = [C.c_int, _f32p, C.c_long, C.c_int, C.c_long,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _ora = L
    return _ora


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libwenet_ref.so"))


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(REF_DIR, "libwenet_ref.so"))
        L.fsk_create_hbr.restype = C.c_void_p
        L.fsk_create_hbr.argtypes = [C.c_int] * 6
        L.fsk_create.restype = C.c_void_p
        L.fsk_create.argtypes = [C.c_int] * 5
        L.fsk_destroy.argtypes = [C.c_void_p]
        L.fsk_set_est_limits.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.fsk_nin.restype = C.c_uint32
        L.fsk_nin.argtypes = [C.c_void_p]
        L.fsk_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.fsk_demod_sd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for n in ("Ndft", "N", "Ts", "Nmem", "P", "Nsym", "Nbits", "nstash", "mode", "est_min", "est_max",
                  "est_space", "nin_field", "neyesamp", "neyetr"):
            getattr(L, "ref_fsk_" + n).argtypes = [C.c_void_p]
        for n in ("norm_rx_timing", "ppm", "EbNodB", "snr_est", "stats_rx_timing", "foff"):
            getattr(L, "ref_fsk_" + n).argtypes = [C.c_void_p]
            getattr(L, "ref_fsk_" + n).restype = C.c_float
        for n in ("f_est", "phi_c", "fft_est", "hann", "samp_old", "rx_eye"):
            getattr(L, "ref_fsk_" + n).argtypes = [C.c_void_p, _f32p]
        L.ref_ldpc_H_rows.restype = C.POINTER(C.c_uint16)
        L.ref_ldpc_H_cols.restype = C.POINTER(C.c_uint16)
        L.ref_ldpc_kat.argtypes = [_f32p, _u8p]
        L.ref_ldpc_decode.argtypes = [_f32p, C.c_int, _u8p, C.POINTER(C.c_int)]
        L.ref_phi0.restype = C.c_float
        L.ref_phi0.argtypes = [C.c_float]
        L.sd_to_llr.argtypes = [_f32p, _f64p, C.c_int]
        L.ref_scramble_code.restype = C.POINTER(C.c_double)
        _ref = L
    return _ref


# --------------------------------------------------------------------------- helpers
def raw_bytes(raw: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(raw).view(np.uint8).reshape(-1)


def oracle_demod(raw, fmt, Fs, Rs, M, P=0, est=(0, 0), hard=False, want_trace=False, lbr=False):
    """Whole-capture oracle demod.  Returns (sd or bits, trace or None).  lbr: the fsk_create geometry (fsk_demod -l)."""
    L = oracle()
    rb = raw_bytes(raw)
    nsamp = rb.size // BYTES_PER_SAMPLE[fmt]
    if P == 0:
        P = Fs // Rs
    Ts = Fs // Rs
    nsym = Rs if lbr else 48
    if lbr:
        P = -1
    nbits = nsym * (1 if M == 2 else 2)
    cap = nsamp // (nsym * Ts - Ts // 2) + 2
    sd = np.zeros(cap * nbits, np.float32)
    bits = np.zeros(cap * nbits, np.uint8)
    trace = np.zeros((cap, 8), np.float32)
    n = L.ora_demod_capture(FMT[fmt], rb.ctypes.data, nsamp, Fs, Rs, P, M, est[0], est[1],
                            None if hard else sd.ctypes.data, bits.ctypes.data if hard else None, cap,
                            trace.ctypes.data if want_trace else None)
    assert n >= 0
    out = bits[:n * nbits] if hard else sd[:n * nbits]
    return out, (trace[:n] if want_trace else None)


def oracle_deframe(sd, mode, max_iter=10, want_llr=False):
    L = oracle()
    sd = np.ascontiguousarray(sd, np.float32)
    spp = 323 * (10 if mode == 1 else 8)
    cap = sd.size // spp + 2
    start = np.zeros(cap, np.int64)
    it = np.zeros(cap, np.int32)
    ok = np.zeros(cap, np.uint8)
    pk = np.zeros((cap, 258), np.uint8)
    llr = np.zeros((cap, 2580), np.float32) if want_llr else None
    n = L.ora_deframe_decode(mode, sd, sd.size, max_iter, cap, start.ctypes.data, it.ctypes.data,
                             ok.ctypes.data, pk.ctypes.data, llr.ctypes.data if want_llr else None)
    res = dict(n=n, start=start[:n], iter=it[:n], crc_ok=ok[:n].astype(bool), bytes=pk[:n])
    if want_llr:
        res["llr"] = llr[:n]
    return res


def ref_cli_demod(raw, fmt, Fs, Rs, M, soft=True, extra=()):
    """Run the reference fsk_demod binary (oracle/_ref/fsk_demod) file -> bytes."""
    args = [os.path.join(REF_DIR, "fsk_demod")]
    if fmt == "cu8":
        args.append("--cu8")
    elif fmt == "cs16":
        args.append("--cs16")
    if soft:
        args.append("-s")
    args += list(extra) + [str(M), str(Fs), str(Rs), "-", "-"]
    p = subprocess.run(args, input=raw_bytes(raw).tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return np.frombuffer(p.stdout, dtype=np.float32 if soft else np.uint8), p.stderr


def ref_cli_ldpc(sd, mode, verbose=""):
    exe = os.path.join(REF_DIR, "drs232_ldpc" if mode == 1 else "wenet_ldpc")
    args = [exe, "-", "-"] + ([verbose] if verbose else [])
    p = subprocess.run(args, input=np.ascontiguousarray(sd, np.float32).tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return p.stdout, p.stderr
