"""Host mirror of the reference's FSK modem API (src/fsk.h:100-202) over libwenet_rx.so.

Names and argument meaning follow the reference: fsk_create_hbr / fsk_nin / fsk_demod /
fsk_demod_sd / fsk_set_est_limits / fsk_destroy.  All arithmetic runs in the gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

FMT = {"s16": 0, "cs16": 1, "cu8": 2, "cf32": 3}
BYTES_PER_SAMPLE = {"s16": 2, "cs16": 4, "cu8": 2, "cf32": 8}
TRACE_FLOATS = 10


class Fsk:
    """struct FSK* handle."""

    def __init__(self, Fs, Rs, P, M, tx_f1=1200, tx_fs=400, lbr=False):
        """fsk_create_hbr(Fs, Rs, P, M, tx_f1, tx_fs); lbr=True: fsk_create(Fs, Rs, M, tx_f1, tx_fs) (P is 8 there)."""
        self._L = _lib.load()
        if lbr:
            self._h = self._L.wenet_fsk_create(Fs, Rs, M, tx_f1, tx_fs)
        else:
            self._h = self._L.wenet_fsk_create_hbr(Fs, Rs, P, M, tx_f1, tx_fs)
        if not self._h:
            raise RuntimeError("fsk_create%s failed (illegal parameters or no GPU)" % ("" if lbr else "_hbr"))
        info = lambda k: self._L.wenet_fsk_info(self._h, k)
        self.Fs, self.Rs, self.P, self.M = Fs, Rs, info(4), M
        self.Ndft, self.N, self.Ts, self.Nmem = info(0), info(1), info(2), info(3)
        self.Nsym, self.Nbits, self.nstash = info(5), info(6), info(7)

    def close(self):
        if self._h:
            self._L.wenet_fsk_destroy(self._h)
            self._h = None

    __del__ = close

    def set_est_limits(self, fmin, fmax):
        self._L.wenet_fsk_set_est_limits(self._h, fmin, fmax)

    def nin(self):
        return int(self._L.wenet_fsk_nin(self._h))

    def demod_sd(self, fsk_in):
        """fsk_demod_sd: one frame of nin() complex64 samples -> Nbits float32."""
        x = np.ascontiguousarray(fsk_in, np.complex64)
        assert x.size == self.nin()
        out = np.zeros(self.Nbits, np.float32)
        self._L.wenet_fsk_demod_sd(self._h, out.ctypes.data, x.ctypes.data)
        return out

    def demod(self, fsk_in):
        x = np.ascontiguousarray(fsk_in, np.complex64)
        assert x.size == self.nin()
        out = np.zeros(self.Nbits, np.uint8)
        self._L.wenet_fsk_demod(self._h, out.ctypes.data, x.ctypes.data)
        return out

    def enable_stats(self, first=1, period=1):
        self._L.wenet_fsk_enable_stats(self._h, first, period)

    def demod_stream(self, raw, fmt, soft=True, want_trace=False):
        """fsk_demod main loop over a block of raw samples.  Returns (out, consumed, trace)."""
        rb = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        nsamp = rb.size // BYTES_PER_SAMPLE[fmt]
        cap = nsamp // (self.N - self.Ts // 2) + 1
        out = np.zeros(cap * self.Nbits, np.float32 if soft else np.uint8)
        trace = np.zeros((cap, TRACE_FLOATS), np.float32) if want_trace else None
        used = C.c_long(0)
        n = self._L.wenet_fsk_demod_stream(self._h, FMT[fmt], rb.ctypes.data, nsamp, 1 if soft else 0,
                                           out.ctypes.data, cap, C.byref(used),
                                           trace.ctypes.data if want_trace else None)
        if n < 0:
            raise RuntimeError(f"wenet_fsk_demod_stream failed ({n})")
        return out[:n * self.Nbits], int(used.value), (trace[:n] if want_trace else None)

    def get_stats(self, cap=64):
        arr = (_lib.ModemStats * cap)()
        n = self._L.wenet_fsk_get_stats(self._h, arr, cap)
        return [arr[i] for i in range(n)]

    def get_demod_stats(self):
        """fsk_get_demod_stats (src/fsk.h:130): the statistics after the last frame a snapshot was kept for."""
        st = _lib.ModemStats()
        self._L.wenet_fsk_get_demod_stats(self._h, C.byref(st))
        return st
