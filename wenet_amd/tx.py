"""Batched Wenet frame builder / M-FSK test-signal generator on the GPU (include/wenet_tx.h).

Host-side mirror of the transmitter's framing vocabulary (tx/PacketTX.py `frame_packet`,
tx/radio_wrappers.py `scramble` / bit expansion); the work is done by the gfx950 kernels in
libwenet_rx.so.  No CPU path: wenet_amd/siggen.py is the independent numpy statement of the same
format that the tests compare against.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .fsk import FMT


class Tx:
    def __init__(self, Fs, Rs, M, framing, f_low, f_space):
        self._L = _lib.load()
        self._h = self._L.wenet_tx_create(Fs, Rs, M, framing, float(f_low), float(f_space))
        if not self._h:
            raise RuntimeError("wenet_tx_create failed (illegal parameters or no GPU)")
        self.Fs, self.Rs, self.M, self.framing, self.Ts = Fs, Rs, M, framing, Fs // Rs
        self.symbols_per_packet = int(self._L.wenet_tx_symbols_per_packet(self._h))

    @classmethod
    def from_config(cls, cfg):
        """cfg: wenet_amd.siggen.ModemConfig"""
        return cls(cfg.Fs, cfg.Rs, cfg.M, cfg.mode, cfg.f_low, cfg.f_space)

    def close(self):
        if self._h:
            self._L.wenet_tx_destroy(self._h)
            self._h = None

    __del__ = close

    def frame_packets(self, payloads) -> np.ndarray:
        """payloads: [n, 256] uint8 (host) -> tone indices of n back-to-back frames (host)."""
        p = np.ascontiguousarray(payloads, dtype=np.uint8).reshape(-1, 256)
        out = np.zeros(p.shape[0] * self.symbols_per_packet, np.uint8)
        rc = self._L.wenet_tx_frame_packets(self._h, p.ctypes.data, p.shape[0], out.ctypes.data, 0, None)
        if rc < 0:
            raise RuntimeError(f"wenet_tx_frame_packets failed ({rc})")
        return out

    def frame_packets_device(self, payload_ptr, npackets, symbols_ptr, stream=None):
        rc = self._L.wenet_tx_frame_packets(self._h, C.c_void_p(payload_ptr), npackets, C.c_void_p(symbols_ptr), 1,
                                            C.c_void_p(stream) if stream else None)
        if rc < 0:
            raise RuntimeError(f"wenet_tx_frame_packets failed ({rc})")

    def modulate_device(self, sym_ptrs, nsyms, out_ptrs, ebno_db, ppm=None, seeds=None, fmt="cu8", stream=None):
        """All pointers are device addresses; out[c] must hold nsyms[c]*Ts samples of format fmt."""
        n = len(sym_ptrs)
        sp = (C.c_void_p * n)(*sym_ptrs)
        op = (C.c_void_p * n)(*out_ptrs)
        ns = (C.c_longlong * n)(*nsyms)
        eb = (C.c_double * n)(*([float(ebno_db)] * n if np.isscalar(ebno_db) else [float(e) for e in ebno_db]))
        pm = None if ppm is None else (C.c_double * n)(*([float(ppm)] * n if np.isscalar(ppm) else [float(e) for e in ppm]))
        sd = None if seeds is None else (C.c_uint64 * n)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds])
        rc = self._L.wenet_tx_modulate(self._h, n, sp, ns, eb, pm, sd, FMT[fmt], op, C.c_void_p(stream) if stream else None)
        if rc < 0:
            raise RuntimeError(f"wenet_tx_modulate failed ({rc})")
