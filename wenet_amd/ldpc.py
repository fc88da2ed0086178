"""Host mirror of the reference's LDPC entry points (src/mpdecode_core.h:35-39) and of the
drs232_ldpc / wenet_ldpc symbol loop, over libwenet_rx.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

CODELENGTH = 2580
NUMBERPARITYBITS = 516
NUMBERROWSHCOLS = 2064
MAX_ROW_WEIGHT = 12
MAX_COL_WEIGHT = 3
MAX_ITER = 10


def make_ldpc_struct(max_iter=MAX_ITER):
    """struct LDPC as the CLI mains fill it (src/drs232_ldpc.c:128-138)."""
    s = _lib.LdpcStruct()
    s.max_iter, s.dec_type, s.q_scale_factor, s.r_scale_factor = max_iter, 0, 1, 1
    s.CodeLength, s.NumberParityBits, s.NumberRowsHcols = CODELENGTH, NUMBERPARITYBITS, NUMBERROWSHCOLS
    s.max_row_weight, s.max_col_weight = MAX_ROW_WEIGHT, MAX_COL_WEIGHT
    return s


def run_ldpc_decoder(ldpc, llr, parity_check_count=0):
    """run_ldpc_decoder: returns (iter, out_char[2580], parityCheckCount)."""
    L = _lib.load()
    x = np.ascontiguousarray(llr, np.float32)
    assert x.size == CODELENGTH
    out = np.zeros(CODELENGTH, np.uint8)
    pcc = C.c_int(parity_check_count)
    it = L.wenet_run_ldpc_decoder(C.byref(ldpc), out.ctypes.data, x.ctypes.data, C.byref(pcc))
    if it < 0:
        raise RuntimeError(f"wenet_run_ldpc_decoder failed ({it})")
    return it, out, pcc.value


def sd_to_llr(sd):
    L = _lib.load()
    x = np.ascontiguousarray(sd, np.float64)
    llr = np.zeros(x.size, np.float32)
    L.wenet_sd_to_llr(llr.ctypes.data, x.ctypes.data, x.size)
    return llr


def ldpc_decode_batch(llrs, max_iter=MAX_ITER):
    L = _lib.load()
    x = np.ascontiguousarray(llrs, np.float32).reshape(-1, CODELENGTH)
    n = x.shape[0]
    bits = np.zeros((n, CODELENGTH), np.uint8)
    iters = np.zeros(n, np.int32)
    pcc = np.full(n, -1, np.int32)
    rc = L.wenet_ldpc_decode_batch(x.ctypes.data, n, max_iter, bits.ctypes.data, iters.ctypes.data, pcc.ctypes.data)
    if rc < 0:
        raise RuntimeError(f"wenet_ldpc_decode_batch failed ({rc})")
    return bits, iters, pcc


class Deframer:
    """The symbol loop of drs232_ldpc (mode 1) / wenet_ldpc (mode 2): push soft symbols, get packets."""

    def __init__(self, mode, max_iter=MAX_ITER):
        self._L = _lib.load()
        self._h = self._L.wenet_deframer_create(mode, max_iter)
        if not self._h:
            raise RuntimeError("wenet_deframer_create failed (no GPU?)")
        self.mode = mode
        self.spp = 323 * (10 if mode == 1 else 8)

    def close(self):
        if self._h:
            self._L.wenet_deframer_destroy(self._h)
            self._h = None

    __del__ = close

    def push(self, symbols):
        x = np.ascontiguousarray(symbols, np.float32)
        cap = x.size // self.spp + 4
        pk = np.zeros((cap, 258), np.uint8)
        info = (_lib.PacketInfo * cap)()
        n = self._L.wenet_deframer_push(self._h, x.ctypes.data, x.size, pk.ctypes.data, info, cap)
        if n < 0:
            raise RuntimeError(f"wenet_deframer_push failed ({n})")
        return dict(n=n, bytes=pk[:n], iter=np.array([info[i].iter for i in range(n)], np.int32),
                    crc_ok=np.array([bool(info[i].crc_ok) for i in range(n)]),
                    start=np.array([info[i].start_symbol for i in range(n)], np.int64))
