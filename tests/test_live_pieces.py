"""CPU: the cut of a live tick's chunk into the pieces the gather kernel publishes (wenet_rx.hip, live_piece) -- the host's staging of pageable chunks and the kernel
use the same arithmetic, so the pieces must tile the chunk exactly, in order, on 16-byte boundaries of the DESTINATION, with the short first piece the demodulator's
prologue needs."""
import ctypes as C

import numpy as np

from wenet_amd.lib import load


def _pieces(L, n, mis, first):
    P = L.wenet_rx_debug_live_pieces()
    lo, hi = C.c_longlong(), C.c_longlong()
    out = []
    for p in range(P):
        L.wenet_rx_debug_live_piece(C.c_longlong(n), C.c_uint(mis), P, p, C.c_longlong(first), C.byref(lo), C.byref(hi))
        out.append((lo.value, hi.value))
    return out


def test_pieces_tile_the_chunk_in_order_on_destination_boundaries():
    L = load()
    L.wenet_rx_debug_live_piece.argtypes = [C.c_longlong, C.c_uint, C.c_int, C.c_int, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.wenet_rx_debug_live_piece.restype = None
    L.wenet_rx_debug_live_pieces.restype = C.c_int
    rng = np.random.default_rng(11)
    sizes = [0, 1, 2, 15, 16, 17, 31, 33, 255, 256, 4096, 230354, 921416] + [int(x) for x in rng.integers(1, 3_000_000, 40)]
    for n in sizes:
        for mis in (0, 1, 2, 7, 8, 15):
            for first in (0, 1, 459, 10**9):
                pc = _pieces(L, n, mis, first)
                assert pc[0][0] == 0 and pc[-1][1] == n, (n, mis, first, pc)
                head = min(n, (16 - mis) % 16)
                for (lo, hi), (lo2, _) in zip(pc, pc[1:]):
                    assert lo <= hi == lo2, (n, mis, first, pc)                      # in order, no gap, no overlap
                    if hi not in (0, n):
                        assert (hi + mis) % 16 == 0 and hi >= head, (n, mis, first, pc)      # inner cuts on 16-byte boundaries of the destination, behind the head
                body = (n - head) // 16
                if body > 0 and len(pc) > 1:
                    assert pc[0][1] == head + min(body, first) * 16, (n, mis, first, pc)    # the short first piece
