"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import os
import re

from conftest import have_gpu
from wenet_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="wenet_rx.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wenet_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert missing == []
    assert sorted(lib.EXPORTS) == names
    names_tx = declared_functions("wenet_tx.h")
    assert [n for n in names_tx if not hasattr(L, n)] == []
    assert sorted(lib.EXPORTS_TX) == names_tx
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["wenet_rx.h", "wenet_tx.h"]


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "wenet_rx.h")).read()
    assert txt.count("src/") >= 20


def test_fails_loudly_without_gpu():
    L = lib.load()
    assert L.wenet_rx_version().startswith(b"wenet_rx")
    if not have_gpu():
        assert not L.wenet_fsk_create_hbr(960000, 96000, 10, 2, 1200, 400)
        assert not L.wenet_rx_create(960000, 96000, 10, 2, 2, 10, 0, 0)
        assert not L.wenet_deframer_create(2, 10)
        assert not L.wenet_tx_create(960000, 96000, 2, 2, 168000.0, 96000.0)


def test_illegal_rates_are_rejected():
    # the reference asserts (src/fsk.c:137-146); the library returns NULL.  Parameter validation is host code
    # but device presence is checked first, so only meaningful on a GPU box; on CPU both are NULL anyway.
    L = lib.load()
    assert not L.wenet_fsk_create_hbr(921600, 96000, 9, 2, 1200, 400)     # Fs % Rs != 0 (BASELINE config 2 as written)
    assert not L.wenet_fsk_create_hbr(960000, 96000, 10, 3, 1200, 400)    # M must be 2 or 4
