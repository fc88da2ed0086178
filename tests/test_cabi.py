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
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["wenet_fsk_compat.h", "wenet_rx.h", "wenet_tx.h"]


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


def test_compat_library_exports_the_reference_link_names_and_a_c_caller_links(tmp_path):
    """libwenet_fsk_compat.so carries the reference's own names (src/fsk.h:100-202, src/mpdecode_core.h:37-39): every function include/wenet_fsk_compat.h
    declares is exported, and a small C caller written against those prototypes compiles and LINKS (it is not run here: that needs a GPU --
    tests/test_gpu_compat.py runs the reference's own mains on the library)."""
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "wenet_amd", "libwenet_fsk_compat.so")
    assert os.path.exists(so), "build with make -C wenet_amd/csrc"
    txt = open(os.path.join(ROOT, "include", "wenet_fsk_compat.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b((?:fsk_|run_ldpc_decoder|sd_to_llr)[a-z0-9_]*)\s*\(", txt)))
    assert {"fsk_create_hbr", "fsk_nin", "fsk_demod_sd", "fsk_get_demod_stats", "fsk_destroy", "run_ldpc_decoder", "sd_to_llr"} <= set(names)
    L = C.CDLL(so)
    assert [n for n in names if not hasattr(L, n)] == []
    src = tmp_path / "caller.c"
    src.write_text("""
#include <stdio.h>
#include <stdlib.h>
#include "wenet_fsk_compat.h"
int main(int argc, char **argv) {                         /* the shape of src/fsk_demod.c:214-300 */
    struct FSK *fsk = fsk_create_hbr(960000, 96000, 10, 2, 1200, 400);
    fsk_set_est_limits(fsk, 100000, 330000);
    COMP *in = calloc(fsk->N + 2 * fsk->Ts, sizeof(COMP));
    float *sd = malloc(sizeof(float) * fsk->Nbits);
    struct MODEM_STATS stats;
    while (fread(in, sizeof(COMP), fsk_nin(fsk), stdin) == fsk_nin(fsk)) {
        fsk_demod_sd(fsk, sd, in);
        fsk_get_demod_stats(fsk, &stats);
        fwrite(sd, sizeof(float), fsk->Nbits, stdout);
        fprintf(stderr, "%f %d %f\\n", stats.snr_est, (int)fsk->ppm, fsk->f_est[0]);
    }
    fsk_destroy(fsk);
    return 0;
}
""")
    exe = tmp_path / "caller"
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "wenet_amd"), "-lwenet_fsk_compat", "-lwenet_rx", "-Wl,-rpath," + os.path.join(ROOT, "wenet_amd")])
    assert os.path.exists(exe)
